// What does v_mfma_f32_32x32x2_f32 sustain on MI355X, as a function of how many INDEPENDENT accumulators a wave rotates
// through and of the waves per SIMD?  (Round 4: the fp32 fused tail tops out at ~77 % of the 157.3 TFLOP/s peak even with
// every load, store and LDS read knocked out; gemm_f32_kernel sits at 76-77 % MFMA busy.)
//   hipcc --offload-arch=gfx950 -O3 tools/hw_probe/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int threads, int blocks_per_cu, int cus) {
    float* out;
    hipMalloc(&out, sizeof(float) * 1024 * 4096);
    const int iters = 4096 / NACC;                 // same MFMA count per wave for every NACC
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<cus * blocks_per_cu, threads>>>(out, 16, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<cus * blocks_per_cu, threads>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)cus * blocks_per_cu * (threads / 64) * iters * 16.0 * NACC;
    printf("accumulators %d, %d waves/CU: %.3f ms, %.1f TFLOP/s (%.1f %% of 157.3)\n", NACC, blocks_per_cu * threads / 64, ms,
           mfmas * 4096.0 / ms / 1e9, mfmas * 4096.0 / ms / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    for (int waves : {4, 8, 16}) {
        const int threads = 256, bpc = waves / 4;
        run<1>(threads, bpc, cus);
        run<2>(threads, bpc, cus);
        run<4>(threads, bpc, cus);
    }
    return 0;
}
