// Do plain VALU instructions overlap with v_mfma_f32_32x32x2_f32 on a gfx950 SIMD?  A wave loops over
//   4 independent MFMAs (256 matrix-pipe cycles) + NV independent v_fma_f32 (4 cycles each on a 16-lane SIMD)
// with 1, 2 and 4 waves per SIMD.  If the two kinds of work overlapped perfectly the loop would cost max(256, 4 NV) cycles
// per wave and iteration; if the VALU instruction stream steals the issue port from the matrix pipe it costs the SUM.
// (Round 4: counters showed the fp32 weight-gradient kernel spends 1.7x the VALU-active cycles of the forward GEMM on the
// same product and runs 15-20 % slower; the fused fp32 tail -- a BatchNorm / ReLU epilogue per output in VALU -- tops out at
// 66 % MFMA busy.)
//   hipcc --offload-arch=gfx950 -O3 tools/hw_probe/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a0 + i); bb[i] = (__bf16)b0; }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);      // 8 passes: 32 cycles
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);                // 16 passes: 64 cycles
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool BF>
void run(int bpc, int cus, double mhz) {
    float* out;
    hipMalloc(&out, sizeof(float) * 1024 * 4096);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, BF><<<cus * bpc, 256>>>(out, 64, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, BF><<<cus * bpc, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * mhz * 1e6 / iters / bpc;       // SIMD cycles per iteration of ONE wave (bpc waves share a SIMD)
    const double mf = BF ? 128.0 : 256.0;
    printf("%s %2d VALU per 4 MFMA, %d wave(s)/SIMD: %7.3f ms  %6.1f cycles per wave-iteration (MFMA alone %.0f, VALU alone %d)  MFMA pipe %.0f %%\n", BF ? "bf16 32x32x16" : "fp32 32x32x2 ", NV, bpc, ms, cyc,
           mf, 4 * NV, mf / cyc * 100);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs, clock %.0f MHz\n", p.name, cus, mhz);
    for (int bpc : {1, 2, 4}) {
        run<0, false>(bpc, cus, mhz);
        run<8, false>(bpc, cus, mhz);
        run<16, false>(bpc, cus, mhz);
        run<32, false>(bpc, cus, mhz);
        run<64, false>(bpc, cus, mhz);
        run<0, true>(bpc, cus, mhz);
        run<4, true>(bpc, cus, mhz);
        run<8, true>(bpc, cus, mhz);
        run<16, true>(bpc, cus, mhz);
        run<32, true>(bpc, cus, mhz);
    }
    return 0;
}
