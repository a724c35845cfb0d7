// Main-loop probe for a 4-wave bf16 GEMM tile on MI355X (round 4): ONE wave per SIMD, wave tile 128 x 128 (16 accumulators
// of 32 x 32 = 256 accumulator registers), workgroup tile 256 x 256, K stage = 64 (two 64 KB LDS stages) -- against the
// shipped 8-wave / 128 x 64 wave-tile kernel, which moves 256 KB through LDS per stage (64 KB DMA + 192 KB fragment reads)
// and sits at ~1.0 PFLOP/s.  Here: 64 + 128 KB.  Timing only (the product kernel keeps its own file).
//   MODE 0: fragment reads + MFMA, no staging      MODE 1: + LDS-DMA staging (L2-hot source)
//   MODE 2: + register staging (global_load -> ds_write_b128)
//   hipcc --offload-arch=gfx950 -O3 tools/hw_probe/bf16_loop.hip -o /tmp/bf16_loop && /tmp/bf16_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16_hidden(const char* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}

constexpr int STAGE = 2 * 256 * 128;     // A + B, bytes

template <int MODE, int ILV = 0>
__global__ __launch_bounds__(256) void loop_kernel(const char* __restrict__ a, const char* __restrict__ w, float* out, int nk, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
    const int srow = lane >> 3, schunk = lane & 7;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging sources: wave w fills 8-row pieces w, w+4, ... (16 per operand per... 32 pieces per operand per stage: 8 per wave)
    uint32_t aoff[8], boff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave + 4 * i) * 8 + srow;
        const uint32_t sw = (uint32_t)((schunk ^ ((r >> 1) & 7)) << 4);
        aoff[i] = (uint32_t)(((blockIdx.x & 7) * 256 + r) * K * 2) + sw;   // (kt * 128 below is taken modulo K * 2 by the caller's nk: see main)
        boff[i] = (uint32_t)(((blockIdx.x >> 3 & 7) * 256 + r) * K * 2) + sw;
    }
    const uint32_t lds0 = (uint32_t)(size_t)((lptr_t)smem);
    uint4 st[MODE == 2 ? 16 : 1];
    auto stage_dma = [&](int s, int kt, int i) {          // piece i of both operands
        glds16_hidden(a, aoff[i] + (uint32_t)(kt & 63) * 128, lds0 + s * STAGE + (wave + 4 * i) * 1024);
        glds16_hidden(w, boff[i] + (uint32_t)(kt & 63) * 128, lds0 + s * STAGE + 256 * 128 + (wave + 4 * i) * 1024);
    };
    auto stage_ld = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            st[MODE == 2 ? 2 * i : 0] = *reinterpret_cast<const uint4*>(a + aoff[i] + (uint32_t)(kt & 63) * 128);
            st[MODE == 2 ? 2 * i + 1 : 0] = *reinterpret_cast<const uint4*>(w + boff[i] + (uint32_t)(kt & 63) * 128);
        }
    };
    auto stage_st = [&](int s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<uint4*>(smem + s * STAGE + (wave + 4 * i) * 1024 + lane * 16) = st[MODE == 2 ? 2 * i : 0];
            *reinterpret_cast<uint4*>(smem + s * STAGE + 256 * 128 + (wave + 4 * i) * 1024 + lane * 16) = st[MODE == 2 ? 2 * i + 1 : 0];
        }
    };
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_dma(0, 0, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 2) { stage_ld(0); stage_st(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* As = smem + cur * STAGE + (wr * 128 + frow) * 128;
        const char* Bs = smem + cur * STAGE + 256 * 128 + (wc * 128 + frow) * 128;
        if (MODE == 2 && kt + 1 < nk) stage_ld(kt + 1);
        bf16x8 af[2][4], bf[2][4];
        auto rd = [&](int set, int q) {
            const int ch = ((2 * q + fhalf) ^ fsw) << 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[set][i] = *reinterpret_cast<const bf16x8*>(As + i * 4096 + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[set][j] = *reinterpret_cast<const bf16x8*>(Bs + j * 4096 + ch);
        };
        rd(0, 0);
        if (ILV == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) rd((q + 1) & 1, q + 1);
            if (MODE == 1 && kt + 1 < nk) { stage_dma(cur ^ 1, kt + 1, 2 * q); stage_dma(cur ^ 1, kt + 1, 2 * q + 1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][i], bf[q & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        } else if (ILV == 3) {                       // no LDS reads at all: the MFMA stream alone
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        } else {                                     // one fragment read (and one DMA pair per 8 MFMAs) in the shadow of every second MFMA
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = ((2 * (q + 1) + fhalf) ^ fsw) << 4;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int i = m >> 2, j = m & 3;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][i], bf[q & 1][j], acc[i][j], 0, 0, 0);
                if ((m & 1) == 1 && q + 1 < 4) {
                    const int f = m >> 1;            // 0..7: A0..A3, B0..B3 of the next k-step
                    if (f < 4) af[(q + 1) & 1][f] = *reinterpret_cast<const bf16x8*>(As + f * 4096 + ch);
                    else bf[(q + 1) & 1][f - 4] = *reinterpret_cast<const bf16x8*>(Bs + (f - 4) * 4096 + ch);
                }
                if (MODE == 1 && kt + 1 < nk && (m == 3 || m == 11)) stage_dma(cur ^ 1, kt + 1, 2 * q + (m == 11));
                if (ILV == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 2 && kt + 1 < nk) {
            stage_st(cur ^ 1);
            __syncthreads();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int ILV = 0>
void run(const char* a, const char* w, float* out, int cus, int nk, int K) {
    hipFuncSetAttribute((const void*)loop_kernel<MODE, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    loop_kernel<MODE, ILV><<<cus, 256, 2 * STAGE>>>(a, w, out, 4, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_kernel<MODE, ILV><<<cus, 256, 2 * STAGE>>>(a, w, out, nk, K);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)cus * 256.0 * 256.0 * 64.0 * nk * 2.0;
    printf("mode %d ilv %d: %.3f ms, %.0f TFLOP/s (%.1f %% of 2500), err %s\n", MODE, ILV, ms, fl / ms / 1e9, fl / ms / 1e9 / 25.0,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, K = 4096, nk = 16 * K / 64;      // (the k offset wraps inside the 2 MB operands: timing only)
    char *a, *w;
    float* out;
    hipMalloc(&a, (size_t)8 * 256 * K * 2);
    hipMalloc(&w, (size_t)8 * 256 * K * 2);
    hipMemset(a, 0, (size_t)8 * 256 * K * 2);
    hipMemset(w, 0, (size_t)8 * 256 * K * 2);
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    run<0, 3>(a, w, out, cus, nk, K);
    run<0, 0>(a, w, out, cus, nk, K);
    run<0, 1>(a, w, out, cus, nk, K);
    run<0, 2>(a, w, out, cus, nk, K);
    run<1, 0>(a, w, out, cus, nk, K);
    run<1, 1>(a, w, out, cus, nk, K);
    run<1, 2>(a, w, out, cus, nk, K);
    return 0;
}
