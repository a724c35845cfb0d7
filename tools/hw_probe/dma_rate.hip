// What does the L2 -> LDS path of a CU sustain?  One 512-thread workgroup per CU streams GEMM-like operand pieces into an
// LDS ring by LDS-DMA (global_load_lds_dwordx4) or into registers (global_load_dwordx4), no compute.
//   piece shape  : SEG bytes contiguous per row (64 / 128 / 256 / 1024), 1024 / SEG rows per wave-instruction
//   row stride   : LD bytes (4096 = the K = 2048 bf16 operand; 4224 = padded by 128 B)
//   sharing      : SHARE workgroups read the same rows (the tiles of one XCD share operand panels)
//   depth        : wave-instructions in flight per wave before it waits (vmcnt)
//   policy       : 0 default, 1 nt, 2 sc1, 3 sc0 sc1
//   hipcc -O3 --offload-arch=gfx950 tools/hw_probe/dma_rate.hip -o tools/hw_probe/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int POL> __device__ __forceinline__ void dma16(const char* sbase, uint32_t voff, uint32_t lds) {
    if constexpr (POL == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
    if constexpr (POL == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
    if constexpr (POL == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
    if constexpr (POL == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0 sc1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Each workgroup owns a panel of 512 rows (256 "A" + 256 "W" rows) x KB bytes; per step it moves 64 x (512 rows) bytes
// = 32 KiB (STEP_SEG = 64) as 32 wave-instructions (4 per wave).  With SEG > 64 an instruction covers SEG bytes of
// 1024 / SEG rows and a step of 32 instructions covers SEG x 512 rows.
template <int SEG, int POL, int DEPTH, bool REG, int PAIR = 0>
__global__ __launch_bounds__(512, 2) void stream_kernel(const char* __restrict__ base, int64_t ld, int kbytes, int share, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RPI = 1024 / SEG;                  // rows per instruction
    constexpr int LPR = SEG / 16;                    // lanes per row
    // blocks b, b + 8, ... run on one XCD: the workgroups that share a panel sit behind the same L2
    const int panel = (blockIdx.x & 7) * (32 / share) + (blockIdx.x >> 3) / share;
    const char* pbase = base + (int64_t)panel * 512 * ld;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // instruction j of a step (j = wave * 4 + i): rows [j * RPI, (j+1) * RPI) of ... 512 rows need 512 / RPI instructions
    // = 8 * SEG / 16 ... keep 32 instructions per step: they cover 32 * RPI rows; SEG = 64: 512 rows.  For larger SEG
    // a step covers 32 * RPI rows x SEG bytes, and the next step the next 32 * RPI rows (same k), then k advances.
    constexpr int ROWS_PER_STEP = PAIR ? 256 : 32 * RPI;
    constexpr int SUB = 512 / ROWS_PER_STEP;         // steps per k position
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = (uint32_t)(((wave * 4 + i) * RPI + lane / LPR) * ld + (lane % LPR) * 16);
    if (PAIR == 1) {      // halves of the same 16 rows back to back: (rows 0-15, k-half 0), (rows 0-15, k-half 1), (rows 16-31, 0), ...
#pragma unroll
        for (int i = 0; i < 4; ++i) voff[i] = (uint32_t)(((wave * 2 + i / 2) * 16 + lane / 4) * ld + (i % 2) * 64 + (lane % 4) * 16);
    }
    if (PAIR == 2) {      // k-half 0 of both row blocks, then k-half 1 of both
#pragma unroll
        for (int i = 0; i < 4; ++i) voff[i] = (uint32_t)(((wave * 2 + i % 2) * 16 + lane / 4) * ld + (i / 2) * 64 + (lane % 4) * 16);
    }
    const int ksteps = PAIR ? kbytes / 128 : kbytes / SEG;
    float acc = 0.f;
    int slot = 0;
    constexpr int INFLIGHT = DEPTH;                  // steps in flight
    int issued = 0;
    for (int it = 0; it < iters; ++it) {
        for (int ks = 0; ks < ksteps; ++ks) {
            for (int sub = 0; sub < SUB; ++sub) {
                const char* sb = pbase + (int64_t)sub * ROWS_PER_STEP * ld + (int64_t)ks * (PAIR ? 128 : SEG);
                if constexpr (REG) {
                    float4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(sb + voff[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc += v[i].x + v[i].w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma16<POL>(sb, voff[i], lds0 + slot * 32768 + (wave * 4 + i) * 1024);
                    slot = slot + 1 == 4 ? 0 : slot + 1;
                    ++issued;
                    if (issued >= INFLIGHT) {
                        wait_vm<4 * (DEPTH - 1)>();
                        if (DEPTH <= 3) __builtin_amdgcn_s_barrier();     // (a consumer would sync here)
                    }
                }
            }
        }
    }
    wait_vm<0>();
    if (acc == 1.2345f) sink[0] = acc;
}

template <int SEG, int POL, int DEPTH, bool REG, int PAIR = 0>
void run(const char* name, const char* buf, int64_t ld, int kbytes, int share, int iters) {
    auto k = stream_kernel<SEG, POL, DEPTH, REG, PAIR>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    float* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, buf, ld, kbytes, share, 1, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, buf, ld, kbytes, share, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 256.0 * 512 * kbytes * iters;
    printf("%-44s ld %5lld share %2d: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", name, (long long)ld, share, ms * 1e3 / iters,
           bytes / 256 / (ms * 1e-3) * 1e-9, bytes / (ms * 1e-3) * 1e-12);
    CK(hipFree(sink));
}

int main() {
    const int64_t LDMAX = 4224;
    const size_t bytes = (size_t)256 * 512 * LDMAX;          // 256 panels x 512 rows
    char* buf;
    CK(hipMalloc(&buf, bytes + 4096));
    CK(hipMemset(buf, 1, bytes + 4096));
    const int kb = 4096;                                     // K = 2048 bf16
    for (int share : {8, 32}) {
        for (int64_t ld : {4096ll, 1024ll, 512ll}) {
            run<64, 0, 4, false>("dma seg64 depth4", buf, ld, ld < kb ? (int)ld : kb, share, 8);
            run<64, 0, 4, false, 1>("dma seg64 paired (same rows back to back)", buf, ld, ld < kb ? (int)ld : kb, share, 8);
            run<64, 0, 4, false, 2>("dma seg64 paired (2 instr apart)", buf, ld, ld < kb ? (int)ld : kb, share, 8);
            run<64, 0, 8, false, 1>("dma seg64 paired depth8", buf, ld, ld < kb ? (int)ld : kb, share, 8);
            run<128, 0, 4, false>("dma seg128 depth4", buf, ld, ld < kb ? (int)ld : kb, share, 8);
            run<128, 0, 2, false>("dma seg128 depth2 + barrier", buf, ld, ld < kb ? (int)ld : kb, share, 8);
        }
    }
    return 0;
}
