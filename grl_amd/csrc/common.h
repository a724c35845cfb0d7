// Shared helpers for the libgrl_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int grl_fail(int code, const char* fmt, ...);          // sets the thread-local message
int grl_check_launch(const char* what);                 // hipGetLastError -> GRL_ELAUNCH

static inline int grl_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// wave64 sum (all lanes receive the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum for <= 16 waves; `red` is >= 16 floats of LDS. All threads get the total.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// gemm_bf16.hip: 256 x 256 bf16-storage tile.  1 = launched, 0 = shape not covered (use the
// 128 x 128 family), < 0 = error.
struct GrlGemm;
int grl_gemm_bf16_256(const GrlGemm& d, hipStream_t s);
bool grl_gemm_bf16_256_takes(const GrlGemm& d);        // would it?  (same predicate)
int grl_gemm_bf16_256_stat_rows(const GrlGemm& d);     // rows of the statistics slab it writes (2 per 256-row tile)
int grl_gemm_validate(const GrlGemm& d);               // gemm_f32.hip: the argument checks of grl_conv_gemm_f32 (GRL_OK or grl_fail)

// train.hip: BatchNorm-backward finalize over an fp32 partial slab (shared with the bf16-storage kernels of train_bf16.hip)
int grl_launch_bn_bwd_finalize(const float* slab, int rows, int C, double count, float* dgamma, float* dbeta, float* coef,
                               hipStream_t s);

// train_bnfuse.hip: BatchNorm finalize inside the apply pass (rows <= 64 slab rows, C % 64 == 0): the backward entry points of
// train.hip / train_bf16.hip launch it instead of bn_bwd_finalize + bn_bwd_apply
bool grl_bn_finapply_takes(int rows, int C);
int grl_launch_bn_bwd_finapply(int b16, const float* slab, int rows, int C, double count, float* dgamma, float* dbeta, const void* dy,
                               const void* z, const void* act, const float* mean, const float* invstd, const float* gamma, void* dz,
                               int M, void* gres, int gres_accumulate, const float* mscale, const float* mbeta, const uint8_t* bits,
                               hipStream_t s);
