// BatchNorm finalize INSIDE the apply pass, without a gate (round 6).
//
// A train-mode BatchNorm costs three launches in each direction: the statistics / reduce (fused into the producing GEMM's
// epilogue where possible), a FINALIZE over the partial slab (bn_stats_finalize_kernel / bn_bwd_finalize_kernel: ~6 us,
// 176 launches per step) and the apply pass.  Round 3 tried to run the finalize inside the apply launch behind a device-side
// gate (publish -> poll -> coherent read): slower in every form.  This file does it WITHOUT any cross-workgroup ordering, for
// the layers where that is cheap: a workgroup of the apply pass owns 64 channels x a block of rows and REDUCES THE SLAB
// COLUMNS OF ITS 64 CHANNELS ITSELF.  Redundant across the row blocks of a channel strip, so it only pays while the slab
// is small: rows <= 64 (M <= 8192 pixel rows -- the TRL memo bottleneck layers, 48 of the 80 BatchNorms of a 32 x 4 step;
// 64 rows x 2 x 64 channels = 32 KiB of L2 reads per workgroup).  The sums are taken in EXACTLY the order of slab_totals
// (train.hip): with at most 64 slab rows every row group of its 64- or 256-slot tree holds at most one row, the tree's upper
// levels add exact zeros, and the rest is the same pairing (i, i + 32), (i, i + 16), ... -- the statistics, and with them
// every output, are bit-identical to the three-launch form (tests/test_gpu_train_kernels.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/grl_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int FA_CH = 64;        // channels per workgroup
constexpr int FA_ROWS = 64;      // most slab rows the fused form takes

template <bool B16> struct Vec;
template <> struct Vec<true> {
    static constexpr int N = 8;
    typedef f32x8 F;
    static __device__ __forceinline__ F ld(const void* p, int64_t i) { return __builtin_convertvector(reinterpret_cast<const bf16x8*>(p)[i], f32x8); }
    static __device__ __forceinline__ void st(void* p, int64_t i, F v) { reinterpret_cast<bf16x8*>(p)[i] = __builtin_convertvector(v, bf16x8); }
    static __device__ __forceinline__ float stored(float v) { return (float)(__bf16)v; }
};
template <> struct Vec<false> {
    static constexpr int N = 4;
    typedef f32x4 F;
    static __device__ __forceinline__ F ld(const void* p, int64_t i) { return reinterpret_cast<const f32x4*>(p)[i]; }
    static __device__ __forceinline__ void st(void* p, int64_t i, F v) { reinterpret_cast<f32x4*>(p)[i] = v; }
    static __device__ __forceinline__ float stored(float v) { return v; }
};

// the two column totals of this workgroup's 64 channels, in slab_totals' order; valid in threads 0..63 (channel = thread)
__device__ __forceinline__ void strip_totals(const float* __restrict__ slab, int rows, int C, int c0, float (*part)[FA_ROWS][FA_CH],
                                             double& s, double& q) {
    for (int i = threadIdx.x; i < 2 * FA_ROWS * FA_CH; i += 256) {
        const int ch = i & (FA_CH - 1), r = (i / FA_CH) & (FA_ROWS - 1), w = i / (FA_ROWS * FA_CH);
        part[w][r][ch] = (r < rows && c0 + ch < C) ? slab[((int64_t)r * 2 + w) * C + c0 + ch] : 0.f;
    }
    __syncthreads();
    s = 0.0; q = 0.0;
    if (threadIdx.x < FA_CH) {
        const int ch = threadIdx.x;
        double a[32], b[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            a[i] = (0.0 + (double)part[0][i][ch]) + (0.0 + (double)part[0][i + 32][ch]);
            b[i] = (0.0 + (double)part[1][i][ch]) + (0.0 + (double)part[1][i + 32][ch]);
        }
#pragma unroll
        for (int half = 16; half > 0; half >>= 1)
#pragma unroll
            for (int i = 0; i < half; ++i) { a[i] += a[i + half]; b[i] += b[i + half]; }
        s = a[0]; q = b[0];
    }
}

// ---- forward: statistics finalize (bn_stats_finalize_kernel) + y = relu?((z - mean) * scale + beta + res) ------------------
template <bool B16>
__global__ __launch_bounds__(256) void bn_finapply_kernel(const float* __restrict__ slab, int rows, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                                          float momentum, float eps, float* mean, float* invstd, float* scale,
                                                          float* shift, const float* __restrict__ pivot,
                                                          const void* __restrict__ z, const void* __restrict__ res, void* __restrict__ y,
                                                          int M, int rows_per_wg, int relu, uint8_t* __restrict__ bits) {
    typedef Vec<B16> V;
    __shared__ float part[2][FA_ROWS][FA_CH];
    __shared__ float vec[3][FA_CH];                    // mean, scale, beta of the strip
    const int c0 = blockIdx.x * FA_CH;
    if (num_batches_tracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
    double s, q;
    strip_totals(slab, rows, C, c0, part, s, q);
    if (threadIdx.x < FA_CH) {
        const int c = c0 + threadIdx.x;
        float mu_f = 0.f, sc_f = 0.f, be_f = 0.f;
        if (c < C) {
            const double md = s / count;                       // mean of (x - pivot)
            double var = q / count - md * md;
            var = var > 0.0 ? var : 0.0;
            const double mu = md + (pivot ? (double)pivot[c] : 0.0);
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
            mu_f = (float)mu; sc_f = g * is; be_f = b;
            if (blockIdx.y == 0) {                             // one workgroup per strip publishes (the backward reads these)
                mean[c] = mu_f;
                invstd[c] = is;
                scale[c] = sc_f;
                shift[c] = b - mu_f * g * is;
                if (running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu_f;
                    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
                }
            }
        }
        vec[0][threadIdx.x] = mu_f; vec[1][threadIdx.x] = sc_f; vec[2][threadIdx.x] = be_f;
    }
    __syncthreads();
    constexpr int TPR = FA_CH / V::N, RPP = 256 / TPR;          // threads per row, rows per pass
    const int cg = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    if (c0 + cg * V::N >= C) return;
    typename V::F mu, sc, be;
#pragma unroll
    for (int e = 0; e < V::N; ++e) { mu[e] = vec[0][cg * V::N + e]; sc[e] = vec[1][cg * V::N + e]; be[e] = vec[2][cg * V::N + e]; }
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    const int CV = C / V::N;
    for (int r = r0 + rl; r < r1; r += RPP) {
        const int64_t i = (int64_t)r * CV + (c0 / V::N) + cg;
        typename V::F v = (V::ld(z, i) - mu) * sc;
        if (beta) v += be;
        if (res) v += V::ld(res, i);
        if (relu) {
#pragma unroll
            for (int e = 0; e < V::N; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        V::st(y, i, v);
        if (bits) {
            uint32_t mk = 0;
#pragma unroll
            for (int e = 0; e < V::N; ++e) mk |= (uint32_t)(V::stored(v[e]) > 0.f) << e;
            bits[i] = (uint8_t)mk;
        }
    }
}

// ---- backward: bn_bwd_finalize_kernel + dz = gamma * invstd * (g - mean_g - xhat * mean_gx) -------------------------------
template <bool B16>
__global__ __launch_bounds__(256) void bn_bwd_finapply_kernel(const float* __restrict__ slab, int rows, int C, double count,
                                                              float* dgamma, float* dbeta,
                                                              const void* __restrict__ dy, const void* __restrict__ z,
                                                              const void* __restrict__ act, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              void* __restrict__ dz, int M, int rows_per_wg, void* gres,
                                                              int gres_accumulate, const float* __restrict__ mscale,
                                                              const float* __restrict__ mbeta, const uint8_t* __restrict__ bits) {
    typedef Vec<B16> V;
    __shared__ float part[2][FA_ROWS][FA_CH];
    __shared__ float vec[2][FA_CH];                    // coef[0] = sum g / M, coef[1] = sum g * xhat / M
    const int c0 = blockIdx.x * FA_CH;
    double s, q;
    strip_totals(slab, rows, C, c0, part, s, q);
    if (threadIdx.x < FA_CH) {
        const int c = c0 + threadIdx.x;
        float k0 = 0.f, k1 = 0.f;
        if (c < C) {
            if (blockIdx.y == 0) {
                if (dbeta) dbeta[c] += (float)s;
                if (dgamma) dgamma[c] += (float)q;
            }
            k0 = (float)(s / count);
            k1 = (float)(q / count);
        }
        vec[0][threadIdx.x] = k0; vec[1][threadIdx.x] = k1;
    }
    __syncthreads();
    constexpr int TPR = FA_CH / V::N, RPP = 256 / TPR;
    const int cg = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int cc = c0 + cg * V::N;
    if (cc >= C) return;
    typename V::F mu, is, gm, k0, k1, ms, mb;
#pragma unroll
    for (int e = 0; e < V::N; ++e) {
        mu[e] = mean[cc + e]; is[e] = invstd[cc + e];
        gm[e] = gamma ? is[e] * gamma[cc + e] : is[e];
        k0[e] = vec[0][cg * V::N + e]; k1[e] = vec[1][cg * V::N + e];
        ms[e] = mscale ? mscale[cc + e] : 0.f;
        mb[e] = mbeta ? mbeta[cc + e] : 0.f;
    }
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    const int CV = C / V::N;
    for (int r = r0 + rl; r < r1; r += RPP) {
        const int64_t i = (int64_t)r * CV + (c0 / V::N) + cg;
        typename V::F g = V::ld(dy, i);
        const typename V::F zc = V::ld(z, i) - mu;
        if (bits) {
            const uint32_t mk = bits[i];
#pragma unroll
            for (int e = 0; e < V::N; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
        } else if (act) {
            const typename V::F a = V::ld(act, i);
#pragma unroll
            for (int e = 0; e < V::N; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
        } else if (mscale) {
            typename V::F t = zc * ms;
            if (mbeta) t += mb;
#pragma unroll
            for (int e = 0; e < V::N; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
        }
        if (gres) {
            typename V::F rr = g;
            if (gres_accumulate) rr += V::ld(gres, i);
            V::st(gres, i, rr);
        }
        V::st(dz, i, gm * (g - k0 - (zc * is) * k1));
    }
}

// rows per workgroup: about 1024 workgroups per launch, whole passes of the row loop
inline int rows_per_wg_for(int M, int C, int rpp) {
    const int strips = C / FA_CH;
    int blocks = 1024 / (strips > 0 ? strips : 1);
    if (blocks < 1) blocks = 1;
    int per = (M + blocks - 1) / blocks;
    per = (per + rpp - 1) / rpp * rpp;
    return per < rpp ? rpp : per;
}

// OFF by default: measured slower than the separate launches (same box, three pairs: fp32 52.74 -> 53.6-53.85 ms, bf16s 17.52 ->
// 17.9-18.0): the strip-structured pass streams worse than the flat one (138-148 VGPRs for the fp64 trees, 33 KiB of LDS, a
// 3-5 us prologue in each of ~1000 workgroups) and that costs more than the 6 us finalize launch it removes -- the same
// verdict as the gated form of round 3, for another reason.  GRL_BN_FINAPPLY=1 / grl_bn_finalize_apply_mode(1) keep it testable.
bool g_finapply_on = [] { const char* e = getenv("GRL_BN_FINAPPLY"); return e && atoi(e) != 0; }();

}  // namespace

// eligibility of the fused finalize + apply form (forward and backward)
bool grl_bn_finapply_takes(int rows, int C) { return g_finapply_on && rows > 0 && rows <= FA_ROWS && C % FA_CH == 0; }

int grl_launch_bn_bwd_finapply(int b16, const float* slab, int rows, int C, double count, float* dgamma, float* dbeta, const void* dy,
                               const void* z, const void* act, const float* mean, const float* invstd, const float* gamma, void* dz,
                               int M, void* gres, int gres_accumulate, const float* mscale, const float* mbeta, const uint8_t* bits,
                               hipStream_t s) {
    if (b16) {
        const int per = rows_per_wg_for(M, C, 32);
        hipLaunchKernelGGL(bn_bwd_finapply_kernel<true>, dim3(C / FA_CH, grl_ceil_div(M, per)), dim3(256), 0, s, slab, rows, C, count, dgamma,
                           dbeta, dy, z, act, mean, invstd, gamma, dz, M, per, gres, gres_accumulate, mscale, mbeta, bits);
    } else {
        const int per = rows_per_wg_for(M, C, 16);
        hipLaunchKernelGGL(bn_bwd_finapply_kernel<false>, dim3(C / FA_CH, grl_ceil_div(M, per)), dim3(256), 0, s, slab, rows, C, count, dgamma,
                           dbeta, dy, z, act, mean, invstd, gamma, dz, M, per, gres, gres_accumulate, mscale, mbeta, bits);
    }
    return grl_check_launch("bn_bwd_finapply");
}

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

static int finalize_apply(int b16, const float* slab, int rows, int C, int64_t count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                          float* mean, float* invstd, float* scale, float* shift, const float* pivot, const void* z, const void* res,
                          void* y, int M, int relu, uint8_t* relu_bits, void* stream) {
    GRL_REQUIRE(slab && mean && invstd && scale && shift && z && y && rows > 0 && M > 0 && count > 0, "bn_finalize_apply: bad args");
    GRL_REQUIRE(rows <= FA_ROWS && C % FA_CH == 0, "bn_finalize_apply: needs rows <= 64 and C % 64 == 0 (grl_bn_finalize_apply_takes)");
    GRL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_apply: running stats come together");
    GRL_REQUIRE((((uintptr_t)z | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "bn_finalize_apply: 16-byte aligned tensors");
    hipStream_t s = (hipStream_t)stream;
    if (b16) {
        const int per = rows_per_wg_for(M, C, 32);
        hipLaunchKernelGGL(bn_finapply_kernel<true>, dim3(C / FA_CH, grl_ceil_div(M, per)), dim3(256), 0, s, slab, rows, C, (double)count, gamma,
                           beta, running_mean, running_var, num_batches_tracked, momentum, eps, mean, invstd, scale, shift, pivot, z, res,
                           y, M, per, relu, relu_bits);
    } else {
        const int per = rows_per_wg_for(M, C, 16);
        hipLaunchKernelGGL(bn_finapply_kernel<false>, dim3(C / FA_CH, grl_ceil_div(M, per)), dim3(256), 0, s, slab, rows, C, (double)count, gamma,
                           beta, running_mean, running_var, num_batches_tracked, momentum, eps, mean, invstd, scale, shift, pivot, z, res,
                           y, M, per, relu, relu_bits);
    }
    return grl_check_launch("grl_bn_finalize_apply");
}

extern "C" int grl_bn_finalize_apply_takes(int rows, int C) { return grl_bn_finapply_takes(rows, C) ? 1 : 0; }
// test hook: on = 0 / 1 switches the fused form off / on for the process (backward entry points included), -1 only queries;
// returns the previous setting
extern "C" int grl_bn_finalize_apply_mode(int on) {
    const int was = g_finapply_on ? 1 : 0;
    if (on >= 0) g_finapply_on = on != 0;
    return was;
}

extern "C" int grl_bn_finalize_apply(const float* slab, int rows, int C, int64_t count, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                     float* mean, float* invstd, float* scale, float* shift, const float* pivot, const float* z,
                                     const float* res, float* y, int M, int relu, uint8_t* relu_bits, void* stream) {
    return finalize_apply(0, slab, rows, C, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, mean, invstd,
                          scale, shift, pivot, z, res, y, M, relu, relu_bits, stream);
}

extern "C" int grl_bn_finalize_apply_bf16(const float* slab, int rows, int C, int64_t count, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                          float* mean, float* invstd, float* scale, float* shift, const float* pivot, const void* z,
                                          const void* res, void* y, int M, int relu, uint8_t* relu_bits, void* stream) {
    return finalize_apply(1, slab, rows, C, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, mean, invstd,
                          scale, shift, pivot, z, res, y, M, relu, relu_bits, stream);
}
