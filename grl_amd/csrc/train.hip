// Train-mode kernels of the GRL path for gfx950: batch-statistics BatchNorm
// (forward finalize/apply, backward reduce/apply), ReLU/max-pool/gating backward,
// weight-gradient GEMM (reduction over pixels, split over workgroups, deterministic
// slab reduction) and the layout helpers the data-gradient path needs.
//
// Reference semantics: torch.nn.BatchNorm{1,2}d in training mode as used by
// reid/models/resnets1.py:76-91, basebranch.py:38-50, grl_model.py:71-83,203-226,
// Siamese.py:84-94,135-139 (biased variance for normalisation, unbiased variance and
// momentum 0.1 for the running estimate), and autograd's backward of those modules
// (reid/train/trainer.py:54).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CHUNK = 128;     // rows per partial of the column reductions

// ---------------------------------------------------------------------------------
// column statistics of x[M][C] (row stride ld): slab[chunk][0][c] = sum, [1][c] = sum sq of
// (x - pivot[c]).  The pivot (row 0 of x for BatchNorm statistics, none for plain column sums)
// keeps E[d^2] - E[d]^2 well conditioned when a column's mean dwarfs its spread -- BatchNorm1d
// over a handful of rows, where fp32 sums of squares would lose the variance.
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x,
                                                        float* __restrict__ slab, int M, int C,
                                                        int ld, const float* __restrict__ pivot) {
    __shared__ f32x4 red[2][4][64];
    const int chunk = blockIdx.y, c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
    if (c < C) {
        const int r1 = min(M, (chunk + 1) * CHUNK);
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (pivot) pv = *reinterpret_cast<const f32x4*>(pivot + c);
#pragma unroll 8
        for (int r = chunk * CHUNK + wave; r < r1; r += 4) {       // (8 loads in flight: on the small BatchNorm1d inputs this loop is pure latency)
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)r * ld + c) - pv;
            s += v; q += v * v;
        }
    }
    red[0][wave][lane] = s; red[1][wave][lane] = q;
    __syncthreads();
    if (wave == 0 && c < C) {
        s = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        q = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        *reinterpret_cast<f32x4*>(slab + ((int64_t)chunk * 2 + 0) * C + c) = s;
        *reinterpret_cast<f32x4*>(slab + ((int64_t)chunk * 2 + 1) * C + c) = q;
    }
}

// out[c] (+)= sum_r slab[r*stride + c]      (64 channels x 16 row groups per workgroup)
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* __restrict__ slab, int rows,
                                                        int64_t stride, int C,
                                                        float* __restrict__ out, int accumulate) {
    __shared__ double sh[16 * 64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double s = 0.0;
    if (c < C)
        for (int r = ry; r < rows; r += 16) s += slab[(int64_t)r * stride + c];
    sh[ry * 64 + cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        s = 0.0;
        for (int g = 0; g < 16; ++g) s += sh[g * 64 + cx];
        out[c] = (accumulate ? out[c] : 0.f) + (float)s;
    }
}

// Sum the [rows][2][C] partial slab for FCH channels with FRG row groups (1024 threads), fp64 accumulation in a
// fixed order (a thread over its rows r = ry, ry + FRG, ..., then the row groups by a fixed binary tree); returns the two
// totals to the threads of row group 0.  16 channels x 64 row groups (round 3; it was 64 x 16): the 64-channel layers
// have up to 2048 slab rows (the stem 8192) and with 64 channels per workgroup ONE workgroup walked them, 128 dependent
// fp64 adds per thread -- ~9 us per launch, 176 launches per training step.
// (FCH is a launch parameter: 16, or 4 where 16 would leave fewer than 32 workgroups -- C <= 256, the layers with the
// most slab rows; FCH * FRG = 1024 threads either way.)
inline int finalize_fch(int C) { return C >= 512 ? 16 : 4; }
__device__ __forceinline__ void slab_totals(const float* __restrict__ slab, int rows, int C, int c,
                                            double* sh /*[2][FRG][FCH]*/, double& s, double& q, const int FCH) {
    const int FRG = 1024 / FCH;
    const int cx = threadIdx.x % FCH, ry = threadIdx.x / FCH;
    s = 0.0; q = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int r = ry; r < rows; r += FRG) {
            s += slab[((int64_t)r * 2 + 0) * C + c];
            q += slab[((int64_t)r * 2 + 1) * C + c];
        }
    }
    sh[ry * FCH + cx] = s;
    sh[FRG * FCH + ry * FCH + cx] = q;
    __syncthreads();
    // the row groups' partials summed by a fixed binary tree (FRG = 64 or 256: 6 or 8 levels).  The serial loop over
    // the groups that stood here was a chain of up to 256 dependent LDS reads on a handful of threads -- 3 to 7 us of
    // the ~9 us these 176 launches per step took.
    for (int half = FRG >> 1; half > 0; half >>= 1) {
        if (ry < half) {
            sh[ry * FCH + cx] += sh[(ry + half) * FCH + cx];
            sh[FRG * FCH + ry * FCH + cx] += sh[FRG * FCH + (ry + half) * FCH + cx];
        }
        __syncthreads();
    }
    if (ry == 0) { s = sh[cx]; q = sh[FRG * FCH + cx]; }
}

// BN forward finalize: batch mean / biased var from the partial slab, running-stat
// update (PyTorch: running = (1-m)*running + m*stat, unbiased var for the running
// estimate), folded scale/shift for the apply pass.
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(
    const float* __restrict__ slab, int rows, int C, double count, const float* gamma,
    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
    float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
    const float* __restrict__ pivot, const int FCH) {
    __shared__ double sh[2 * 1024];
    const int c = blockIdx.x * FCH + (threadIdx.x % FCH);
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
    double s, q;
    slab_totals(slab, rows, C, c, sh, s, q, FCH);
    if (threadIdx.x >= FCH || c >= C) return;
    const double md = s / count;                       // mean of (x - pivot)
    double var = q / count - md * md;
    var = var > 0.0 ? var : 0.0;
    const double mu = md + (pivot ? (double)pivot[c] : 0.0);
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * is;
    shift[c] = b - (float)mu * g * is;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// y = relu?(z*scale + shift + res)
__global__ void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ res,
                                float* __restrict__ y, int C4, int64_t total4, int relu) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = reinterpret_cast<const f32x4*>(z)[i];
        v = v * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
        if (res) v += reinterpret_cast<const f32x4*>(res)[i];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// y = relu?((z - mean)*scale + beta + res): the train-mode form.  Centering BEFORE the multiply
// matters for BatchNorm1d over a few similar rows (|z - mean| << |mean|): the folded
// z*scale + (beta - mean*scale) of the eval path would cancel there and lose ~|mean|/|z - mean| of
// the fp32 significand; torch's train-mode kernel centres first too.
__global__ void bn_apply_centered_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                         const float* __restrict__ scale, const float* __restrict__ beta,
                                         const float* __restrict__ res, float* __restrict__ y, int C4,
                                         int64_t total4, int relu, uint8_t* __restrict__ bits) {
    // bits (may be NULL): one byte per four outputs, bit e = (y[e] > 0) -- the ReLU mask the backward of a RESIDUAL
    // BatchNorm reads instead of the whole activation (1/16 of its bytes; grl_bn_bwd's relu_bits)
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (step % C4 == 0) {
        // the launch's thread count is a multiple of the row's channel groups: a thread meets the SAME four channels
        // in every iteration -- their vectors live in registers (three 16-byte loads per 16 bytes of payload were
        // L1 traffic, not HBM traffic: the pass ran at 4.8-5.6 TB/s)
        const int c = (int)(i % C4) * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), sc = *reinterpret_cast<const f32x4*>(scale + c);
        f32x4 be = {0.f, 0.f, 0.f, 0.f};
        if (beta) be = *reinterpret_cast<const f32x4*>(beta + c);
        for (; i < total4; i += step) {
            f32x4 v = reinterpret_cast<const f32x4*>(z)[i] - mu;
            v = v * sc;
            if (beta) v += be;
            if (res) v += reinterpret_cast<const f32x4*>(res)[i];
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            reinterpret_cast<f32x4*>(y)[i] = v;
            if (bits) bits[i] = (uint8_t)((v[0] > 0.f) | (v[1] > 0.f) << 1 | (v[2] > 0.f) << 2 | (v[3] > 0.f) << 3);
        }
        return;
    }
    for (; i < total4; i += step) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = reinterpret_cast<const f32x4*>(z)[i] - *reinterpret_cast<const f32x4*>(mean + c);
        v = v * *reinterpret_cast<const f32x4*>(scale + c);
        if (beta) v += *reinterpret_cast<const f32x4*>(beta + c);
        if (res) v += reinterpret_cast<const f32x4*>(res)[i];
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
        if (bits) bits[i] = (uint8_t)((v[0] > 0.f) | (v[1] > 0.f) << 1 | (v[2] > 0.f) << 2 | (v[3] > 0.f) << 3);
    }
}

// BN backward, pass 1: g = dy * (act > 0) ; slab[chunk][0][c] = sum g, [1][c] = sum g*xhat
// LPR lanes cover one row's 4*LPR channels (LPR = 16 / 32 / 64 for C = 64 / 128 / >= 256), so a wave
// reads 64 / LPR rows per instruction and every lane works on the 64- and 128-channel layers too
// (the largest tensors of the step; one lane group per row left 48 of 64 lanes idle there).
// Fixed summation order: per lane over its rows, then across the lane groups of the wave
// (xor-shuffles), then the four waves through LDS.
// dy and gout carry NO __restrict__: in the in-place form (grl_bn_bwd with gres == dy) they are the same buffer -- every
// lane reads its element before it writes it.
template <int LPR>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* dy, const float* __restrict__ z, const float* __restrict__ act,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ slab,
    int M, int C, const float* __restrict__ mscale, const float* __restrict__ mbeta, float* gout,
    const uint8_t* __restrict__ bits) {
    constexpr int RPW = 64 / LPR;                     // rows per wave-instruction
    __shared__ f32x4 red[2][4][LPR];
    const int chunk = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane / LPR, cl = lane % LPR;
    const int c = blockIdx.x * (LPR * 4) + cl * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
    if (c < C) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        f32x4 ms = {0.f, 0.f, 0.f, 0.f}, mb = ms;
        if (mscale) ms = *reinterpret_cast<const f32x4*>(mscale + c);
        if (mbeta) mb = *reinterpret_cast<const f32x4*>(mbeta + c);
        const int r1 = min(M, (chunk + 1) * CHUNK);
        const int rend = min(r1, chunk * CHUNK + (wave + 1) * (CHUNK / 4));
#pragma unroll 4
        for (int r = chunk * CHUNK + wave * (CHUNK / 4) + sub; r < rend; r += RPW) {       // 12 loads in flight
            f32x4 g = *reinterpret_cast<const f32x4*>(dy + (int64_t)r * C + c);
            const f32x4 zc = *reinterpret_cast<const f32x4*>(z + (int64_t)r * C + c) - mu;
            if (bits) {                               // the forward's recorded (y > 0) bits instead of the activation
                const uint32_t mk = bits[((int64_t)r * C + c) >> 2];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
            } else if (act) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(act + (int64_t)r * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
            } else if (mscale) {                      // the forward's (z - mean) * scale + beta, term for term
                f32x4 t = zc * ms;
                if (mbeta) t += mb;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
            }
            const f32x4 xh = zc * is;
            s += g; q += g * xh;
            if (gout) *reinterpret_cast<f32x4*>(gout + (int64_t)r * C + c) = g;      // masked gradient, in place over dy
        }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {              // lane groups of the wave (same channels, other rows)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += __shfl_xor(s[e], o); q[e] += __shfl_xor(q[e], o); }
    }
    if (sub == 0) { red[0][wave][cl] = s; red[1][wave][cl] = q; }
    __syncthreads();
    if (wave == 0 && sub == 0 && c < C) {
        s = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        q = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        *reinterpret_cast<f32x4*>(slab + ((int64_t)chunk * 2 + 0) * C + c) = s;
        *reinterpret_cast<f32x4*>(slab + ((int64_t)chunk * 2 + 1) * C + c) = q;
    }
}

// finalize pass 1: dgamma += sum g*xhat, dbeta += sum g, coef[0][c] = sum g / M,
// coef[1][c] = sum g*xhat / M
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ slab,
                                                               int rows, int C, double count,
                                                               float* dgamma, float* dbeta,
                                                               float* __restrict__ coef, const int FCH) {
    __shared__ double sh[2 * 1024];
    const int c = blockIdx.x * FCH + (threadIdx.x % FCH);
    double s, q;
    slab_totals(slab, rows, C, c, sh, s, q, FCH);
    if (threadIdx.x >= FCH || c >= C) return;
    if (dbeta) dbeta[c] += (float)s;
    if (dgamma) dgamma[c] += (float)q;
    coef[c] = (float)(s / count);
    coef[C + c] = (float)(q / count);
}

// pass 2: dz = gamma*invstd * (g - mean_g - xhat * mean_gx)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                    const float* __restrict__ act, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ coef, float* __restrict__ dz, int C,
                                    int64_t total4, float* __restrict__ gres, int gres_accumulate,
                                    const float* __restrict__ mscale, const float* __restrict__ mbeta,
                                    const uint8_t* __restrict__ bits) {
    const int C4 = C >> 2;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (step % C4 == 0) {              // a thread's four channels are loop-invariant: five to seven vectors in registers
        const int c = (int)(i % C4) * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(coef + c), k1 = *reinterpret_cast<const f32x4*>(coef + C + c);
        f32x4 gm = is;
        if (gamma) gm = gm * *reinterpret_cast<const f32x4*>(gamma + c);
        f32x4 ms = {0.f, 0.f, 0.f, 0.f}, mb = ms;
        if (mscale) ms = *reinterpret_cast<const f32x4*>(mscale + c);
        if (mbeta) mb = *reinterpret_cast<const f32x4*>(mbeta + c);
        for (; i < total4; i += step) {
            f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
            const f32x4 zc = reinterpret_cast<const f32x4*>(z)[i] - mu;
            if (bits) {
                const uint32_t mk = bits[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
            } else if (act) {
                const f32x4 a = reinterpret_cast<const f32x4*>(act)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
            } else if (mscale) {
                f32x4 t = zc * ms;
                if (mbeta) t += mb;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
            }
            if (gres) {
                f32x4 r = g;
                if (gres_accumulate) r += reinterpret_cast<const f32x4*>(gres)[i];
                reinterpret_cast<f32x4*>(gres)[i] = r;
            }
            const f32x4 xh = zc * is;
            reinterpret_cast<f32x4*>(dz)[i] = gm * (g - k0 - xh * k1);
        }
        return;
    }
    for (; i < total4; i += step) {
        const int c = (int)(i % C4) * 4;
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        const f32x4 zc = reinterpret_cast<const f32x4*>(z)[i] - *reinterpret_cast<const f32x4*>(mean + c);
        if (bits) {
            const uint32_t mk = bits[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
        } else if (act) {
            const f32x4 a = reinterpret_cast<const f32x4*>(act)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
        } else if (mscale) {
            f32x4 t = zc * *reinterpret_cast<const f32x4*>(mscale + c);
            if (mbeta) t += *reinterpret_cast<const f32x4*>(mbeta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
        }
        if (gres) {        // the residual branch receives the same masked gradient (y = relu(bn(z) + res))
            f32x4 r = g;
            if (gres_accumulate) r += reinterpret_cast<const f32x4*>(gres)[i];
            reinterpret_cast<f32x4*>(gres)[i] = r;
        }
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 xh = zc * is;
        f32x4 gm = is;
        if (gamma) gm = gm * *reinterpret_cast<const f32x4*>(gamma + c);
        reinterpret_cast<f32x4*>(dz)[i] =
            gm * (g - *reinterpret_cast<const f32x4*>(coef + c) - xh * *reinterpret_cast<const f32x4*>(coef + C + c));
    }
}

// out = (accumulate ? out : 0) + dy * (act > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act,
                                float* __restrict__ out, int64_t total4, int accumulate) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        if (act) {
            const f32x4 a = reinterpret_cast<const f32x4*>(act)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
        }
        if (accumulate) g += reinterpret_cast<const f32x4*>(out)[i];
        reinterpret_cast<f32x4*>(out)[i] = g;
    }
}

// y = alpha*a + beta*b (b may be NULL)
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b,
                             float* __restrict__ y, float alpha, float beta, int64_t total4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(a)[i] * alpha;
        if (b) v += reinterpret_cast<const f32x4*>(b)[i] * beta;
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// dst[b*dstride + i] (+)= alpha * src[b*sstride + i]   (i < inner)
__global__ void axpy_strided_kernel(float* __restrict__ dst, int64_t dstride4,
                                    const float* __restrict__ src, int64_t sstride4, int64_t inner4,
                                    float alpha, int accumulate, int64_t total4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / inner4, r = i - b * inner4;
        f32x4 v = reinterpret_cast<const f32x4*>(src)[b * sstride4 + r] * alpha;
        f32x4* d = reinterpret_cast<f32x4*>(dst) + b * dstride4 + r;
        if (accumulate) v += *d;
        *d = v;
    }
}

// ---------------------------------------------------------------------------------
// layout helpers
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C,
                                 int ldx) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = x[(int64_t)(r0 + j) * ldx + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) y[(int64_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// ---------------------------------------------------------------------------------
// All weight re-layouts of one training step in ONE launch (round 3).  A step needs ~90 packed / transposed copies of
// the parameters (tap-major 3x3 packs, transposes and tap-flipped packs for the data gradients, the parity-class packs
// of the stride-2 data gradients) and, in bf16 storage, ~140 bf16 casts: 230 tiny launches, 15 % of all launches of a
// step that is host-bound in that mode.  Every one of them is a gather dst[j] = src[base + sum_d i_d * stride_d]
// over at most four destination dimensions (+ an optional bf16 rounding); 2-D transposes go through LDS tiles.  The
// table lives in device memory and is built once per model (train_engine.WeightPrep): sources are the parameters
// themselves, destinations persistent buffers.
__global__ __launch_bounds__(256) void weight_prep_kernel(const GrlPrepEntry* __restrict__ table) {
    __shared__ float tile[32][33];
    const GrlPrepEntry e = table[blockIdx.y];
    const float* __restrict__ src = e.src + e.base;
    float* const d32 = reinterpret_cast<float*>(e.dst);
    __bf16* const d16 = reinterpret_cast<__bf16*>(e.dst);
    if (e.tiled) {                     // dst[r][c] = src[r * strides[2] + c * strides[3]], strides[2] == 1 (a transpose)
        const int R = e.dims[2], Cn = e.dims[3];
        const int tr = (R + 31) / 32, tc = (Cn + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
            const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
            __syncthreads();
            for (int j = ty; j < 32; j += 8)            // read along r (the source's contiguous direction)
                if (c0 + j < Cn && r0 + tx < R) tile[j][tx] = src[(int64_t)(c0 + j) * e.strides[3] + (r0 + tx)];
            __syncthreads();
            for (int j = ty; j < 32; j += 8)
                if (r0 + j < R && c0 + tx < Cn) {
                    const float v = tile[tx][j];
                    const int64_t o = (int64_t)(r0 + j) * Cn + c0 + tx;
                    if (e.out_bf16) d16[o] = (__bf16)v; else d32[o] = v;
                }
        }
        return;
    }
    const int64_t n3 = e.dims[3], n2 = e.dims[2], n1 = e.dims[1];
    const int64_t total = (int64_t)e.dims[0] * n1 * n2 * n3;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i3 = j % n3, r3 = j / n3;
        const int64_t i2 = r3 % n2, r2 = r3 / n2;
        const int64_t i1 = r2 % n1, i0 = r2 / n1;
        const float v = src[i0 * e.strides[0] + i1 * e.strides[1] + i2 * e.strides[2] + i3 * e.strides[3]];
        if (e.out_bf16) d16[j] = (__bf16)v; else d32[j] = v;
    }
}

// data-gradient weights of a kxk conv: out[c][kk-1-t][n] = w[n][c][t]
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ out,
                                         int N, int C, int taps) {
    const int64_t total = (int64_t)N * C * taps;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = i % N;
        const int t = (i / N) % taps;
        const int c = i / ((int64_t)N * taps);
        out[i] = w[((int64_t)n * C + c) * taps + (taps - 1 - t)];
    }
}

// zero-stuffing for stride-2 data gradients: up[img][2oy][2ox][:] = dz[img][oy][ox][:], zero
// elsewhere; accumulate: up[img][2oy][2ox][:] += dz[...], every other element left as it is
__global__ void dilate2_kernel(const float* __restrict__ dz, float* __restrict__ up, int Ho, int Wo,
                               int H, int W, int C4, int64_t total4, int accumulate, int oy_off, int ox_off) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C4;
        int64_t r = i / C4;
        const int x = r % W; r /= W;
        const int y = r % H;
        const int img = r / H;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const bool hit = (x & 1) == ox_off && (y & 1) == oy_off && (y >> 1) < Ho && (x >> 1) < Wo;
        if (hit) v = reinterpret_cast<const f32x4*>(dz)[(((int64_t)img * Ho + (y >> 1)) * Wo + (x >> 1)) * C4 + c];
        if (accumulate) {                       // 1: add at the class pixels, 2: write only them
            if (!hit) continue;
            if (accumulate == 1) v += reinterpret_cast<const f32x4*>(up)[i];
        }
        reinterpret_cast<f32x4*>(up)[i] = v;
    }
}

// max-pool 3x3/s2/p1 backward, gather form (deterministic): an input pixel receives
// dy of every window in which it is the FIRST maximum in (ky,kx) scan order -- the
// element torch.nn.MaxPool2d records as argmax.
// One lane per 2x2 block of input pixels (x 4 channels): the block touches the four windows
// (a..a+1, b..b+1), whose union is a 5x5 input patch -- every window's first maximum is found
// once (9 compares) instead of once per input pixel it covers.
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                   float* __restrict__ dx, int H, int W, int C4, int64_t total4) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C = C4 * 4;
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;           // 2x2 input blocks
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int b = r % Wb; r /= Wb;
        const int a = r % Hb;
        const int img = r / Hb;
        const float* xi = x + (int64_t)img * H * W * C + c;
        f32x4 g[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) g[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
        // windows (oy, ox) in {a, a+1} x {b, b+1}; window o covers inputs 2o-1 .. 2o+1
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int oy = a + wy;
            if (oy >= Ho) continue;
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ox = b + wx;
                if (ox >= Wo) continue;
                // first maximum of the window in (ky, kx) scan order, per channel
                f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int by[4] = {-1, -1, -1, -1}, bx[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 - 1 + ky;
                    if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        const f32x4 o = *reinterpret_cast<const f32x4*>(xi + ((int64_t)yy * W + xx) * C);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (o[e] > best[e] || by[e] < 0) { best[e] = o[e]; by[e] = yy; bx[e] = xx; }
                    }
                }
                const f32x4 d = *reinterpret_cast<const f32x4*>(
                    dy + (((int64_t)img * Ho + oy) * Wo + ox) * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int uy = by[e] - 2 * a, ux = bx[e] - 2 * b;      // position inside this 2x2 block?
                    if (uy >= 0 && uy < 2 && ux >= 0 && ux < 2) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int v = 0; v < 2; ++v)
                                if (u == uy && v == ux) g[u][v][e] += d[e];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int iy = 2 * a + u, ix = 2 * b + v;
                if (iy < H && ix < W)
                    *reinterpret_cast<f32x4*>(dx + (((int64_t)img * H + iy) * W + ix) * C + c) = g[u][v];
            }
    }
}

// Stem tail of the TRAINING forward in one pass (round 3): p = maxpool3x3/s2/p1(relu((z - mean) * scale + beta)) with
// the window's first maximum recorded (idx = ky*3 + kx in scan order, torch's argmax rule) -- the post-ReLU map (268 MB
// per 32 x 4 step) is never written: the backward routes dp through idx (maxpool_bwd_idx_kernel) and recomputes the
// ReLU mask from z (grl_bn_bwd's mask_scale form).  Same three fp32 operations per element as bn_apply_centered_kernel.
__global__ void bn_relu_maxpool_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                       const float* __restrict__ scale, const float* __restrict__ beta,
                                       float* __restrict__ y, uint8_t* __restrict__ idx, int n, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C >> 2;
    const int64_t total = (int64_t)n * Ho * Wo * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int img = r / Ho;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), sc = *reinterpret_cast<const f32x4*>(scale + c);
        f32x4 be = {0.f, 0.f, 0.f, 0.f};
        if (beta) be = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint32_t pos = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(z + (((int64_t)img * H + iy) * W + ix) * C + c) - mu;
                v = v * sc;
                if (beta) v += be;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = v[e] > 0.f ? v[e] : 0.f;
                    if (a > m[e]) { m[e] = a; pos = (pos & ~(0xffu << (8 * e))) | ((uint32_t)(ky * 3 + kx) << (8 * e)); }
                }
            }
        }
        reinterpret_cast<f32x4*>(y)[i] = m;
        reinterpret_cast<uint32_t*>(idx)[i] = pos;
    }
}

// max-pool 3x3/s2/p1 backward from the recorded first-maximum positions: one lane per 2x2 block of input pixels
// (x 4 channels); the block meets the windows (a..a+1, b..b+1): reads 4 x (dy + idx) instead of a 5 x 5 input patch.
__global__ void maxpool_bwd_idx_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy,
                                       float* __restrict__ dx, int H, int W, int C4, int64_t total4) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C = C4 * 4;
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int b = r % Wb; r /= Wb;
        const int a = r % Hb;
        const int img = r / Hb;
        f32x4 g[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) g[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int oy = a + wy;
            if (oy >= Ho) continue;
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ox = b + wx;
                if (ox >= Wo) continue;
                const int64_t o = (((int64_t)img * Ho + oy) * Wo + ox) * C + c;
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
                const uint32_t pos = *reinterpret_cast<const uint32_t*>(idx + o);
#pragma unroll
                for (int u = wy; u < 2; ++u)              // window a+1 only reaches the block's second row (ky = 0)
#pragma unroll
                    for (int v = wx; v < 2; ++v) {
                        const uint32_t k = (uint32_t)((u - 2 * wy + 1) * 3 + (v - 2 * wx + 1));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (((pos >> (8 * e)) & 0xffu) == k) g[u][v][e] += d[e];
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int iy = 2 * a + u, ix = 2 * b + v;
                if (iy < H && ix < W)
                    *reinterpret_cast<f32x4*>(dx + (((int64_t)img * H + iy) * W + ix) * C + c) = g[u][v];
            }
    }
}

// stem im2col for the 7x7 weight gradient: col[m][k], k = (c*7+ky)*7+kx, padded to Kp
__global__ void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int H, int W,
                                   int Kp, int64_t total) {
    const int Ho = H / 2, Wo = W / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = i % Kp;
        int64_t m = i / Kp;
        const int ox = m % Wo; m /= Wo;
        const int oy = m % Ho;
        const int img = m / Ho;
        float v = 0.f;
        if (k < 147) {
            const int c = k / 49, ky = (k / 7) % 7, kx = k % 7;
            const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = x[(((int64_t)img * 3 + c) * H + iy) * W + ix];
        }
        col[i] = v;
    }
}

// ---------------------------------------------------------------------------------
// Weight gradient of the 7x7/s2 stem conv WITHOUT the im2col matrix (round 3): dW[o][k] = sum_m dz[m][o] * col[m][k],
// k = (c*7+ky)*7+kx (torch's [64][3][7][7]), col[m][k] = x[img][c][2oy-3+ky][2ox-3+kx].  The im2col matrix is
// 1048576 x 160 floats = 671 MB per 32 x 4 step, written by one launch and read back by the next (0.44 + 0.26 ms at
// the very end of the backward, where nothing overlaps it).  Here a persistent workgroup walks 8 x 16-pixel tiles:
// the tile's NCHW input patch (3 x 21 x 37, as in the forward stem kernel) and its dz rows go to LDS, the B operand
// col[m][k] is read out of the patch through a k -> offset table (ds_read_b32), and each of the eight waves accumulates one 32-channel
// half of the 64 x 160 result over its quarter of the tile's pixels in 5 v_mfma_f32_32x32x2_f32 accumulators.  Exact fp32
// products in every training math mode (dz may be stored as bf16: converted exactly).  The workgroups' partial
// results go to `ws` slabs, summed in slab order by wgrad_reduce_kernel: deterministic.
constexpr int SW_TH = 8, SW_TW = 16, SW_PH = 2 * SW_TH + 5, SW_PW = 2 * SW_TW + 5, SW_PWP = SW_PW + 1;
constexpr int SW_PATCH = 3 * SW_PH * SW_PWP;          // floats; cell SW_PATCH is a zero (padded k)
constexpr int SW_K = 160;
constexpr int SW_WGS = 512;                           // persistent workgroups = partial slabs

template <bool DZ16>
__global__ __launch_bounds__(512) void stem_wgrad_kernel(const float* __restrict__ x, const void* __restrict__ dzv,
                                                         float* __restrict__ ws, int n, int H, int W) {
    __shared__ __attribute__((aligned(16))) float patch[SW_PATCH + 4];
    __shared__ __attribute__((aligned(16))) float dzs[128 * 64];      // [pixel][channel ^ 32*(pixel & 1)]: see below
    const int Ho = H >> 1, Wo = W >> 1;
    const int tx_n = (Wo + SW_TW - 1) / SW_TW, ty_n = (Ho + SW_TH - 1) / SW_TH;
    const int tiles = n * ty_n * tx_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // eight waves: wave = 2 * (pixel quarter) + (channel tile).  A wave accumulates its 32 output channels x 160 k over
    // its quarter of every tile's pixels: five accumulators (80 AGPRs), two waves per SIMD
    const int ot = wave & 1, pq = wave >> 1;
    // the lane's five B columns k = kt*32 + l31 -> offset into the patch (k >= 147: the zero cell)
    int koff[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
        const int k = kt * 32 + l31;
        koff[kt] = k < 147 ? ((k / 49) * SW_PH + (k / 7) % 7) * SW_PWP + k % 7 : -1;
    }
    f32x16 acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    if (tid < 4) patch[SW_PATCH + tid] = 0.f;
    constexpr int P_N = 3 * SW_PH * SW_PW, P_IT = (P_N + 511) / 512;
    float pv[P_IT];
    f32x4 dv[4];                                       // 128 pixels x 16 float4: item = tid + it*512 -> (pixel, quad)
    // a thread stages the SAME patch cells and dz quads of every tile: their coordinates are computed once
    int p_rq[P_IT], p_g[P_IT], p_l[P_IT];              // (row << 8 | col) inside the patch, global / LDS offsets
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int i = tid + it * 512;
        const int c = i / (SW_PH * SW_PW), r = (i / SW_PW) % SW_PH, q = i % SW_PW;
        p_rq[it] = i < P_N ? (r << 8 | q) : -1;
        p_g[it] = (c * H + r) * W + q;
        p_l[it] = (c * SW_PH + r) * SW_PWP + q;
    }
    // global -> registers for tile t (every load issued before anything waits)
    auto fetch = [&](const int t) {
        const int img = t / (ty_n * tx_n), tr = t - img * (ty_n * tx_n);
        const int oy0 = (tr / tx_n) * SW_TH, ox0 = (tr % tx_n) * SW_TW;
        const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
        const float* xi = x + (int64_t)img * 3 * H * W + (int64_t)iy0 * W + ix0;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int iy = iy0 + (p_rq[it] >> 8), ix = ix0 + (p_rq[it] & 255);
            const bool ok = p_rq[it] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            pv[it] = ok ? xi[p_g[it]] : 0.f;
        }
        const int64_t row0 = ((int64_t)img * Ho + oy0) * Wo + ox0;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + it * 512, m = i >> 4, q4 = (i & 15) * 4;
            const int my = m / SW_TW, mx = m % SW_TW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (oy0 + my < Ho && ox0 + mx < Wo) {
                const int64_t row = row0 + (int64_t)my * Wo + mx;
                if (DZ16) {
                    const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(dzv) + row * 64 + q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)h[e];
                } else {
                    v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dzv) + row * 64 + q4);
                }
            }
            dv[it] = v;
        }
    };
    if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        __syncthreads();                               // the previous tile's MFMAs are done with the LDS tiles
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (p_rq[it] >= 0) patch[p_l[it]] = pv[it];
        // dz rows: odd pixels are stored with their two 32-channel halves swapped, so the two half-waves of an A read
        // (pixels m, m+1; same channel tile) hit disjoint banks
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + it * 512, m = i >> 4, q4 = (i & 15) * 4;
            *reinterpret_cast<f32x4*>(dzs + m * 64 + (q4 ^ ((m & 1) << 5))) = dv[it];
        }
        __syncthreads();
        if (t + (int)gridDim.x < tiles) fetch(t + gridDim.x);      // the next tile's loads fly under this tile's MFMAs
        // pixels pq*32 .. pq*32+31, two per MFMA step (lane half lh takes pixel 2s + lh)
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            const int m = pq * 32 + 2 * sidx + lh;
            const int pb = (2 * (pq * 2 + (sidx >> 3))) * SW_PWP + 2 * ((2 * sidx + lh) & 15);
            const float a = dzs[m * 64 + ((ot * 32 + l31) ^ ((m & 1) << 5))];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                const float b = patch[koff[kt] < 0 ? SW_PATCH : pb + koff[kt]];
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kt], 0, 0, 0);
            }
        }
    }
    // the four pixel quarters summed in order through LDS (reusing the dz tile: 2 channel tiles x 3 x 4 KB per
    // accumulator), the quarter-0 waves write the workgroup's slab [64][160]
    float* red = dzs;
    float* out = ws + (int64_t)blockIdx.x * 64 * SW_K;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        __syncthreads();
        if (pq > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((ot * 3 + pq - 1) * 16 + r) * 64 + lane] = acc[b][r];
        }
        __syncthreads();
        if (pq == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[b][r];
                v += red[((ot * 3 + 0) * 16 + r) * 64 + lane];
                v += red[((ot * 3 + 1) * 16 + r) * 64 + lane];
                v += red[((ot * 3 + 2) * 16 + r) * 64 + lane];
                const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[o * SW_K + b * 32 + l31] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Weight-gradient GEMM:  dW[n][k] = sum_m dz[m][n] * Xg[m][k]   (reduction over pixels)
// Both operands are stored with the reduction index as the ROW, so a stage is a copy of
// 32 rows of dz (BM floats) and 32 (gathered) rows of X (BN floats) into LDS as
// [kr][out]; lane (i = l&31, h = l>>5) feeds v_mfma_f32_32x32x2_f32 with
// A = tile[2s+h][i] by ds_read_b32 (consecutive lanes -> consecutive banks).
// grid = (tiles, splits): each split reduces its own pixel range into its own slab.
struct WgradArgs {
    const float* dz; const float* x; float* slab;
    int M, N, K;                 // pixels, Cout, taps*C
    int ldz, ldx;
    int64_t slab_stride;         // N*K
    int chunk;                   // pixels per split (multiple of 32)
    int conv, H, W, C, Ho, Wo, kh, kw, stride, pad;
    int tiles_n, tile_mode;      // XCD-aware tile map (wgrad_tile_of)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// MATH 0: exact fp32 (above).  MATH 3 / 1 (128 x 128 tiles only): the staging pass splits every
// operand value into bf16 hi (+ lo = bf16(v - hi) for MATH 3) and writes [m][out] bf16 planes with
// 256-byte rows, the 16-byte chunk XOR-swizzled by ((m & 3) << 2) | ((m >> 2) & 3); the reduction index
// m is the ROW of both tiles, so the MFMA fragments (8 consecutive m of one output column) are read
// with the hardware transpose read `ds_read_b64_tr_b16` (4 rows x 16 columns per 16-lane group,
// conflict-free with that swizzle) and feed v_mfma_f32_32x32x16_bf16: lo*hi + hi*lo + hi*hi.
__device__ uint4 g_wgrad_zero_chunk;       // 16 zero bytes: LDS-DMA source of rows past the pixel range / columns past N or K / out-of-image taps
typedef __attribute__((address_space(3))) char* lds_cptr_t;
// LDS-DMA piece hipcc does not see (see gemm_f32.hip: a VISIBLE global_load_lds makes every later LDS read wait vmcnt(0));
// the loop waits for its pieces itself in front of the stage barrier
// ("m0" is on the clobber list: the statement overwrites it, and the compiler keeps its own LDS-DMA / indexing state there.
//  hipcc accepts the clobber with a -Winline-asm note about reserved registers, silenced for these statements only.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_hidden_v(const char* vaddr, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
#ifndef GRL_WGRAD_KO
#define GRL_WGRAD_KO 0      // timing-only knock-outs (1: no in-loop staging, 2: no slab store); wrong results
#endif
// XCD-aware tile map of the weight-gradient kernels (round 5).  Workgroups are dealt round-robin over the 8 XCDs by their
// linear id (b and b + 8 share one L2); with tile_k fastest the 32 tiles an XCD holds of one 16 x 16 pixel-range row were
// 16 tile_n x 2 tile_k = 18 operand panels through its L2 -- every XCD read ALL of dz.  Here XCD j gets a rectangle of
// the tile grid, (tiles_n / 2) x (tiles_k / 4) (or / 4 x / 2): 8 x 4 tiles = 12 panels.  Which tile a workgroup computes
// does not enter any sum: bit-identical.  `mode` 0: the plain map (gridDim.x not a multiple of 8, odd tile counts, or
// GRL_WGRAD_XCD=0).
__device__ __forceinline__ void wgrad_tile_of(int bid, int tiles_n, int tiles_k, int mode, int& tile_n, int& tile_k) {
    if (mode == 0) {
        tile_n = bid / tiles_k;
        tile_k = bid - tile_n * tiles_k;
        return;
    }
    const int gn = mode == 1 ? 2 : 4, gk = 8 / gn;             // XCD regions along n and k
    const int rn = tiles_n / gn, rk = tiles_k / gk;            // tiles per region
    const int xcd = bid & 7, slot = bid >> 3;                  // slot: 0 .. rn * rk - 1
    const int xn = xcd / gk, xk = xcd - xn * gk;
    const int sn = slot / rk, sk = slot - sn * rk;
    tile_n = xn * rn + sn;
    tile_k = xk * rk + sk;
}
static int wgrad_tile_mode(int tiles_n, int tiles_k) {
    static const bool on = [] { const char* e = getenv("GRL_WGRAD_XCD"); return !e || atoi(e) != 0; }();
    if (!on || tiles_n * tiles_k < 64) return 0;
    // the squarer rectangle first: regions of (tiles_n / gn) x (tiles_k / gk) tiles
    const bool m1 = tiles_n % 2 == 0 && tiles_k % 4 == 0, m2 = tiles_n % 4 == 0 && tiles_k % 2 == 0;
    if (m1 && m2) return (tiles_n / 2 + tiles_k / 4) <= (tiles_n / 4 + tiles_k / 2) ? 1 : 2;
    return m1 ? 1 : m2 ? 2 : 0;
}
template <int BM, int BN, bool CONV, int MATH = 0>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs p, const int tiles_k) {
    static_assert(MATH == 0 || (BM == 128 && BN == 128), "the bf16 datapaths are written for 128 x 128 tiles");
    constexpr int WTM = BM / 2, WTN = BN / 2, MT = WTM / 32, NT = WTN / 32;
    constexpr int A_ITEMS = BM / 32, B_ITEMS = BN / 32;   // float4 per thread per stage (32 rows)
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;         // threads per row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][32][BM]
    float* Bs = smem + 2 * 32 * BM;        // [2][32][BN]
    int tile_n, tile_k;
    wgrad_tile_of((int)blockIdx.x, p.tiles_n, tiles_k, p.tile_mode, tile_n, tile_k);
    const int n0 = tile_n * BM, k0 = tile_k * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_begin = blockIdx.y * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    int tap = 0, c0 = k0, ky = 0, kx = 0;
    if (CONV) { tap = k0 / p.C; c0 = k0 - tap * p.C; ky = tap / p.kw; kx = tap - ky * p.kw; }

    f32x4 areg[A_ITEMS], breg[B_ITEMS];
    // conv: each staged X row is an output pixel (img, oy, ox); a stage advances every row by 32
    // pixels, so the decomposition is carried incrementally instead of two divisions per row
    // and stage (they sat in front of every stage's MFMAs)
    int pimg[B_ITEMS], poy[B_ITEMS], pox[B_ITEMS];
    const int adv_y = 32 / p.Wo, adv_x = 32 - adv_y * p.Wo;
    const bool small_img = CONV && p.Ho * p.Wo < 64;
    if (CONV) {
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) {
            const int e = tid + 256 * i, row = e / B_TPR;
            const int m = m_begin + row, hw = p.Ho * p.Wo;
            pimg[i] = m / hw;
            const int rem = m - pimg[i] * hw;
            poy[i] = rem / p.Wo;
            pox[i] = rem - poy[i] * p.Wo;
        }
    }
    // The loads are UNCONDITIONAL (an out-of-range item reads the tensor's first element instead) and the zeroing moves to
    // the item's LDS store: with a branch around every load hipcc cannot count them and waits vmcnt(0) in front of the first
    // ds_write of the stage; counted, each store waits for its own load only.
    bool aok[A_ITEMS], bok[B_ITEMS];
    auto load_stage = [&](int m0) {
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            const int e = tid + 256 * i, row = e / A_TPR, col = (e - row * A_TPR) * 4;
            const int m = m0 + row;
            aok[i] = m < m_end && n0 + col < p.N;
            const float* src = aok[i] ? p.dz + (int64_t)m * p.ldz + n0 + col : p.dz;
            areg[i] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) {
            const int e = tid + 256 * i, row = e / B_TPR, col = (e - row * B_TPR) * 4;
            const int m = m0 + row;
            const float* src = p.x;
            if (CONV) {
                const int iy = poy[i] * p.stride - p.pad + ky, ix = pox[i] * p.stride - p.pad + kx;
                bok[i] = m < m_end && k0 + col < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                if (bok[i]) src = p.x + (((int64_t)pimg[i] * p.H + iy) * p.W + ix) * p.C + c0 + col;
                // advance this row's pixel by the 32 rows of a stage (stages are loaded in order)
                pox[i] += adv_x;
                if (pox[i] >= p.Wo) { pox[i] -= p.Wo; ++poy[i]; }
                poy[i] += adv_y;
                while (poy[i] >= p.Ho) { poy[i] -= p.Ho; ++pimg[i]; }
            } else {
                bok[i] = m < m_end && k0 + col < p.K;
                if (bok[i]) src = p.x + (int64_t)m * p.ldx + k0 + col;
            }
            breg[i] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto item_a = [&](int i) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; return aok[i] ? areg[i] : z; };
    auto item_b = [&](int i) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; return bok[i] ? breg[i] : z; };
    // MATH == 0: the stage goes global -> LDS by LDS-DMA, item by item (item = one 16-byte element per thread, the same
    // thread -> element map as the register path: element e = tid + 256 it lands at byte 16 e of its operand's stage, so
    // a wave's piece is 1 KB of consecutive LDS); whatever must read as zero comes from a zero chunk.
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_cptr_t) reinterpret_cast<char*>(smem);
    const char* zsrc = reinterpret_cast<const char*>(&g_wgrad_zero_chunk);
    asm volatile("" : "+v"(zsrc));      // keep the pointer in registers: rematerialised, it is a GOT load (and an lgkmcnt(0) wait) per item
    // Dense operands: every item carries its source pointer and advances it by 32 rows per stage (a pointer that must read
    // zeros for its column stays on the zero chunk with stride 0), so a stage costs one 64-bit add per item -- plus, in
    // the one stage of a pixel range that is not full, a compare and a select.  Counters (tools/wgrad_vs_gemm_probe.py)
    // showed what separates this kernel from the forward GEMM on the same product: not TLB or L2 behaviour but 1.7x the
    // VALU-active cycles and 2.1x the LDS instructions -- per-item 64-bit multiply-adds, a v_add per fragment read.
    const char* pa[A_ITEMS];
    const char* pb[B_ITEMS];
    unsigned sa[A_ITEMS], sb[B_ITEMS];
    int ra[A_ITEMS], rb[B_ITEMS];
#pragma unroll
    for (int i = 0; i < A_ITEMS; ++i) {
        const int e = tid + 256 * i, row = e / A_TPR, col = (e - row * A_TPR) * 4;
        const bool ok = n0 + col < p.N;
        ra[i] = row;
        pa[i] = ok ? reinterpret_cast<const char*>(p.dz + (int64_t)(m_begin + row) * p.ldz + n0 + col) : zsrc;
        sa[i] = ok ? (unsigned)(32 * p.ldz * 4) : 0u;
    }
    const bool lin = CONV && !small_img && p.stride == 1 && p.Ho == p.H && p.Wo == p.W && 32 % p.Wo == 0;
#pragma unroll
    for (int i = 0; i < B_ITEMS; ++i) {
        const int e = tid + 256 * i, row = e / B_TPR, col = (e - row * B_TPR) * 4;
        rb[i] = row;
        if (CONV) {
            // lin: pox is this item's column for every stage; its tap column must lie inside the image
            const bool ok = lin && k0 + col < p.K && (unsigned)(pox[i] + kx - p.pad) < (unsigned)p.W;
            const int64_t pix = (int64_t)(m_begin + row) + (int64_t)(ky - p.pad) * p.W + (kx - p.pad);
            pb[i] = ok ? reinterpret_cast<const char*>(p.x + pix * p.C + c0 + col) : zsrc;
            sb[i] = ok ? (unsigned)(32 * p.C * 4) : 0u;
        } else {
            const bool ok = k0 + col < p.K;
            pb[i] = ok ? reinterpret_cast<const char*>(p.x + (int64_t)(m_begin + row) * p.ldx + k0 + col) : zsrc;
            sb[i] = ok ? (unsigned)(32 * p.ldx * 4) : 0u;
        }
    }
    // (items are requested stage by stage in order, starting at m_begin: the pointers are always those of stage `m0`)
    auto dma_item = [&](int buf, int m0, int it, bool full) {
        const char* src;
        unsigned dst;
        if (it < A_ITEMS) {
            src = (full || m0 + ra[it] < m_end) ? pa[it] : zsrc;
            pa[it] += sa[it];
            dst = lds0 + (unsigned)((buf * 32 * BM) * 4 + wave_u * 1024 + 4096 * it);
        } else {
            const int i = it - A_ITEMS;
            if (CONV && lin) {
                // same-size stride-1 window: the tap's input pixel is the output pixel shifted by a constant, so the source
                // is LINEAR in the pixel index (pointer + 32 rows per stage); the column test is fixed per item (Wo divides
                // 32), only the row test follows the pixel's oy
                src = ((unsigned)(poy[i] + ky - p.pad) < (unsigned)p.H && (full || m0 + rb[i] < m_end)) ? pb[i] : zsrc;
                pb[i] += sb[i];
                poy[i] += adv_y;
                if (poy[i] >= p.Ho) poy[i] -= p.Ho;
            } else if (CONV) {
                src = zsrc;
                const int e = tid + 256 * i, row = e / B_TPR, col = (e - row * B_TPR) * 4;
                const int m = m0 + row;
                const int iy = poy[i] * p.stride - p.pad + ky, ix = pox[i] * p.stride - p.pad + kx;
                if (m < m_end && k0 + col < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    src = reinterpret_cast<const char*>(p.x + (((int64_t)pimg[i] * p.H + iy) * p.W + ix) * p.C + c0 + col);
                pox[i] += adv_x;                                   // this row's pixel 32 rows on (stages are requested in order)
                if (pox[i] >= p.Wo) { pox[i] -= p.Wo; ++poy[i]; }
                poy[i] += adv_y;
                if (poy[i] >= p.Ho) { poy[i] -= p.Ho; ++pimg[i]; }
                if (small_img)                                     // (images of fewer than 64 pixels: 32 rows can span several)
                    while (poy[i] >= p.Ho) { poy[i] -= p.Ho; ++pimg[i]; }
            } else {
                src = (full || m0 + rb[i] < m_end) ? pb[i] : zsrc;
                pb[i] += sb[i];
            }
            dst = lds0 + (unsigned)((2 * 32 * BM + buf * 32 * BN) * 4 + wave_u * 1024 + 4096 * i);
        }
        dma16_hidden_v(src, dst);
    };
    constexpr int PLANE = 32 * 256;                       // bytes of one bf16 plane: 32 rows x 128 columns
    constexpr int NPL = MATH == 3 ? 2 : 1;                // planes per operand (hi, lo)
    char* const sm8 = reinterpret_cast<char*>(smem);      // MATH != 0: [2 buf][A hi, (A lo), B hi, (B lo)][PLANE]
    auto split_store = [&](char* plane0, int e, const f32x4 v) {
        const int row = e >> 5, col = (e & 31) * 4;       // 32 threads per 128-wide row
        const int off = 256 * row + 16 * ((col >> 3) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 2 * (col & 7);
        bf16x4 hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) hi[k] = (__bf16)v[k];
        *reinterpret_cast<bf16x4*>(plane0 + off) = hi;
        if (MATH == 3) {
            bf16x4 lo;
#pragma unroll
            for (int k = 0; k < 4; ++k) lo[k] = (__bf16)(v[k] - (float)hi[k]);
            *reinterpret_cast<bf16x4*>(plane0 + PLANE + off) = lo;
        }
    };
    auto store_stage = [&](int buf) {
        if constexpr (MATH == 0) {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<f32x4*>(As + buf * 32 * BM + e * 4) = item_a(i);
            }
#pragma unroll
            for (int i = 0; i < B_ITEMS; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<f32x4*>(Bs + buf * 32 * BN + e * 4) = item_b(i);
            }
        } else {
            char* const base = sm8 + buf * (2 * NPL * PLANE);
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) split_store(base, tid + 256 * i, item_a(i));
#pragma unroll
            for (int i = 0; i < B_ITEMS; ++i) split_store(base + NPL * PLANE, tid + 256 * i, item_b(i));
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nst = (m_end - m_begin + 31) / 32;
    const int frow = lane & 31, fhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    constexpr int FD = 4;                                        // MATH == 0: fragment ring, FD - 1 k-steps of LDS reads in flight
    float af[FD][MT], bf[FD][NT];
    // MATH == 0 tile -> column map: with two MFMA tiles per wave and operand, tile i takes the columns 2 c + i of the wave's
    // 64 (not c + 32 i), so a lane's two A (two B) values of a k-step are NEIGHBOURS in the linear LDS row: one ds_read_b64
    // at a compile-time offset instead of a ds_read2_b32 behind a v_add.  Only the labels of the accumulators change (the
    // slab store below un-permutes): every output still sums its pixels in the same order -- bit-identical.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto rdf = [&](int set, int buf, int s) {
        const float* Ab = As + buf * 32 * BM + wm * WTM + (2 * s + fhalf) * BM;
        const float* Bb = Bs + buf * 32 * BN + wn * WTN + (2 * s + fhalf) * BN;
        if constexpr (MT == 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(Ab + 2 * frow);
            af[set][0] = v[0];
            af[set][1] = v[1];
        } else {
            af[set][0] = Ab[frow];
        }
        if constexpr (NT == 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(Bb + 2 * frow);
            bf[set][0] = v[0];
            bf[set][1] = v[1];
        } else {
            bf[set][0] = Bb[frow];
        }
    };
    if (nst > 0) {
        if constexpr (MATH == 0) {
#pragma unroll
            for (int it = 0; it < A_ITEMS + B_ITEMS; ++it) dma_item(0, m_begin, it, m_begin + 32 <= m_end);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            load_stage(m_begin);
            store_stage(0);
        }
    }
    __syncthreads();
    if constexpr (MATH == 0)
        if (nst > 0) {
#pragma unroll
            for (int q = 0; q < FD - 1; ++q) rdf(q, 0, q);
        }
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if constexpr (MATH != 0)
            if (st + 1 < nst) load_stage(m_begin + (st + 1) * 32);
        if constexpr (MATH == 0) {
        // One stage = 16 k-steps of MT x NT MFMAs.  Fragments are double-buffered in registers ACROSS the stage boundary and
        // the next stage's LDS-DMA items go out two per k-step under the MFMAs of the first steps (their address arithmetic
        // -- the conv gather's pixel walk included -- interleaves with matrix work instead of sitting in front of it); the
        // stage barrier (after `vmcnt(0)`: this wave's items have landed) sits in front of the LAST step's MFMAs -- its
        // fragments have arrived, so nobody reads this buffer again -- and the next stage's first fragments are requested
        // right behind it, under those MFMAs.  History: hipcc's own schedule (read, lgkmcnt(0), 4 MFMAs per step) 66 %
        // MFMA busy; reads of step s+1 fenced in front of the MFMAs of step s, register staging 71 %; a timing-only build
        // without any staging (GRL_WGRAD_KO=1) runs 15 % (dense) to 24 % (conv) faster than that.  Same MFMA order:
        // bit-identical.
        constexpr int NI = A_ITEMS + B_ITEMS;
        // fragment ring: the reads of k-step s + FD - 1 go out in front of the MFMAs of step s (a k-step is only 4 MFMAs =
        // 256 cycles; one step of lookahead did not cover the LDS latency under load).  The last FD - 1 steps of a stage
        // read the NEXT buffer, so the stage barrier sits in front of step 16 - (FD - 1): by then every read of this
        // buffer has returned.
        constexpr int SB = 16 - (FD - 1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s == SB) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (s + FD - 1 < 16) rdf((s + FD - 1) % FD, buf, s + FD - 1);
            else rdf((s + FD - 1) % FD, buf ^ 1, s + FD - 1 - 16);       // (past the last stage: unused values, no branch)
            // (no branch around the items: past the last stage every row is out of range, the items copy the zero chunk into the
            // buffer nobody reads again -- a branch here makes hipcc drain lgkmcnt at its join, once per stage)
            if (!(GRL_WGRAD_KO & 1) && 2 * s < NI) {
                const int mnext = m_begin + (st + 1) * 32;
                const bool full = mnext + 32 <= m_end;
                dma_item(buf ^ 1, mnext, 2 * s, full);
                dma_item(buf ^ 1, mnext, 2 * s + 1, full);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % FD][i], bf[s % FD][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
            // transpose reads: 16-lane group g = lane >> 4 reads the 4-row x 16-column block
            // rows 16*ks + 8*(g >> 1) + 4*rd .. +3, columns tile + 16*(g & 1) .. +15; lane 4q + p of the
            // group supplies the address of row q, columns 4p .. 4p+3 and receives column (lane & 15)
            const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
            const unsigned sbase = lds_base + buf * (2 * NPL * PLANE);
            unsigned aaddr[MT][2], baddr[NT][2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int row = 8 * (g >> 1) + 4 * rd + q;
                const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int col = wm * WTM + i * 32 + 16 * (g & 1) + 4 * pp;
                    aaddr[i][rd] = sbase + 256 * row + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = wn * WTN + j * 32 + 16 * (g & 1) + 4 * pp;
                    baddr[j][rd] = sbase + NPL * PLANE + 256 * row + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
                }
            }
#define GRL_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GRL_FRAG(lo4, hi4) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7))
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {               // two k-steps of 16 rows per 32-row stage
                s16x4 ra[NPL][MT][2], rb[NPL][NT][2];
                // the hi planes' fragments are requested first and the hi x hi products start as soon as THEY have
                // arrived (counted wait), while the lo planes' reads are still in flight: half of the LDS latency of a
                // k-step used to sit in front of all twelve MFMAs (one lgkmcnt(0) after sixteen reads)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            if (ks == 0 && pl == 0) GRL_TR(ra[pl][i][rd], aaddr[i][rd], 0);
                            if (ks == 0 && pl == 1) GRL_TR(ra[pl][i][rd], aaddr[i][rd], PLANE);
                            if (ks == 1 && pl == 0) GRL_TR(ra[pl][i][rd], aaddr[i][rd], 4096);
                            if (ks == 1 && pl == 1) GRL_TR(ra[pl][i][rd], aaddr[i][rd], PLANE + 4096);
                        }
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            if (ks == 0 && pl == 0) GRL_TR(rb[pl][j][rd], baddr[j][rd], 0);
                            if (ks == 0 && pl == 1) GRL_TR(rb[pl][j][rd], baddr[j][rd], PLANE);
                            if (ks == 1 && pl == 0) GRL_TR(rb[pl][j][rd], baddr[j][rd], 4096);
                            if (ks == 1 && pl == 1) GRL_TR(rb[pl][j][rd], baddr[j][rd], PLANE + 4096);
                        }
                    }
                if (MATH == 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the 8 hi-plane reads (issued first)
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const bf16x8 ah = GRL_FRAG(ra[0][i][0], ra[0][i][1]), bh = GRL_FRAG(rb[0][j][0], rb[0][j][1]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                    }
                if (MATH == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const bf16x8 ah = GRL_FRAG(ra[0][i][0], ra[0][i][1]), bh = GRL_FRAG(rb[0][j][0], rb[0][j][1]);
                            const bf16x8 al = GRL_FRAG(ra[NPL - 1][i][0], ra[NPL - 1][i][1]);
                            const bf16x8 bl = GRL_FRAG(rb[NPL - 1][j][0], rb[NPL - 1][j][1]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef GRL_TR
#undef GRL_FRAG
        }
        if constexpr (MATH != 0) {
            if (st + 1 < nst) store_stage(buf ^ 1);
            __syncthreads();
        }
    }
    float* out = p.slab + (int64_t)blockIdx.y * p.slab_stride;
    const int col_l = lane & 31;
    if constexpr (MATH == 0) {
        // (tile i / j of a two-tile wave holds the columns 2 c + i / 2 c + j: see rdf)
        const bool pair_ok = NT == 2 && (p.K & 1) == 0 && (((uintptr_t)out & 7) == 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowt = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                const int n = n0 + wm * WTM + (MT == 2 ? 2 * rowt + i : rowt);
                if (n >= p.N) continue;
                const bool live = !(GRL_WGRAD_KO & 2) || acc[i][0][r] == 1.2345e-30f;
                if constexpr (NT == 2) {
                    const int k = k0 + wn * WTN + 2 * col_l;
                    if (pair_ok && k + 1 < p.K) {
                        if (live) *reinterpret_cast<f32x2*>(out + (int64_t)n * p.K + k) = f32x2{acc[i][0][r], acc[i][1][r]};
                    } else {
                        if (k < p.K && live) out[(int64_t)n * p.K + k] = acc[i][0][r];
                        if (k + 1 < p.K && live) out[(int64_t)n * p.K + k + 1] = acc[i][1][r];
                    }
                } else {
                    const int k = k0 + wn * WTN + col_l;
                    if (k < p.K && live) out[(int64_t)n * p.K + k] = acc[i][0][r];
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int k = k0 + wn * WTN + j * 32 + col_l;
        if (k >= p.K) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (n < p.N) out[(int64_t)n * p.K + k] = acc[i][j][r];
            }
    }
}

// Weight gradient with bf16 operands IN HBM (train_engine math 'bf16s'): dz [M][ldz] and X are bf16 tensors, the
// products run on v_mfma_f32_32x32x16_bf16, accumulation / slabs / dW are fp32.  Same LDS planes and transpose reads
// as wgrad_kernel<128, 128, .., 1> -- [m][out] bf16 rows of 256 bytes, 16-byte chunk XOR-swizzled by
// ((m & 3) << 2) | ((m >> 2) & 3), fragments by ds_read_b64_tr_b16 -- but the staging pass is a plain 16-byte copy
// (8 channels per item, two items per thread and operand) instead of a convert-and-split.  One tile shape, 128 x 128,
// for every layer: columns past N / K are zero-filled, and in the implicit-GEMM form every staged item resolves its
// OWN tap from its column (tap = k / C), so a tile may straddle taps (C = 64: two taps per tile).
template <bool CONV>
__global__ __launch_bounds__(256, 2) void wgrad_b16in_kernel(const WgradArgs p, const int tiles_k) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, MT = 2, NT = 2;
    constexpr int PLANE = 32 * 256;                       // bytes of one bf16 plane: 32 rows x 128 columns
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm8 = reinterpret_cast<char*>(smem);      // [2 buf][A plane, B plane]
    const __bf16* const dzp = reinterpret_cast<const __bf16*>(p.dz);
    const __bf16* const xp = reinterpret_cast<const __bf16*>(p.x);
    int tile_n, tile_k;
    wgrad_tile_of((int)blockIdx.x, p.tiles_n, tiles_k, p.tile_mode, tile_n, tile_k);
    const int n0 = tile_n * BM, k0 = tile_k * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_begin = blockIdx.y * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    // staging: item i of a thread is (row = tid / 16 + 16 i, chunk = tid % 16): columns 8 chunk .. 8 chunk + 7
    const int s_row = tid >> 4, s_chunk = tid & 15;
    const bool a_ok = n0 + 8 * s_chunk < p.N;
    const int kcol = k0 + 8 * s_chunk;
    const bool b_ok = kcol < p.K;
    int ky = 0, kx = 0, cc = kcol;
    if (CONV && b_ok) { const int tap = kcol / p.C; cc = kcol - tap * p.C; ky = tap / p.kw; kx = tap - ky * p.kw; }
    int pimg[2], poy[2], pox[2];
    const int adv_y = 32 / p.Wo, adv_x = 32 - adv_y * p.Wo;
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m_begin + s_row + 16 * i, hw = p.Ho * p.Wo;
            pimg[i] = m / hw;
            const int rem = m - pimg[i] * hw;
            poy[i] = rem / p.Wo;
            pox[i] = rem - poy[i] * p.Wo;
        }
    }
    uint4 areg[2], breg[2];
    auto load_stage = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + s_row + 16 * i;
            uint4 v = {0u, 0u, 0u, 0u};
            if (m < m_end && a_ok) v = *reinterpret_cast<const uint4*>(dzp + (int64_t)m * p.ldz + n0 + 8 * s_chunk);
            areg[i] = v;
            uint4 w = {0u, 0u, 0u, 0u};
            if (CONV) {
                const int iy = poy[i] * p.stride - p.pad + ky, ix = pox[i] * p.stride - p.pad + kx;
                if (m < m_end && b_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    w = *reinterpret_cast<const uint4*>(xp + (((int64_t)pimg[i] * p.H + iy) * p.W + ix) * p.C + cc);
                pox[i] += adv_x;                              // this row's pixel, one stage (32 rows) further on
                if (pox[i] >= p.Wo) { pox[i] -= p.Wo; ++poy[i]; }
                poy[i] += adv_y;
                while (poy[i] >= p.Ho) { poy[i] -= p.Ho; ++pimg[i]; }
            } else if (m < m_end && b_ok) {
                w = *reinterpret_cast<const uint4*>(xp + (int64_t)m * p.ldx + kcol);
            }
            breg[i] = w;
        }
    };
    auto store_stage = [&](int buf) {
        char* const base = sm8 + buf * (2 * PLANE);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = s_row + 16 * i;
            const int off = 256 * row + 16 * (s_chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
            *reinterpret_cast<uint4*>(base + off) = areg[i];
            *reinterpret_cast<uint4*>(base + PLANE + off) = breg[i];
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nst = (m_end - m_begin + 31) / 32;
    const int fhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    if (nst > 0) {
        load_stage(m_begin);
        store_stage(0);
    }
    __syncthreads();
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) load_stage(m_begin + (st + 1) * 32);
        const unsigned sbase = lds_base + buf * (2 * PLANE);
        unsigned aaddr[MT][2], baddr[NT][2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int row = 8 * (g >> 1) + 4 * rd + q;
            const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int col = wm * WTM + i * 32 + 16 * (g & 1) + 4 * pp;
                aaddr[i][rd] = sbase + 256 * row + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = wn * WTN + j * 32 + 16 * (g & 1) + 4 * pp;
                baddr[j][rd] = sbase + PLANE + 256 * row + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
            }
        }
#define GRL_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GRL_FRAG(lo4, hi4) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {               // two k-steps of 16 rows per 32-row stage
            s16x4 ra[MT][2], rb[NT][2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (ks == 0) GRL_TR(ra[i][rd], aaddr[i][rd], 0);
                    else GRL_TR(ra[i][rd], aaddr[i][rd], 4096);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (ks == 0) GRL_TR(rb[j][rd], baddr[j][rd], 0);
                    else GRL_TR(rb[j][rd], baddr[j][rd], 4096);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(GRL_FRAG(ra[i][0], ra[i][1]), GRL_FRAG(rb[j][0], rb[j][1]),
                                                                        acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef GRL_TR
#undef GRL_FRAG
        if (st + 1 < nst) store_stage(buf ^ 1);
        __syncthreads();
    }
    float* out = p.slab + (int64_t)blockIdx.y * p.slab_stride;
    const int col_l = lane & 31;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int k = k0 + wn * WTN + j * 32 + col_l;
        if (k >= p.K) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (n < p.N) out[(int64_t)n * p.K + k] = acc[i][j][r];
            }
    }
}

// The same weight gradient on a 256 x 256 tile (N >= 256 and K >= 256: layers 3 / 4, GCE, every TRL 1x1 -- most of
// the FLOPs).  The 128 x 128 form above moves 16 KB out of L2 per 2*128*128*32 FLOP (64 FLOP/B) and sits at the
// L2 -> CU ceiling (~10 TB/s = 0.65 PFLOP/s); this tile halves the bytes per FLOP:
//   * 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator registers), one workgroup
//     per CU, pixel range split over gridDim.y as before;
//   * a stage = 32 pixel rows of dz and of X, [m][256 columns] bf16 = 512-byte rows, 16-byte chunks XOR-swizzled by
//     ((m & 3) << 2) | ((m >> 2) & 3) (the transpose reads `ds_read_b64_tr_b16` stay conflict free);
//   * staging is LDS-DMA (`global_load_lds_dwordx4`, the swizzle and -- for the implicit-GEMM gather -- the tap on
//     the per-lane SOURCE; rows past the pixel range / columns past N or K read a zero chunk), THREE stage buffers
//     (96 KiB): the DMA of stage t + 2 is issued when stage t starts, waits are counted (`vmcnt(4)`: this wave's
//     four pieces of the stage about to be read) in front of a raw barrier.
template <bool CONV>
__global__ __launch_bounds__(512) void wgrad_b16in_256_kernel(const WgradArgs p, const int tiles_k) {
    constexpr int TBW = 256, PLANE = 32 * 512, STG = 2 * PLANE, NBUF = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const sm8 = reinterpret_cast<char*>(smem);
    const char* const dz8 = reinterpret_cast<const char*>(p.dz);
    const char* const x8 = reinterpret_cast<const char*>(p.x);
    int tile_n, tile_k;
    wgrad_tile_of((int)blockIdx.x, p.tiles_n, tiles_k, p.tile_mode, tile_n, tile_k);
    const int n0 = tile_n * TBW, k0 = tile_k * TBW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int m_begin = blockIdx.y * p.chunk;
    const int m_end = min(p.M, m_begin + p.chunk);
    // DMA pieces of this wave: piece i (0, 1) fills LDS rows 2 * (wave + 8 i) + (lane >> 5); a lane's chunk position
    // is lane & 31 and its SOURCE chunk position ^ swizzle(row) -- the same for both pieces (rows differ by 16)
    const int drow = 2 * wave + (lane >> 5);
    const int sc = (lane & 31) ^ (((drow & 3) << 2) | ((drow >> 2) & 3));
    const bool a_ok = n0 + 8 * sc < p.N;
    const int kcol = k0 + 8 * sc;
    const bool b_ok = kcol < p.K;
    int ky = 0, kx = 0, cc = kcol;
    if (CONV && b_ok) { const int tap = kcol / p.C; cc = kcol - tap * p.C; ky = tap / p.kw; kx = tap - ky * p.kw; }
    int pimg[2], poy[2], pox[2];
    const int adv_y = 32 / p.Wo, adv_x = 32 - adv_y * p.Wo;
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m_begin + drow + 16 * i, hw = p.Ho * p.Wo;
            pimg[i] = m / hw;
            const int rem = m - pimg[i] * hw;
            poy[i] = rem / p.Wo;
            pox[i] = rem - poy[i] * p.Wo;
        }
    }
    const char* const zsrc = reinterpret_cast<const char*>(&g_wgrad_zero_chunk);
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    auto stage = [&](int buf, int m0) {                        // 4 LDS-DMA wave-instructions per wave
        char* const base = sm8 + buf * STG;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + drow + 16 * i;
            const char* sa = (m < m_end && a_ok) ? dz8 + ((int64_t)m * p.ldz + n0 + 8 * sc) * 2 : zsrc;
            const char* sb = zsrc;
            if (CONV) {
                const int iy = poy[i] * p.stride - p.pad + ky, ix = pox[i] * p.stride - p.pad + kx;
                if (m < m_end && b_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    sb = x8 + ((((int64_t)pimg[i] * p.H + iy) * p.W + ix) * p.C + cc) * 2;
                pox[i] += adv_x;
                if (pox[i] >= p.Wo) { pox[i] -= p.Wo; ++poy[i]; }
                poy[i] += adv_y;
                while (poy[i] >= p.Ho) { poy[i] -= p.Ho; ++pimg[i]; }
            } else if (m < m_end && b_ok) {
                sb = x8 + ((int64_t)m * p.ldx + kcol) * 2;
            }
            __builtin_amdgcn_global_load_lds((gp_t)sa, (lp_t)(base + (wave + 8 * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gp_t)sb, (lp_t)(base + PLANE + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nst = (m_end - m_begin + 31) / 32;
    const int fhalf = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    if (nst > 0) stage(0, m_begin);
    if (nst > 1) stage(1, m_begin + 32);
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    // per-lane fragment offsets inside a stage (row group rd, MFMA tile): the swizzle depends on the row only
    unsigned aoff[4][2], boff[2][2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int row = 8 * (g >> 1) + 4 * rd + q;
        const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = wr * 128 + i * 32 + 16 * (g & 1) + 4 * pp;
            aoff[i][rd] = 512 * row + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wc * 64 + j * 32 + 16 * (g & 1) + 4 * pp;
            boff[j][rd] = PLANE + 512 * row + 16 * ((col >> 3) ^ sw) + 2 * (col & 7);
        }
    }
    for (int st = 0; st < nst; ++st) {
        // this wave's four pieces of stage st have landed (the four of stage st + 1 may still be in flight) ...
        if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... and every wave's; all waves are out of stage st - 1
        if (st + 2 < nst) stage((st + 2) % NBUF, m_begin + (st + 2) * 32);
        const unsigned sbase = lds_base + (st % NBUF) * STG;
#define GRL_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define GRL_FRAG(lo4, hi4) __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                   // two k-steps of 16 rows (8192 bytes) per stage
            s16x4 ra[4][2], rb[2][2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (ks == 0) GRL_TR(ra[i][rd], sbase + aoff[i][rd], 0);
                    else GRL_TR(ra[i][rd], sbase + aoff[i][rd], 8192);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ks == 0) GRL_TR(rb[j][rd], sbase + boff[j][rd], 0);
                    else GRL_TR(rb[j][rd], sbase + boff[j][rd], 8192);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(GRL_FRAG(ra[i][0], ra[i][1]), GRL_FRAG(rb[j][0], rb[j][1]),
                                                                        acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef GRL_TR
#undef GRL_FRAG
    }
    float* out = p.slab + (int64_t)blockIdx.y * p.slab_stride;
    const int col_l = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wc * 64 + j * 32 + col_l;
        if (k >= p.K) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (n < p.N) out[(int64_t)n * p.K + k] = acc[i][j][r];
            }
    }
}

// dW (torch layout [N][C][taps], or [N][K] when taps == 1) (+)= sum_z slab[z][n][t*C + c]
// A lane owns FOUR consecutive slab columns (same tap: C % 4 == 0) and keeps four slabs' loads in flight; the sum
// runs over z = 0, 1, 2, ... in that order whatever the unrolling (round 3: the scalar one-load-at-a-time form ran at
// 1.9 TB/s, 2.1 ms per step over 112 launches).
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int64_t stride,
                                    int N, int C, int taps, int Kslab, float* __restrict__ dw,
                                    int Kout, int accumulate) {
    const int K4 = Kslab >> 2;
    const int64_t total4 = (int64_t)N * K4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K4), kk = (int)(i - (int64_t)n * K4) * 4;      // slab column kk = t*C + c
        const float* src = slab + (int64_t)n * Kslab + kk;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 0) * stride);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 1) * stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 2) * stride);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (int64_t)(z + 3) * stride);
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(src + (int64_t)z * stride);
        if (taps > 1) {                               // scatter into torch's [N][C][taps]
            const int t = kk / C, c = kk - t * C;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t dst = (int64_t)n * Kout + (int64_t)(c + e) * taps + t;
                dw[dst] = (accumulate ? dw[dst] : 0.f) + s[e];
            }
        } else if (Kout == Kslab) {
            f32x4* d = reinterpret_cast<f32x4*>(dw + (int64_t)n * Kout + kk);
            if (accumulate) s = *d + s;
            *d = s;
        } else {                                      // stem: K padded to 160, 147 real columns
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (kk + e < Kout) {
                    const int64_t dst = (int64_t)n * Kout + kk + e;
                    dw[dst] = (accumulate ? dw[dst] : 0.f) + s[e];
                }
        }
    }
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    return (int)(g > 1 ? (g + 1) & ~(int64_t)1 : g);   // even: grid * 256 is then a multiple of every C/4 <= 512 of the path
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_col_stats_rows(int M) { return (M + CHUNK - 1) / CHUNK; }

int grl_launch_bn_bwd_finalize(const float* slab, int rows, int C, double count, float* dgamma, float* dbeta, float* coef,
                               hipStream_t s) {
    const int fch = finalize_fch(C);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(grl_ceil_div(C, fch)), dim3(1024), 0, s, slab, rows, C, count, dgamma,
                       dbeta, coef, fch);
    return grl_check_launch("bn_bwd_finalize");
}

extern "C" int grl_col_stats(const float* x, float* slab, int M, int C, int ld, const float* pivot, void* stream) {
    GRL_REQUIRE(x && slab && M > 0 && C % 4 == 0 && ld % 4 == 0, "col_stats: bad args");
    hipLaunchKernelGGL(col_stats_kernel, dim3(grl_ceil_div(C, 256), grl_col_stats_rows(M)), dim3(256), 0,
                       (hipStream_t)stream, x, slab, M, C, ld, pivot);
    return grl_check_launch("grl_col_stats");
}

extern "C" int grl_slab_sum(const float* slab, int rows, int64_t stride, int C, float* out, int accumulate,
                            void* stream) {
    GRL_REQUIRE(slab && out && rows > 0 && C > 0, "slab_sum: bad args");
    hipLaunchKernelGGL(slab_sum_kernel, dim3(grl_ceil_div(C, 64)), dim3(1024), 0, (hipStream_t)stream, slab, rows,
                       stride, C, out, accumulate);
    return grl_check_launch("grl_slab_sum");
}

extern "C" int grl_bn_stats_finalize(const float* slab, int rows, int C, int64_t count, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, float momentum, float eps, float* mean,
                                     float* invstd, float* scale, float* shift, const float* pivot,
                                     void* stream) {
    GRL_REQUIRE(slab && mean && invstd && scale && shift && rows > 0 && C > 0 && count > 0, "bn_stats_finalize: bad args");
    GRL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_finalize: running stats come together");
    const int fch = finalize_fch(C);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(grl_ceil_div(C, fch)), dim3(1024), 0, (hipStream_t)stream, slab,
                       rows, C, (double)count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                       eps, mean, invstd, scale, shift, pivot, fch);
    return grl_check_launch("grl_bn_stats_finalize");
}

extern "C" int grl_bn_apply(const float* z, const float* scale, const float* shift, const float* res, float* y,
                            int64_t M, int C, int relu, void* stream) {
    GRL_REQUIRE(z && scale && shift && y && M > 0 && C % 4 == 0, "bn_apply: bad args");
    const int64_t total4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, z, scale, shift,
                       res, y, C / 4, total4, relu);
    return grl_check_launch("grl_bn_apply");
}

extern "C" int grl_bn_apply_centered(const float* z, const float* mean, const float* scale, const float* beta,
                                     const float* res, float* y, int64_t M, int C, int relu, uint8_t* relu_bits,
                                     void* stream) {
    GRL_REQUIRE(z && mean && scale && y && M > 0 && C % 4 == 0, "bn_apply_centered: bad args");
    const int64_t total4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_centered_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, z, mean,
                       scale, beta, res, y, C / 4, total4, relu, relu_bits);
    return grl_check_launch("grl_bn_apply_centered");
}

extern "C" int grl_bn_bwd(const float* dy, const float* z, const float* act, const float* mean,
                          const float* invstd, const float* gamma, float* dz, float* dgamma, float* dbeta,
                          float* slab_ws, float* coef_ws, int M, int C, float* gres, int gres_accumulate,
                          const float* mask_scale, const float* mask_beta, const uint8_t* relu_bits, void* stream) {
    GRL_REQUIRE(dy && z && mean && invstd && dz && slab_ws && coef_ws && M > 0 && C % 4 == 0, "bn_bwd: bad args");
    const int rows = grl_col_stats_rows(M);
    hipStream_t s = (hipStream_t)stream;
    // gres == dy (y = relu(bn(z) + res), the residual's gradient not yet started): the reduce pass overwrites dy with
    // the masked gradient g -- which IS the residual's gradient -- and the apply pass reads it back: the activation is
    // read once instead of twice and no separate gres tensor is written (round 3; -2 of 9 tensor passes on the widest
    // BatchNorms of the step)
    const bool inplace = (act || relu_bits) && gres == dy && !gres_accumulate;
    float* const gout = inplace ? const_cast<float*>(dy) : nullptr;
    if (C <= 64)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<16>, dim3(grl_ceil_div(C, 64), rows), dim3(256), 0, s, dy, z, act, mean,
                           invstd, slab_ws, M, C, mask_scale, mask_beta, gout, relu_bits);
    else if (C <= 128)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<32>, dim3(grl_ceil_div(C, 128), rows), dim3(256), 0, s, dy, z, act, mean,
                           invstd, slab_ws, M, C, mask_scale, mask_beta, gout, relu_bits);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<64>, dim3(grl_ceil_div(C, 256), rows), dim3(256), 0, s, dy, z, act, mean,
                           invstd, slab_ws, M, C, mask_scale, mask_beta, gout, relu_bits);
    if (grl_bn_finapply_takes(rows, C)) {        // round 6: finalize inside the apply pass (train_bnfuse.hip), bit-identical
        if (inplace)
            return grl_launch_bn_bwd_finapply(0, slab_ws, rows, C, (double)M, dgamma, dbeta, dy, z, nullptr, mean, invstd, gamma, dz, M,
                                              nullptr, 0, nullptr, nullptr, nullptr, s);
        return grl_launch_bn_bwd_finapply(0, slab_ws, rows, C, (double)M, dgamma, dbeta, dy, z, act, mean, invstd, gamma, dz, M, gres,
                                          gres_accumulate, mask_scale, mask_beta, relu_bits, s);
    }
    if (int e = grl_launch_bn_bwd_finalize(slab_ws, rows, C, (double)M, dgamma, dbeta, coef_ws, s)) return e;
    const int64_t total4 = (int64_t)M * C / 4;
    if (inplace)
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, s, dy, z, (const float*)nullptr, mean,
                           invstd, gamma, coef_ws, dz, C, total4, (float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr,
                           (const uint8_t*)nullptr);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, s, dy, z, act, mean, invstd, gamma,
                           coef_ws, dz, C, total4, gres, gres_accumulate, mask_scale, mask_beta, relu_bits);
    return grl_check_launch("grl_bn_bwd");
}

extern "C" int grl_bn_bwd_finish(const float* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                                 float* dz, float* dgamma, float* dbeta, const float* slab, int rows, float* coef_ws, int M,
                                 int C, float* gres, int gres_accumulate, void* stream) {
    GRL_REQUIRE(g && z && mean && invstd && dz && slab && coef_ws && rows > 0 && M > 0 && C % 4 == 0, "bn_bwd_finish: bad args");
    hipStream_t s = (hipStream_t)stream;
    // g is masked already: no activation, no mask recomputation; gres == g (the residual adopts the buffer) needs nothing
    float* const gres2 = gres == g ? nullptr : gres;
    if (grl_bn_finapply_takes(rows, C))
        return grl_launch_bn_bwd_finapply(0, slab, rows, C, (double)M, dgamma, dbeta, g, z, nullptr, mean, invstd, gamma, dz, M, gres2,
                                          gres2 ? gres_accumulate : 0, nullptr, nullptr, nullptr, s);
    if (int e = grl_launch_bn_bwd_finalize(slab, rows, C, (double)M, dgamma, dbeta, coef_ws, s)) return e;
    const int64_t total4 = (int64_t)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, s, g, z, (const float*)nullptr, mean, invstd,
                       gamma, coef_ws, dz, C, total4, gres2, gres2 ? gres_accumulate : 0, (const float*)nullptr,
                       (const float*)nullptr, (const uint8_t*)nullptr);
    return grl_check_launch("grl_bn_bwd_finish");
}

extern "C" int grl_relu_bwd(const float* dy, const float* act, float* out, int64_t n, int accumulate, void* stream) {
    GRL_REQUIRE(dy && out && n > 0 && n % 4 == 0, "relu_bwd: bad args");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, act, out, n / 4,
                       accumulate);
    return grl_check_launch("grl_relu_bwd");
}

extern "C" int grl_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n, void* stream) {
    GRL_REQUIRE(a && y && n > 0 && n % 4 == 0, "axpby: bad args");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, y, alpha, beta,
                       n / 4);
    return grl_check_launch("grl_axpby");
}

extern "C" int grl_axpy_strided(float* dst, int64_t dst_stride, const float* src, int64_t src_stride, int nb,
                                int64_t inner, float alpha, int accumulate, void* stream) {
    GRL_REQUIRE(dst && src && nb > 0 && inner > 0 && inner % 4 == 0 && dst_stride % 4 == 0 && src_stride % 4 == 0,
                "axpy_strided: bad args");
    const int64_t total4 = (int64_t)nb * inner / 4;
    hipLaunchKernelGGL(axpy_strided_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dst,
                       dst_stride / 4, src, src_stride / 4, inner / 4, alpha, accumulate, total4);
    return grl_check_launch("grl_axpy_strided");
}

extern "C" int grl_transpose(const float* x, float* y, int R, int C, int ldx, void* stream) {
    GRL_REQUIRE(x && y && R > 0 && C > 0 && ldx >= C, "transpose: bad args");
    hipLaunchKernelGGL(transpose_kernel, dim3(grl_ceil_div(C, 32), grl_ceil_div(R, 32)), dim3(256), 0,
                       (hipStream_t)stream, x, y, R, C, ldx);
    return grl_check_launch("grl_transpose");
}

extern "C" int grl_weight_prep(const GrlPrepEntry* table_dev, int count, void* stream) {
    GRL_REQUIRE(table_dev && count > 0, "weight_prep: bad args");
    hipLaunchKernelGGL(weight_prep_kernel, dim3(64, count), dim3(256), 0, (hipStream_t)stream, table_dev);
    return grl_check_launch("grl_weight_prep");
}

extern "C" int grl_pack_dgrad_weight(const float* w, float* out, int N, int C, int kh, int kw, void* stream) {
    GRL_REQUIRE(w && out && N > 0 && C > 0 && kh > 0 && kw > 0, "pack_dgrad_weight: bad args");
    hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3(grid_for((int64_t)N * C * kh * kw)), dim3(256), 0,
                       (hipStream_t)stream, w, out, N, C, kh * kw);
    return grl_check_launch("grl_pack_dgrad_weight");
}

extern "C" int grl_dilate2(const float* dz, float* up, int n, int Ho, int Wo, int H, int W, int C, int accumulate,
                           int oy_off, int ox_off, void* stream) {
    GRL_REQUIRE(dz && up && n > 0 && C % 4 == 0, "dilate2: bad args");
    GRL_REQUIRE((oy_off == 0 || oy_off == 1) && (ox_off == 0 || ox_off == 1), "dilate2: offsets are 0 or 1");
    const int64_t total4 = (int64_t)n * H * W * (C / 4);
    hipLaunchKernelGGL(dilate2_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dz, up, Ho, Wo, H, W,
                       C / 4, total4, accumulate, oy_off, ox_off);
    return grl_check_launch("grl_dilate2");
}

extern "C" int grl_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int n, int H, int W, int C,
                                    void* stream) {
    GRL_REQUIRE(x && dy && dx && n > 0 && C % 4 == 0, "maxpool_bwd: bad args");
    const int64_t total4 = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);      // 2x2 input blocks
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, H, W,
                       C / 4, total4);
    return grl_check_launch("grl_maxpool3x3s2_bwd");
}

extern "C" int grl_bn_relu_maxpool3x3s2(const float* z, const float* mean, const float* scale, const float* beta, float* y,
                                        uint8_t* idx, int n, int H, int W, int C, void* stream) {
    GRL_REQUIRE(z && mean && scale && y && idx && n > 0 && C % 4 == 0 && ((uintptr_t)idx & 3) == 0, "bn_relu_maxpool: bad args");
    const int64_t total = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, z, mean, scale, beta,
                       y, idx, n, H, W, C);
    return grl_check_launch("grl_bn_relu_maxpool3x3s2");
}

extern "C" int grl_maxpool3x3s2_bwd_idx(const uint8_t* idx, const float* dy, float* dx, int n, int H, int W, int C,
                                        void* stream) {
    GRL_REQUIRE(idx && dy && dx && n > 0 && C % 4 == 0 && ((uintptr_t)idx & 3) == 0, "maxpool_bwd_idx: bad args");
    const int64_t total4 = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);      // 2x2 input blocks
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, idx, dy, dx, H, W,
                       C / 4, total4);
    return grl_check_launch("grl_maxpool3x3s2_bwd_idx");
}

extern "C" int grl_stem_im2col(const float* x, float* col, int n, int H, int W, int Kp, void* stream) {
    GRL_REQUIRE(x && col && n > 0 && Kp >= 147 && Kp % 32 == 0, "stem_im2col: bad args");
    const int64_t total = (int64_t)n * (H / 2) * (W / 2) * Kp;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, col, H, W, Kp,
                       total);
    return grl_check_launch("grl_stem_im2col");
}

static int stem_wgrad_wgs(int n, int H, int W) {
    const int64_t tiles = (int64_t)n * grl_ceil_div(H / 2, SW_TH) * grl_ceil_div(W / 2, SW_TW);
    return (int)(tiles < SW_WGS ? tiles : SW_WGS);
}

extern "C" int64_t grl_stem_wgrad_workspace_floats(int n, int H, int W) {
    return n > 0 && H > 0 && W > 0 ? (int64_t)stem_wgrad_wgs(n, H, W) * 64 * SW_K : 0;
}

extern "C" int grl_stem_wgrad(const float* x, const void* dz, int dz_bf16, float* dw, float* ws, int n, int H, int W,
                              int accumulate, void* stream) {
    GRL_REQUIRE(x && dz && dw && ws && n > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_wgrad: bad args");
    GRL_REQUIRE(((uintptr_t)dz & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)dw & 3) == 0, "stem_wgrad: alignment");
    hipStream_t s = (hipStream_t)stream;
    const int wgs = stem_wgrad_wgs(n, H, W);
    if (dz_bf16)
        hipLaunchKernelGGL(stem_wgrad_kernel<true>, dim3(wgs), dim3(512), 0, s, x, dz, ws, n, H, W);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(wgs), dim3(512), 0, s, x, dz, ws, n, H, W);
    if (int e = grl_check_launch("grl_stem_wgrad")) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((int64_t)64 * (SW_K / 4))), dim3(256), 0, s, ws, wgs,
                       (int64_t)64 * SW_K, 64, SW_K, 1, SW_K, dw, 147, accumulate ? 1 : 0);
    return grl_check_launch("grl_stem_wgrad (reduce)");
}

// How many private slabs the pixel range is split into.  The grid is tiles x splits workgroups
// and the chip holds `slots` of them at once (two 64 KiB-LDS workgroups per CU for the 128-wide
// tiles, four for 64 x 64): pick the split whose last round of workgroups is (nearly) full --
// 576 workgroups on 512 slots run as long as 1024 -- while keeping enough K stages per
// workgroup to amortise its prologue/epilogue and charging the reduce pass for every slab.
static int wgrad_splits(const GrlWgrad& d, int bm, int bn) {
    const int64_t tiles = (int64_t)((d.N + bm - 1) / bm) * ((d.K + bn - 1) / bn);
    int64_t max_splits = (d.M + 255) / 256;
    if (max_splits < 1) max_splits = 1;
    if (const char* e = getenv("GRL_WGRAD_SPLITS")) {               // kernel tuning only
        int64_t f = atoi(e);
        return (int)(f < 1 ? 1 : (f > max_splits ? max_splits : f));
    }
    const double slots = (bm == 64 && bn == 64) ? 1024.0 : (bm == 256 ? 256.0 : 512.0);
    int64_t smax = (int64_t)(3.0 * slots / (double)tiles) + 1;
    if (smax > max_splits) smax = max_splits;
    int best = 1;
    double best_cost = 1e30;
    for (int64_t s = 1; s <= smax; ++s) {
        const double wgs = (double)tiles * (double)s;
        const double rounds = ceil(wgs / slots);
        const double stages = (double)d.M / (32.0 * (double)s);
        // time ~ rounds x (stages + fixed per-workgroup cost) + slab reduce traffic (in stage units)
        const double cost = rounds * (stages + 8.0) + 0.012 * (double)s * (double)tiles;
        if (cost < best_cost * 0.999) { best_cost = cost; best = (int)s; }
    }
    return best;
}

static void wgrad_tile(const GrlWgrad& d, int* bm, int* bn) {
    if (d.in_bf16) {                                       // edges are zero-filled, tiles may straddle taps
        static const bool big = [] { const char* e = getenv("GRL_WGRAD_B16_256"); return !e || atoi(e) != 0; }();     // (0: A/B only)
        // (measured per shape, tools/wgrad_bench.py: with fewer than ~8 tiles of 256 x 256 the pixel range is cut into
        // too many short splits -- 16384 x 1024 x 256: 28 -> 34 us -- so small N x K stays on the 128 x 128 tile)
        const bool t256 = big && d.N >= 256 && d.K >= 256 && (int64_t)d.N * d.K >= (1 << 19);
        *bm = *bn = t256 ? 256 : 128;
        return;
    }
    *bm = d.N >= 128 ? 128 : 64;
    const int cdiv = d.conv ? d.C : d.K;
    *bn = (cdiv % 128 == 0) ? 128 : 64;
}

extern "C" int64_t grl_wgrad_workspace_floats(const GrlWgrad* d) {
    if (!d) return -1;
    int bm, bn;
    wgrad_tile(*d, &bm, &bn);
    return (int64_t)wgrad_splits(*d, bm, bn) * d->N * d->K;
}

extern "C" int grl_conv_wgrad_f32(const GrlWgrad* desc, void* stream) {
    GRL_REQUIRE(desc, "wgrad: null desc");
    const GrlWgrad& d = *desc;
    GRL_REQUIRE(d.dz && d.x && d.dw && d.workspace, "wgrad: null pointer");
    GRL_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.ldz % 4 == 0, "wgrad: bad shape");
    GRL_REQUIRE(d.N % 4 == 0 && d.K % 4 == 0, "wgrad: N and K must be multiples of 4");
    GRL_REQUIRE(d.math == GRL_MATH_F32 || d.math == GRL_MATH_BF16X3 || d.math == GRL_MATH_BF16, "wgrad: unknown math mode");
    if (d.in_bf16) {
        GRL_REQUIRE(d.N % 8 == 0 && d.K % 8 == 0 && d.ldz % 8 == 0 && (d.conv || d.ldx % 8 == 0) &&
                    ((uintptr_t)d.dz & 15) == 0 && ((uintptr_t)d.x & 15) == 0, "wgrad bf16-in: N, K, ld % 8, 16-byte aligned operands");
        GRL_REQUIRE(!d.conv || d.C % 8 == 0, "wgrad bf16-in conv: C % 8");
    }
    if (d.conv) {
        GRL_REQUIRE(d.C % 64 == 0 && d.K == d.kh * d.kw * d.C, "wgrad conv: C % 64, K = kh*kw*C");
        GRL_REQUIRE(d.M % (d.Ho * d.Wo) == 0, "wgrad conv: M must be nimg*Ho*Wo");
    } else {
        GRL_REQUIRE(d.ldx % 4 == 0, "wgrad: ldx % 4");
    }
    int bm, bn;
    wgrad_tile(d, &bm, &bn);
    const int splits = wgrad_splits(d, bm, bn);
    WgradArgs a;
    a.dz = d.dz; a.x = d.x; a.slab = d.workspace;
    a.M = d.M; a.N = d.N; a.K = d.K; a.ldz = d.ldz; a.ldx = d.ldx;
    a.slab_stride = (int64_t)d.N * d.K;
    a.chunk = (int)((((int64_t)d.M + splits - 1) / splits + 31) / 32 * 32);
    a.conv = d.conv; a.H = d.H; a.W = d.W; a.C = d.C; a.Ho = d.Ho; a.Wo = d.Wo;
    a.kh = d.kh; a.kw = d.kw; a.stride = d.stride; a.pad = d.pad;
    const int real_splits = (d.M + a.chunk - 1) / a.chunk;
    const int tiles_n = (d.N + bm - 1) / bm, tiles_k = (d.K + bn - 1) / bn;
    a.tiles_n = tiles_n;
    a.tile_mode = wgrad_tile_mode(tiles_n, tiles_k);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)2 * 32 * (bm + bn) * sizeof(float);
    dim3 grid(tiles_n * tiles_k, real_splits);
#define GRL_WGRAD_LAUNCH(BM_, BN_)                                                                        \
    do {                                                                                                  \
        if (d.conv) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true>), grid, dim3(256), lds, s, a, tiles_k);  \
        else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, false>), grid, dim3(256), lds, s, a, tiles_k);        \
    } while (0)
    if (d.in_bf16 && bm == 256) {
        constexpr size_t lds256 = (size_t)3 * 2 * 32 * 512;              // three stages of (dz, X) planes: 96 KiB
        static const bool attr = [] {
            (void)hipFuncSetAttribute((const void*)wgrad_b16in_256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
            (void)hipFuncSetAttribute((const void*)wgrad_b16in_256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
            return true;
        }();
        (void)attr;
        if (d.conv) hipLaunchKernelGGL((wgrad_b16in_256_kernel<true>), grid, dim3(512), lds256, s, a, tiles_k);
        else hipLaunchKernelGGL((wgrad_b16in_256_kernel<false>), grid, dim3(512), lds256, s, a, tiles_k);
    } else if (d.in_bf16) {
        if (d.conv) hipLaunchKernelGGL((wgrad_b16in_kernel<true>), grid, dim3(256), (size_t)4 * 32 * 256, s, a, tiles_k);
        else hipLaunchKernelGGL((wgrad_b16in_kernel<false>), grid, dim3(256), (size_t)4 * 32 * 256, s, a, tiles_k);
    } else if (bm == 128 && bn == 128 && d.math == GRL_MATH_BF16X3) {
        if (d.conv) hipLaunchKernelGGL((wgrad_kernel<128, 128, true, 3>), grid, dim3(256), lds, s, a, tiles_k);
        else hipLaunchKernelGGL((wgrad_kernel<128, 128, false, 3>), grid, dim3(256), lds, s, a, tiles_k);
    } else if (bm == 128 && bn == 128 && d.math == GRL_MATH_BF16) {
        if (d.conv) hipLaunchKernelGGL((wgrad_kernel<128, 128, true, 1>), grid, dim3(256), lds, s, a, tiles_k);
        else hipLaunchKernelGGL((wgrad_kernel<128, 128, false, 1>), grid, dim3(256), lds, s, a, tiles_k);
    } else if (bm == 128 && bn == 128) GRL_WGRAD_LAUNCH(128, 128);
    else if (bm == 128 && bn == 64) GRL_WGRAD_LAUNCH(128, 64);
    else if (bm == 64 && bn == 128) GRL_WGRAD_LAUNCH(64, 128);
    else GRL_WGRAD_LAUNCH(64, 64);
#undef GRL_WGRAD_LAUNCH
    const int taps = d.conv ? d.kh * d.kw : 1;
    const int Cc = d.conv ? d.C : d.K;
    const int kout = d.k_out > 0 ? d.k_out : d.K;      // stem: K padded to 160, 147 real
    GRL_REQUIRE(taps == 1 || kout == d.K, "wgrad conv: k_out is a dense-only option");
    GRL_REQUIRE(kout == d.K || ((uintptr_t)d.dw & 3) == 0, "wgrad: dw alignment");
    GRL_REQUIRE(kout != d.K || taps > 1 || ((uintptr_t)d.dw & 15) == 0, "wgrad: dw must be 16-byte aligned");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((int64_t)d.N * (d.K / 4))), dim3(256), 0, s, d.workspace,
                       real_splits, a.slab_stride, d.N, Cc, taps, d.K, d.dw, kout, d.accumulate);
    return grl_check_launch("grl_conv_wgrad_f32");
}
