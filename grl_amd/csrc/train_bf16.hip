// bf16-STORAGE training kernels for gfx950 (train_engine math mode 'bf16s': BASELINE configs[2] -- "bf16 MFMA,
// T = 8, P x K = 16 x 4" -- as the TRAINING batch it describes).  Every [pixels][channels] activation, every saved
// tensor and every activation gradient is bf16 in HBM (half the bytes of the HBM-bound passes); arithmetic, all
// reductions, BatchNorm statistics, per-channel vectors, parameter gradients and the optimizer state stay fp32.
// Same semantics and reference call sites as the fp32 twins in train.hip / train_head.hip
// (reid/models/resnets1.py:76-91, basebranch.py:38-66, grl_model.py:71-83,131-180 and their autograd backward,
// reid/train/trainer.py:54).  Every lane moves 16 bytes = 8 channels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CHUNK = 128;     // rows per partial of the column reductions (= grl_col_stats_rows)

__device__ __forceinline__ f32x8 zero8() { return f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x8 ld8(const __bf16* p) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
    f32x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (float)v[e];
    return r;
}
__device__ __forceinline__ void st8(__bf16* p, const f32x8 v) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = r;
}
__device__ __forceinline__ f32x8 ld8f(const float* p) {      // 8 consecutive floats (32-byte aligned vectors)
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void st8f(float* p, const f32x8 v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// ---------------------------------------------------------------------------------
// y = relu?((z - mean) * scale + beta + res)        (train-mode BatchNorm apply, centred first as torch does)
__global__ void bn_apply_centered_b16_kernel(const __bf16* __restrict__ z, const float* __restrict__ mean,
                                             const float* __restrict__ scale, const float* __restrict__ beta,
                                             const __bf16* __restrict__ res, __bf16* __restrict__ y, int C8,
                                             int64_t total8, int relu, uint8_t* __restrict__ bits) {
    // bits (may be NULL): one byte per eight outputs, bit e = (stored bf16 y[e] > 0): grl_bn_bwd_bf16's relu_bits
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (step % C8 == 0) {
        // a thread meets the SAME eight channels in every iteration: their vectors live in registers (three 32-byte
        // vector loads per 16 bytes of payload were L1 traffic that capped the pass below the HBM rate)
        const int c = (int)(i % C8) * 8;
        const f32x8 mu = ld8f(mean + c), sc = ld8f(scale + c);
        f32x8 be = zero8();
        if (beta) be = ld8f(beta + c);
        for (; i < total8; i += step) {
            f32x8 v = (ld8(z + i * 8) - mu) * sc;
            if (beta) v += be;
            if (res) v += ld8(res + i * 8);
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            st8(y + i * 8, v);
            if (bits) {
                uint32_t mk = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) mk |= (uint32_t)((float)(__bf16)v[e] > 0.f) << e;
                bits[i] = (uint8_t)mk;
            }
        }
        return;
    }
    for (; i < total8; i += step) {
        const int c = (int)(i % C8) * 8;
        f32x8 v = (ld8(z + i * 8) - ld8f(mean + c)) * ld8f(scale + c);
        if (beta) v += ld8f(beta + c);
        if (res) v += ld8(res + i * 8);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        st8(y + i * 8, v);
        if (bits) {
            uint32_t mk = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) mk |= (uint32_t)((float)(__bf16)v[e] > 0.f) << e;
            bits[i] = (uint8_t)mk;
        }
    }
}

// Column statistics of x[M][C] (row stride ld, bf16): slab[chunk][0][c] = sum, [1][c] = sum of squares of
// (x - pivot[c]); pivot is an fp32 VECTOR here (NULL = 0).  LPR = min(32, C / 8) lanes cover a row's channels of
// one 256-channel column block, the 256 / LPR lane groups take rows chunk*128 + group, + groups, ...; fixed
// order: a lane over its rows, then the lane groups in index order through LDS.
__global__ __launch_bounds__(256) void col_stats_b16_kernel(const __bf16* __restrict__ x, float* __restrict__ slab,
                                                            int M, int C, int ld, const float* __restrict__ pivot,
                                                            int LPR) {
    __shared__ f32x8 red[2][256];
    const int chunk = blockIdx.y, sub = threadIdx.x % LPR, part = threadIdx.x / LPR, parts = 256 / LPR;
    const int c = blockIdx.x * 256 + sub * 8;
    f32x8 s = zero8(), q = zero8();
    if (c < C) {
        const int r1 = min(M, (chunk + 1) * CHUNK);
        const f32x8 pv = pivot ? ld8f(pivot + c) : zero8();
#pragma unroll 4
        for (int r = chunk * CHUNK + part; r < r1; r += parts) {
            const f32x8 v = ld8(x + (int64_t)r * ld + c) - pv;
            s += v; q += v * v;
        }
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (part == 0 && c < C) {
        s = red[0][sub]; q = red[1][sub];
        for (int k = 1; k < parts; ++k) { s += red[0][k * LPR + sub]; q += red[1][k * LPR + sub]; }
        st8f(slab + ((int64_t)chunk * 2 + 0) * C + c, s);
        st8f(slab + ((int64_t)chunk * 2 + 1) * C + c, q);
    }
}

// BN backward pass 1: g = dy * mask ; slab[chunk][0][c] = sum g, [1][c] = sum g * xhat.  mask = (act > 0), or
// recomputed from z with the forward's own (z - mean) * mscale + mbeta (y = relu(bn(z)) without a residual), or none.
// (dy and gout carry no __restrict__: the in-place form passes the same buffer for both)
__global__ __launch_bounds__(256) void bn_bwd_reduce_b16_kernel(
    const __bf16* dy, const __bf16* __restrict__ z, const __bf16* __restrict__ act,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ slab, int M, int C,
    const float* __restrict__ mscale, const float* __restrict__ mbeta, int LPR, __bf16* gout,
    const uint8_t* __restrict__ bits) {
    __shared__ f32x8 red[2][256];
    const int chunk = blockIdx.y, sub = threadIdx.x % LPR, part = threadIdx.x / LPR, parts = 256 / LPR;
    const int c = blockIdx.x * 256 + sub * 8;
    f32x8 s = zero8(), q = zero8();
    if (c < C) {
        const f32x8 mu = ld8f(mean + c), is = ld8f(invstd + c);
        const f32x8 ms = mscale ? ld8f(mscale + c) : zero8(), mb = mbeta ? ld8f(mbeta + c) : zero8();
        const int r1 = min(M, (chunk + 1) * CHUNK);
#pragma unroll 4
        for (int r = chunk * CHUNK + part; r < r1; r += parts) {
            const int64_t o = (int64_t)r * C + c;
            f32x8 g = ld8(dy + o);
            const f32x8 zc = ld8(z + o) - mu;
            if (bits) {                               // the forward's recorded (y > 0) bits instead of the activation
                const uint32_t mk = bits[o >> 3];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
            } else if (act) {
                const f32x8 a = ld8(act + o);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
            } else if (mscale) {
                // the forward stored y = bf16(relu(t)) with this same t = (z - mean) * scale + beta (same operations on the
                // same bf16 z): y > 0 <=> t > 0, except where a positive t below bf16's smallest subnormal (9e-41) rounded to 0
                const f32x8 t = zc * ms + mb;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
            }
            s += g; q += g * (zc * is);
            if (gout) st8(gout + o, g);               // masked gradient, in place over dy (see grl_bn_bwd)
        }
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (part == 0 && c < C) {
        s = red[0][sub]; q = red[1][sub];
        for (int k = 1; k < parts; ++k) { s += red[0][k * LPR + sub]; q += red[1][k * LPR + sub]; }
        st8f(slab + ((int64_t)chunk * 2 + 0) * C + c, s);
        st8f(slab + ((int64_t)chunk * 2 + 1) * C + c, q);
    }
}

// pass 2: dz = gamma * invstd * (g - mean_g - xhat * mean_gx); the residual branch receives the masked g
__global__ void bn_bwd_apply_b16_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ z,
                                        const __bf16* __restrict__ act, const float* __restrict__ mean,
                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                        const float* __restrict__ coef, __bf16* __restrict__ dz, int C,
                                        int64_t total8, __bf16* __restrict__ gres, int gres_accumulate,
                                        const float* __restrict__ mscale, const float* __restrict__ mbeta,
                                        const uint8_t* __restrict__ bits) {
    const int C8 = C >> 3;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (step % C8 == 0) {              // a thread's eight channels are loop-invariant: the vectors live in registers
        const int c = (int)(i % C8) * 8;
        const f32x8 mu = ld8f(mean + c), is = ld8f(invstd + c), k0 = ld8f(coef + c), k1 = ld8f(coef + C + c);
        f32x8 gm = is;
        if (gamma) gm = gm * ld8f(gamma + c);
        f32x8 ms = zero8(), mb = zero8();
        if (mscale) ms = ld8f(mscale + c);
        if (mbeta) mb = ld8f(mbeta + c);
        for (; i < total8; i += step) {
            f32x8 g = ld8(dy + i * 8);
            const f32x8 zc = ld8(z + i * 8) - mu;
            if (bits) {
                const uint32_t mk = bits[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
            } else if (act) {
                const f32x8 a = ld8(act + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
            } else if (mscale) {
                f32x8 t = zc * ms;
                if (mbeta) t += mb;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
            }
            if (gres) {
                f32x8 r = g;
                if (gres_accumulate) r += ld8(gres + i * 8);
                st8(gres + i * 8, r);
            }
            st8(dz + i * 8, gm * (g - k0 - (zc * is) * k1));
        }
        return;
    }
    for (; i < total8; i += step) {
        const int c = (int)(i % C8) * 8;
        f32x8 g = ld8(dy + i * 8);
        const f32x8 zc = ld8(z + i * 8) - ld8f(mean + c);
        if (bits) {
            const uint32_t mk = bits[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (mk >> e) & 1u ? g[e] : 0.f;
        } else if (act) {
            const f32x8 a = ld8(act + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
        } else if (mscale) {
            f32x8 t = zc * ld8f(mscale + c);
            if (mbeta) t += ld8f(mbeta + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = t[e] > 0.f ? g[e] : 0.f;
        }
        if (gres) {
            f32x8 r = g;
            if (gres_accumulate) r += ld8(gres + i * 8);
            st8(gres + i * 8, r);
        }
        const f32x8 is = ld8f(invstd + c);
        f32x8 gm = is;
        if (gamma) gm = gm * ld8f(gamma + c);
        st8(dz + i * 8, gm * (g - ld8f(coef + c) - (zc * is) * ld8f(coef + C + c)));
    }
}

// out = (accumulate ? out : 0) + dy * (act > 0)
__global__ void relu_bwd_b16_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ act,
                                    __bf16* __restrict__ out, int64_t total8, int accumulate) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        f32x8 g = ld8(dy + i * 8);
        if (act) {
            const f32x8 a = ld8(act + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
        }
        if (accumulate) g += ld8(out + i * 8);
        st8(out + i * 8, g);
    }
}

__global__ void axpby_b16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ y,
                                 float alpha, float beta, int64_t total8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        f32x8 v = ld8(a + i * 8) * alpha;
        if (b) v += ld8(b + i * 8) * beta;
        st8(y + i * 8, v);
    }
}

// dst[b*dstride + i] (+)= alpha * src[b*sstride + i]   (i < inner)
__global__ void axpy_strided_b16_kernel(__bf16* __restrict__ dst, int64_t dstride8, const __bf16* __restrict__ src,
                                        int64_t sstride8, int64_t inner8, float alpha, int accumulate, int64_t total8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / inner8, r = i - b * inner8;
        f32x8 v = ld8(src + (b * sstride8 + r) * 8) * alpha;
        __bf16* d = dst + (b * dstride8 + r) * 8;
        if (accumulate) v += ld8(d);
        st8(d, v);
    }
}

// stride-2 data gradients: scatter dz (output resolution) to the pixels of parity class (oy_off, ox_off)
__global__ void dilate2_b16_kernel(const __bf16* __restrict__ dz, __bf16* __restrict__ up, int Ho, int Wo, int H, int W,
                                   int C8, int64_t total8, int accumulate, int oy_off, int ox_off) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C8;
        int64_t r = i / C8;
        const int x = r % W; r /= W;
        const int y = r % H;
        const int img = r / H;
        f32x8 v = zero8();
        const bool hit = (x & 1) == ox_off && (y & 1) == oy_off && (y >> 1) < Ho && (x >> 1) < Wo;
        if (hit) v = ld8(dz + ((((int64_t)img * Ho + (y >> 1)) * Wo + (x >> 1)) * C8 + c) * 8);
        if (accumulate) {                       // 1: add at the class pixels, 2: write only them
            if (!hit) continue;
            if (accumulate == 1) v += ld8(up + i * 8);
        }
        st8(up + i * 8, v);
    }
}

// bf16-storage twins of bn_relu_maxpool_kernel / maxpool_bwd_idx_kernel (train.hip): a lane owns 8 channels; every tap
// is rounded to bf16 before the comparison, exactly what the separate passes compared (the stored bf16 activation).
__global__ void bn_relu_maxpool_b16_kernel(const __bf16* __restrict__ z, const float* __restrict__ mean,
                                           const float* __restrict__ scale, const float* __restrict__ beta,
                                           __bf16* __restrict__ y, uint8_t* __restrict__ idx, int n, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C8 = C >> 3;
    const int64_t total = (int64_t)n * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        int64_t r = i / C8;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int img = r / Ho;
        const f32x8 mu = ld8f(mean + c), sc = ld8f(scale + c);
        const f32x8 be = beta ? ld8f(beta + c) : zero8();
        f32x8 m;
        uint32_t pos[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; pos[e] = 0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                f32x8 v = (ld8(z + (((int64_t)img * H + iy) * W + ix) * C + c) - mu) * sc;
                if (beta) v += be;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = (float)(__bf16)(v[e] > 0.f ? v[e] : 0.f);
                    if (a > m[e]) { m[e] = a; pos[e] = (uint32_t)(ky * 3 + kx); }
                }
            }
        }
        st8(y + i * 8, m);
        uint2 pk;
        pk.x = pos[0] | pos[1] << 8 | pos[2] << 16 | pos[3] << 24;
        pk.y = pos[4] | pos[5] << 8 | pos[6] << 16 | pos[7] << 24;
        reinterpret_cast<uint2*>(idx)[i] = pk;
    }
}

__global__ void maxpool_bwd_idx_b16_kernel(const uint8_t* __restrict__ idx, const __bf16* __restrict__ dy,
                                           __bf16* __restrict__ dx, int H, int W, int C8, int64_t total8) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C = C8 * 8;
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        int64_t r = i / C8;
        const int b = r % Wb; r /= Wb;
        const int a = r % Hb;
        const int img = r / Hb;
        f32x8 g[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) g[u][v] = zero8();
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int oy = a + wy;
            if (oy >= Ho) continue;
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ox = b + wx;
                if (ox >= Wo) continue;
                const int64_t o = (((int64_t)img * Ho + oy) * Wo + ox) * C + c;
                const f32x8 d = ld8(dy + o);
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
                for (int u = wy; u < 2; ++u)
#pragma unroll
                    for (int v = wx; v < 2; ++v) {
                        const uint32_t k = (uint32_t)((u - 2 * wy + 1) * 3 + (v - 2 * wx + 1));
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const uint32_t pe = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                            if (pe == k) g[u][v][e] += d[e];
                        }
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int iy = 2 * a + u, ix = 2 * b + v;
                if (iy < H && ix < W) st8(dx + (((int64_t)img * H + iy) * W + ix) * C + c, g[u][v]);
            }
    }
}

// max-pool 3x3/s2/p1 backward, gather form with torch's first-maximum rule (see maxpool_bwd_kernel in train.hip):
// one lane per 2x2 block of input pixels x 8 channels
__global__ void maxpool_bwd_b16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                       __bf16* __restrict__ dx, int H, int W, int C8, int64_t total8) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C = C8 * 8;
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        int64_t r = i / C8;
        const int b = r % Wb; r /= Wb;
        const int a = r % Hb;
        const int img = r / Hb;
        const __bf16* xi = x + (int64_t)img * H * W * C + c;
        f32x8 g[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) g[u][v] = zero8();
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int oy = a + wy;
            if (oy >= Ho) continue;
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ox = b + wx;
                if (ox >= Wo) continue;
                f32x8 best;
                int bpos[8];                         // (yy << 16) | xx of the first maximum, per channel
#pragma unroll
                for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bpos[e] = -1; }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 - 1 + ky;
                    if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        const f32x8 o = ld8(xi + ((int64_t)yy * W + xx) * C);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (o[e] > best[e] || bpos[e] < 0) { best[e] = o[e]; bpos[e] = (yy << 16) | xx; }
                    }
                }
                const f32x8 d = ld8(dy + (((int64_t)img * Ho + oy) * Wo + ox) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int uy = (bpos[e] >> 16) - 2 * a, ux = (bpos[e] & 0xffff) - 2 * b;
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
                if (iy < H && ix < W) st8(dx + (((int64_t)img * H + iy) * W + ix) * C + c, g[u][v]);
            }
    }
}

// stem im2col for the 7x7 weight gradient (fp32 clip in, bf16 columns out): col[m][k], k = (c*7+ky)*7+kx, padded to
// Kp; a lane builds 8 consecutive k of one pixel and stores them as one 16-byte chunk
__global__ void stem_im2col_b16_kernel(const float* __restrict__ x, __bf16* __restrict__ col, int H, int W, int Kp,
                                       int64_t total8) {
    const int Ho = H / 2, Wo = W / 2, K8 = Kp / 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int k0 = (int)(i % K8) * 8;
        int64_t m = i / K8;
        const int ox = m % Wo; m /= Wo;
        const int oy = m % Ho;
        const int img = m / Ho;
        f32x8 v = zero8();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            if (k < 147) {
                const int c = k / 49, ky = (k / 7) % 7, kx = k % 7;
                const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v[e] = x[(((int64_t)img * 3 + c) * H + iy) * W + ix];
            }
        }
        st8(col + i * 8, v);
    }
}

// map = sigmoid(y[m*ldy]) ; xc = x*map ; xu = x*(1-map)        (one wave per pixel row)
__global__ __launch_bounds__(256) void gate_apply_b16_kernel(const __bf16* __restrict__ y, int ldy,
                                                             const __bf16* __restrict__ x, float* __restrict__ cmap,
                                                             __bf16* __restrict__ xc, __bf16* __restrict__ xu, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float g = sigmoidf_((float)y[(int64_t)m * ldy]);
    if (lane == 0) cmap[m] = g;
    const float gu = 1.f - g;
    for (int c = lane * 8; c < C; c += 512) {
        const f32x8 v = ld8(x + (int64_t)m * C + c);
        st8(xc + (int64_t)m * C + c, v * g);
        st8(xu + (int64_t)m * C + c, v * gu);
    }
}

// dx (+)= dxc*map + dxu*(1-map);  dy[m*ldy] = map(1-map) * sum_c (dxc-dxu)*x  (other dy columns untouched)
__global__ __launch_bounds__(256) void gate_bwd_b16_kernel(const __bf16* __restrict__ dxc, const __bf16* __restrict__ dxu,
                                                           const __bf16* __restrict__ x, const float* __restrict__ cmap,
                                                           __bf16* __restrict__ dx, int accumulate, __bf16* __restrict__ dy,
                                                           int ldy, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float g = cmap[m], gu = 1.f - g;
    float s = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const int64_t o = (int64_t)m * C + c;
        const f32x8 a = ld8(dxc + o), b = ld8(dxu + o), v = ld8(x + o);
        const f32x8 t = (a - b) * v;
        s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        f32x8 d = a * g + b * gu;
        if (accumulate) d += ld8(dx + o);
        st8(dx + o, d);
    }
    s = wave_sum(s);
    if (lane == 0) dy[(int64_t)m * ldy] = (__bf16)(s * g * gu);
}

// dst[m][c] (+)= v[m / rpg][c] * scale      (v fp32 per-group vector, or a bf16 tensor: temporal-mean backward)
template <bool VB16>
__global__ void add_rowbcast_b16_kernel(__bf16* __restrict__ dst, const void* __restrict__ v, int64_t C8, int64_t rpg,
                                        float scale, int accumulate, int64_t total8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / C8, c = i - m * C8;
        const int64_t j = ((m / rpg) * C8 + c) * 8;
        f32x8 d = (VB16 ? ld8(reinterpret_cast<const __bf16*>(v) + j) : ld8f(reinterpret_cast<const float*>(v) + j)) * scale;
        if (accumulate) d += ld8(dst + i * 8);
        st8(dst + i * 8, d);
    }
}

// d = mean_px (f1-f2)^2 backward: df1[b][r][c] = 2 (f1-f2) dd[b][c] / rows ; df2 (+)= -df1
__global__ void sqdiff_bwd_b16_kernel(const __bf16* __restrict__ f1, const __bf16* __restrict__ f2,
                                      const float* __restrict__ dd, __bf16* __restrict__ df1, __bf16* __restrict__ df2,
                                      int rows, int C8, int64_t f2_stride8, int acc2, int64_t total8) {
    const float k = 2.f / rows;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % C8, r = (i / C8) % rows, b = i / ((int64_t)C8 * rows);
        const int64_t j = b * f2_stride8 + r * C8 + c;
        const f32x8 g = (ld8(f1 + i * 8) - ld8(f2 + j * 8)) * ld8f(dd + (b * C8 + c) * 8) * k;
        st8(df1 + i * 8, g);
        f32x8 h = -g;
        if (acc2) h += ld8(df2 + j * 8);
        st8(df2 + j * 8, h);
    }
}

__global__ void cast_f32_b16_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x)
        st8f(y + i * 8, ld8(x + i * 8));
}

inline int lpr_for(int C) {         // lanes per row of the column reductions: 8 channels per lane, at most 32 lanes
    int l = C / 8;
    if (l > 32) l = 32;
    int p = 1;
    while (p * 2 <= l) p *= 2;
    return p;
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)
#define B16(p) reinterpret_cast<__bf16*>(p)
#define CB16(p) reinterpret_cast<const __bf16*>(p)
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int grl_bn_apply_centered_bf16(const void* z, const float* mean, const float* scale, const float* beta,
                                          const void* res, void* y, int64_t M, int C, int relu, uint8_t* relu_bits,
                                          void* stream) {
    GRL_REQUIRE(z && mean && scale && y && M > 0 && C % 8 == 0 && al16(z) && al16(y) && al16(res), "bn_apply_centered_bf16: bad args");
    const int64_t total8 = M * C / 8;
    hipLaunchKernelGGL(bn_apply_centered_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, CB16(z),
                       mean, scale, beta, CB16(res), B16(y), C / 8, total8, relu, relu_bits);
    return grl_check_launch("grl_bn_apply_centered_bf16");
}

extern "C" int grl_col_stats_bf16(const void* x, float* slab, int M, int C, int ld, const float* pivot, void* stream) {
    GRL_REQUIRE(x && slab && M > 0 && C % 8 == 0 && ld % 8 == 0 && al16(x), "col_stats_bf16: bad args");
    hipLaunchKernelGGL(col_stats_b16_kernel, dim3(grl_ceil_div(C, 256), grl_col_stats_rows(M)), dim3(256), 0,
                       (hipStream_t)stream, CB16(x), slab, M, C, ld, pivot, lpr_for(C));
    return grl_check_launch("grl_col_stats_bf16");
}

extern "C" int grl_bn_bwd_bf16(const void* dy, const void* z, const void* act, const float* mean, const float* invstd,
                               const float* gamma, void* dz, float* dgamma, float* dbeta, float* slab_ws, float* coef_ws,
                               int M, int C, void* gres, int gres_accumulate, const float* mask_scale,
                               const float* mask_beta, const uint8_t* relu_bits, void* stream) {
    GRL_REQUIRE(dy && z && mean && invstd && dz && slab_ws && coef_ws && M > 0 && C % 8 == 0, "bn_bwd_bf16: bad args");
    GRL_REQUIRE(al16(dy) && al16(z) && al16(act) && al16(dz) && al16(gres), "bn_bwd_bf16: 16-byte aligned tensors");
    const int rows = grl_col_stats_rows(M);
    hipStream_t s = (hipStream_t)stream;
    const bool inplace = (act || relu_bits) && gres == dy && !gres_accumulate;       // (as grl_bn_bwd: dy becomes the masked gradient)
    hipLaunchKernelGGL(bn_bwd_reduce_b16_kernel, dim3(grl_ceil_div(C, 256), rows), dim3(256), 0, s, CB16(dy), CB16(z),
                       CB16(act), mean, invstd, slab_ws, M, C, mask_scale, mask_beta, lpr_for(C),
                       inplace ? const_cast<__bf16*>(CB16(dy)) : (__bf16*)nullptr, relu_bits);
    if (grl_bn_finapply_takes(rows, C)) {        // round 6: finalize inside the apply pass (train_bnfuse.hip), bit-identical
        if (inplace)
            return grl_launch_bn_bwd_finapply(1, slab_ws, rows, C, (double)M, dgamma, dbeta, dy, z, nullptr, mean, invstd, gamma, dz, M,
                                              nullptr, 0, nullptr, nullptr, nullptr, s);
        return grl_launch_bn_bwd_finapply(1, slab_ws, rows, C, (double)M, dgamma, dbeta, dy, z, act, mean, invstd, gamma, dz, M, gres,
                                          gres_accumulate, mask_scale, mask_beta, relu_bits, s);
    }
    if (int e = grl_launch_bn_bwd_finalize(slab_ws, rows, C, (double)M, dgamma, dbeta, coef_ws, s)) return e;
    const int64_t total8 = (int64_t)M * C / 8;
    if (inplace)
        hipLaunchKernelGGL(bn_bwd_apply_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, s, CB16(dy), CB16(z), (const __bf16*)nullptr,
                           mean, invstd, gamma, coef_ws, B16(dz), C, total8, (__bf16*)nullptr, 0, (const float*)nullptr,
                           (const float*)nullptr, (const uint8_t*)nullptr);
    else
        hipLaunchKernelGGL(bn_bwd_apply_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, s, CB16(dy), CB16(z), CB16(act),
                           mean, invstd, gamma, coef_ws, B16(dz), C, total8, B16(gres), gres_accumulate, mask_scale, mask_beta,
                           relu_bits);
    return grl_check_launch("grl_bn_bwd_bf16");
}

// The tail of grl_bn_bwd_bf16 when the data-gradient GEMM that completed the gradient already masked it and left the two
// column sums in `slab` (GrlGemm.bn_z on bf16 storage, round 5): finalize + apply only.
extern "C" int grl_bn_bwd_finish_bf16(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
                                      void* dz, float* dgamma, float* dbeta, const float* slab, int rows, float* coef_ws, int M,
                                      int C, void* gres, int gres_accumulate, void* stream) {
    GRL_REQUIRE(g && z && mean && invstd && dz && slab && coef_ws && rows > 0 && M > 0 && C % 8 == 0, "bn_bwd_finish_bf16: bad args");
    GRL_REQUIRE(al16(g) && al16(z) && al16(dz) && al16(gres), "bn_bwd_finish_bf16: 16-byte aligned tensors");
    hipStream_t s = (hipStream_t)stream;
    // g is masked already: no activation, no mask recomputation; gres == g (the residual adopts the buffer) needs nothing
    __bf16* const gres2 = gres == g ? nullptr : B16(gres);
    if (grl_bn_finapply_takes(rows, C))
        return grl_launch_bn_bwd_finapply(1, slab, rows, C, (double)M, dgamma, dbeta, g, z, nullptr, mean, invstd, gamma, dz, M, gres2,
                                          gres2 ? gres_accumulate : 0, nullptr, nullptr, nullptr, s);
    if (int e = grl_launch_bn_bwd_finalize(slab, rows, C, (double)M, dgamma, dbeta, coef_ws, s)) return e;
    const int64_t total8 = (int64_t)M * C / 8;
    hipLaunchKernelGGL(bn_bwd_apply_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, s, CB16(g), CB16(z), (const __bf16*)nullptr,
                       mean, invstd, gamma, coef_ws, B16(dz), C, total8, gres2, gres2 ? gres_accumulate : 0, (const float*)nullptr,
                       (const float*)nullptr, (const uint8_t*)nullptr);
    return grl_check_launch("grl_bn_bwd_finish_bf16");
}

extern "C" int grl_relu_bwd_bf16(const void* dy, const void* act, void* out, int64_t n, int accumulate, void* stream) {
    GRL_REQUIRE(dy && out && n > 0 && n % 8 == 0 && al16(dy) && al16(act) && al16(out), "relu_bwd_bf16: bad args");
    hipLaunchKernelGGL(relu_bwd_b16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, CB16(dy), CB16(act),
                       B16(out), n / 8, accumulate);
    return grl_check_launch("grl_relu_bwd_bf16");
}

extern "C" int grl_axpby_bf16(const void* a, const void* b, void* y, float alpha, float beta, int64_t n, void* stream) {
    GRL_REQUIRE(a && y && n > 0 && n % 8 == 0 && al16(a) && al16(b) && al16(y), "axpby_bf16: bad args");
    hipLaunchKernelGGL(axpby_b16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, CB16(a), CB16(b), B16(y),
                       alpha, beta, n / 8);
    return grl_check_launch("grl_axpby_bf16");
}

extern "C" int grl_axpy_strided_bf16(void* dst, int64_t dst_stride, const void* src, int64_t src_stride, int nb,
                                     int64_t inner, float alpha, int accumulate, void* stream) {
    GRL_REQUIRE(dst && src && nb > 0 && inner > 0 && inner % 8 == 0 && dst_stride % 8 == 0 && src_stride % 8 == 0 &&
                al16(dst) && al16(src), "axpy_strided_bf16: bad args");
    const int64_t total8 = (int64_t)nb * inner / 8;
    hipLaunchKernelGGL(axpy_strided_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, B16(dst),
                       dst_stride / 8, CB16(src), src_stride / 8, inner / 8, alpha, accumulate, total8);
    return grl_check_launch("grl_axpy_strided_bf16");
}

extern "C" int grl_dilate2_bf16(const void* dz, void* up, int n, int Ho, int Wo, int H, int W, int C, int accumulate,
                                int oy_off, int ox_off, void* stream) {
    GRL_REQUIRE(dz && up && n > 0 && C % 8 == 0 && al16(dz) && al16(up), "dilate2_bf16: bad args");
    GRL_REQUIRE((oy_off == 0 || oy_off == 1) && (ox_off == 0 || ox_off == 1), "dilate2_bf16: offsets are 0 or 1");
    const int64_t total8 = (int64_t)n * H * W * (C / 8);
    hipLaunchKernelGGL(dilate2_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, CB16(dz), B16(up), Ho,
                       Wo, H, W, C / 8, total8, accumulate, oy_off, ox_off);
    return grl_check_launch("grl_dilate2_bf16");
}

extern "C" int grl_maxpool3x3s2_bwd_bf16(const void* x, const void* dy, void* dx, int n, int H, int W, int C, void* stream) {
    GRL_REQUIRE(x && dy && dx && n > 0 && C % 8 == 0 && H < 32768 && W < 32768, "maxpool_bwd_bf16: bad args");
    const int64_t total8 = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, CB16(x), CB16(dy),
                       B16(dx), H, W, C / 8, total8);
    return grl_check_launch("grl_maxpool3x3s2_bwd_bf16");
}

extern "C" int grl_bn_relu_maxpool3x3s2_bf16(const void* z, const float* mean, const float* scale, const float* beta, void* y,
                                             uint8_t* idx, int n, int H, int W, int C, void* stream) {
    GRL_REQUIRE(z && mean && scale && y && idx && n > 0 && C % 8 == 0 && al16(z) && al16(y) && ((uintptr_t)idx & 7) == 0,
                "bn_relu_maxpool_bf16: bad args");
    const int64_t total = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(bn_relu_maxpool_b16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, CB16(z), mean, scale,
                       beta, B16(y), idx, n, H, W, C);
    return grl_check_launch("grl_bn_relu_maxpool3x3s2_bf16");
}

extern "C" int grl_maxpool3x3s2_bwd_idx_bf16(const uint8_t* idx, const void* dy, void* dx, int n, int H, int W, int C,
                                             void* stream) {
    GRL_REQUIRE(idx && dy && dx && n > 0 && C % 8 == 0 && al16(dy) && al16(dx) && ((uintptr_t)idx & 7) == 0,
                "maxpool_bwd_idx_bf16: bad args");
    const int64_t total8 = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd_idx_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, idx, CB16(dy),
                       B16(dx), H, W, C / 8, total8);
    return grl_check_launch("grl_maxpool3x3s2_bwd_idx_bf16");
}

extern "C" int grl_stem_im2col_bf16(const float* x, void* col, int n, int H, int W, int Kp, void* stream) {
    GRL_REQUIRE(x && col && n > 0 && Kp >= 147 && Kp % 32 == 0, "stem_im2col_bf16: bad args");
    const int64_t total8 = (int64_t)n * (H / 2) * (W / 2) * (Kp / 8);
    hipLaunchKernelGGL(stem_im2col_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, x, B16(col), H, W,
                       Kp, total8);
    return grl_check_launch("grl_stem_im2col_bf16");
}

extern "C" int grl_gate_apply_bf16(const void* y, int ldy, const void* x, float* cmap, void* xc, void* xu, int M, int C,
                                   void* stream) {
    GRL_REQUIRE(y && x && cmap && xc && xu && M > 0 && C % 8 == 0, "gate_apply_bf16: bad args");
    hipLaunchKernelGGL(gate_apply_b16_kernel, dim3(grl_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, CB16(y), ldy,
                       CB16(x), cmap, B16(xc), B16(xu), M, C);
    return grl_check_launch("grl_gate_apply_bf16");
}

extern "C" int grl_gate_bwd_bf16(const void* dxc, const void* dxu, const void* x, const float* cmap, void* dx,
                                 int accumulate, void* dy, int ldy, int M, int C, void* stream) {
    GRL_REQUIRE(dxc && dxu && x && cmap && dx && dy && M > 0 && C % 8 == 0, "gate_bwd_bf16: bad args");
    hipLaunchKernelGGL(gate_bwd_b16_kernel, dim3(grl_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, CB16(dxc), CB16(dxu),
                       CB16(x), cmap, B16(dx), accumulate, B16(dy), ldy, M, C);
    return grl_check_launch("grl_gate_bwd_bf16");
}

extern "C" int grl_add_rowbcast_bf16(void* dst, const void* v, int64_t M, int64_t C, int64_t rows_per_group, float scale,
                                     int accumulate, int v_is_bf16, void* stream) {
    GRL_REQUIRE(dst && v && M > 0 && C > 0 && C % 8 == 0 && rows_per_group > 0 && al16(dst) && al16(v), "add_rowbcast_bf16: bad args");
    const int64_t total8 = M * C / 8;
    if (v_is_bf16)
        hipLaunchKernelGGL(add_rowbcast_b16_kernel<true>, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, B16(dst), v,
                           C / 8, rows_per_group, scale, accumulate, total8);
    else
        hipLaunchKernelGGL(add_rowbcast_b16_kernel<false>, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, B16(dst), v,
                           C / 8, rows_per_group, scale, accumulate, total8);
    return grl_check_launch("grl_add_rowbcast_bf16");
}

extern "C" int grl_sqdiff_bwd_bf16(const void* f1, const void* f2, const float* dd, void* df1, void* df2, int b, int rows,
                                   int C, int64_t f2_clip_stride, int accumulate_df2, void* stream) {
    GRL_REQUIRE(f1 && f2 && dd && df1 && df2 && b > 0 && rows > 0 && C % 8 == 0 && f2_clip_stride % 8 == 0, "sqdiff_bwd_bf16: bad args");
    const int64_t total8 = (int64_t)b * rows * C / 8;
    hipLaunchKernelGGL(sqdiff_bwd_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, CB16(f1), CB16(f2), dd,
                       B16(df1), B16(df2), rows, C / 8, f2_clip_stride / 8, accumulate_df2, total8);
    return grl_check_launch("grl_sqdiff_bwd_bf16");
}

extern "C" int grl_cast_f32(const void* x, float* y, int64_t n, void* stream) {
    GRL_REQUIRE(x && y && n > 0 && n % 8 == 0 && al16(x) && al16(y), "cast_f32: n % 8 == 0, aligned");
    hipLaunchKernelGGL(cast_f32_b16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, CB16(x), y, n / 8);
    return grl_check_launch("grl_cast_f32");
}
