// bf16-STORAGE variants of the bandwidth-bound kernels (BASELINE configs[2] pipeline,
// engine math mode 'bf16s'): activations are bf16 in HBM, every lane moves 16 bytes
// (8 channels), arithmetic and all reductions are fp32, per-channel / per-clip vectors
// stay fp32.  Same semantics and reference call sites as their fp32 twins in pointwise.hip.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

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
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

__global__ void cast_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        st8(y + i * 8, v);
    }
}

__global__ void maxpool_b16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int n, int H,
                                   int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C8 = C >> 3;
    const int64_t total = (int64_t)n * Ho * Wo * C8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = i % C8;
        int64_t r = i / C8;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int img = r / Ho;
        f32x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x8 v = ld8(x + (((int64_t)img * H + iy) * W + ix) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        st8(y + i * 8, m);
    }
}

// y[g][c] (+)= mul * sum_r x[g][r][c]   (bf16 in, fp32 out); 32 lanes x 8 channels = 256-channel slab
__global__ __launch_bounds__(256) void group_mean_b16_kernel(const __bf16* __restrict__ x,
                                                             float* __restrict__ y, int rows, int C,
                                                             int ldy, float mul, int accumulate) {
    __shared__ f32x8 red[8][32];
    const int g = blockIdx.y, sub = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + sub * 8;
    f32x8 s;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (c < C) {
        const __bf16* xp = x + (int64_t)g * rows * C + c;
#pragma unroll 8
        for (int r = part; r < rows; r += 8) s += ld8(xp + (int64_t)r * C);
    }
    red[part][sub] = s;
    __syncthreads();
    if (part == 0 && c < C) {
        f32x8 t = red[0][sub];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][sub];
        float* yp = y + (int64_t)g * ldy + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) yp[e] = (accumulate ? yp[e] : 0.f) + t[e] * mul;
    }
}

__global__ __launch_bounds__(256) void sqdiff_mean_b16_kernel(const __bf16* __restrict__ f1,
                                                              const __bf16* __restrict__ f2,
                                                              float* __restrict__ d, int rows, int C,
                                                              int64_t f2_stride) {
    __shared__ f32x8 red[8][32];
    const int g = blockIdx.y, sub = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + sub * 8;
    f32x8 s;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (c < C) {
        const __bf16* p1 = f1 + (int64_t)g * rows * C + c;
        const __bf16* p2 = f2 + (int64_t)g * f2_stride + c;
#pragma unroll 8
        for (int r = part; r < rows; r += 8) {
            const f32x8 t = ld8(p1 + (int64_t)r * C) - ld8(p2 + (int64_t)r * C);
            s += t * t;
        }
    }
    red[part][sub] = s;
    __syncthreads();
    if (part == 0 && c < C) {
        f32x8 t = red[0][sub];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][sub];
        const float inv = 1.f / rows;
#pragma unroll
        for (int e = 0; e < 8; ++e) d[(int64_t)g * C + c + e] = t[e] * inv;
    }
}

// GCE gate, one wave per pixel row: map = sigmoid(bn(h[m] . w3)); xc = x*map; xu = x*(1-map)
__global__ __launch_bounds__(256) void gce_gate_b16_kernel(
    const __bf16* __restrict__ h, const float* __restrict__ w3, const float* __restrict__ bsc,
    const float* __restrict__ bsh, const __bf16* __restrict__ x, float* __restrict__ cmap,
    __bf16* __restrict__ xc, __bf16* __restrict__ xu, int M, int Ch, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float s = 0.f;
    for (int k = lane * 8; k < Ch; k += 512) {
        const f32x8 a = ld8(h + (int64_t)m * Ch + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += a[e] * w3[k + e];
    }
    s = wave_sum(s);
    const float g = sigmoidf_(s * bsc[0] + bsh[0]);
    if (lane == 0 && cmap) cmap[m] = g;
    const float gu = 1.f - g;
    for (int c = lane * 8; c < C; c += 512) {
        const f32x8 v = ld8(x + (int64_t)m * C + c);
        st8(xc + (int64_t)m * C + c, v * g);
        st8(xu + (int64_t)m * C + c, v * gu);
    }
}

__global__ void temporal_mean_b16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int T,
                                         int64_t inner8, int64_t total8) {
    const float inv = 1.f / T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / inner8, r = i - b * inner8;
        const __bf16* xp = x + (b * T * inner8 + r) * 8;
        f32x8 s = ld8(xp);
        for (int t = 1; t < T; ++t) s += ld8(xp + t * inner8 * 8);
        st8(y + i * 8, s * inv);
    }
}

__global__ void add_strided_b16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ bsrc,
                                       __bf16* __restrict__ y, int64_t inner8, int64_t bstride8,
                                       int64_t total8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / inner8, r = i - b * inner8;
        st8(y + i * 8, ld8(a + i * 8) + ld8(bsrc + (b * bstride8 + r) * 8));
    }
}

// ---------------------------------------------------------------------------------
// Stem for the bf16-storage pipeline: 7x7/s2 conv + folded BN + ReLU on the bf16 MFMA.  One workgroup =
// 8 x 16 output pixels x 64 channels.  The reduction index is ordered (channel, ky, kx) with kx padded
// from 7 to 8 taps (zero weight): K = 3 * 7 * 8 = 168 (+ one zero chunk = 176 = 11 MFMA steps), so that a
// pixel's eight taps of one (channel, ky) are eight CONSECUTIVE input pixels.  The NCHW patch (3 x 21 x 37)
// is staged in LDS already rounded to bf16; the im2col tile [128 px][176 k] is then built from plain
// 16-byte copies -- no per-element index arithmetic, no conversion -- lanes walking pixels (conflict-free
// reads and writes); rows are padded to 368 bytes so that the ds_read_b128 fragment reads hit distinct
// bank slots.  (Round 1 ordered k as (c, ky, kx) unpadded, K = 147 -> 160, and gathered element by
// element with k/49, k/7, k%7 per element: 0.96 ms per 512 frames, VALU-bound at 4 % MFMA busy.)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SB_TH = 8, SB_TW = 16, SB_PH = 2 * SB_TH + 5, SB_PW = 2 * SB_TW + 5, SB_PWP = SB_PW + 1;
constexpr int SB_CH = 22;                                      // 8-wide k chunks: 21 (channel, ky) rows + one of zeros
constexpr int SB_K = SB_CH * 8, SB_ROWB = SB_K * 2 + 16;       // 176 k; bytes per bf16 row (352 + 16 pad)
constexpr int SB_PATCH = 3 * SB_PH * SB_PWP;                   // bf16 cells (column SB_PW of every row is a zero pad)

constexpr int SB_TPW = 4;                             // y-tiles one workgroup walks (weights staged once, round 3)

__global__ __launch_bounds__(256) void stem_b16_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
    const float* __restrict__ shift, __bf16* __restrict__ y, int H, int W, int relu,
    const __bf16* __restrict__ wp, const float* __restrict__ norm) {
    // norm != NULL: x holds raw u8 pixels, normalised here as (u/255 - mean[c]) / std[c]
    // One workgroup walks SB_TPW tiles down its column strip: the 23.5 KB weight image is staged once (the epilogue's
    // fp32 staging overlays the im2col tile only), and the NEXT tile's patch is requested before this tile's im2col /
    // MFMA / epilogue phases, which used to wait for it with nothing else to do (two workgroups per CU).
    extern __shared__ __attribute__((aligned(16))) char smb[];
    char* At = smb;                                            // [128][368 B]
    char* Wt = At + 128 * SB_ROWB;                             // [64][368 B]
    __bf16* patch = reinterpret_cast<__bf16*>(Wt + 64 * SB_ROWB);   // [3][21][38]
    float* Cs = reinterpret_cast<float*>(smb);                 // epilogue staging [128][64] fp32 (32 KB < At)
    const int Ho = H >> 1, Wo = W >> 1;
    const int img = blockIdx.z, ox0 = blockIdx.x * SB_TW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ix0 = ox0 * 2 - 3;
    const float* xi = x + (int64_t)img * 3 * H * W;
    const uint8_t* xu = reinterpret_cast<const uint8_t*>(x) + (int64_t)img * 3 * H * W;
    constexpr int P_IT = (SB_PATCH + 255) / 256, W_IT = (64 * SB_ROWB / 16 + 255) / 256;
    float pv[P_IT];
    // (the u8 / fp32 choice is made OUTSIDE the loop: with `norm ? xu[o] : xi[o]` inside it every iteration was a
    //  branch, and hipcc -- which counts vmcnt per basic block -- waited vmcnt(0) after each load: P_IT exposed
    //  latencies per tile instead of one.  Round 5.)
    auto load_patch = [&](const int oy0) {
        const int iy0 = oy0 * 2 - 3;
        auto body = [&](auto u8_) {
            constexpr bool U8 = decltype(u8_)::value;
#pragma unroll
            for (int it = 0; it < P_IT; ++it) {
                const int i = tid + it * 256;
                const int cr = i / SB_PWP, q = i - cr * SB_PWP, c = cr / SB_PH, r = cr - c * SB_PH;
                const int iy = iy0 + r, ix = ix0 + q;
                const bool ok = i < SB_PATCH && q < SB_PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int64_t o = ok ? ((int64_t)c * H + iy) * W + ix : 0;
                float v;
                if constexpr (U8) v = ((float)xu[o] / 255.f - norm[c < 3 ? c : 0]) / norm[3 + (c < 3 ? c : 0)];
                else v = xi[o];
                pv[it] = ok ? v : 0.f;
            }
        };
        if (norm) body(std::true_type{});
        else body(std::false_type{});
    };
    int oy0 = blockIdx.y * (SB_TPW * SB_TH);
    load_patch(oy0);
    // weights: the LDS image [64][368 B] made once by grl_stem_pack_weight_bf16, or converted here
    if (wp) {
        bf16x8 wv[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            wv[it] = reinterpret_cast<const bf16x8*>(wp)[i < 64 * SB_ROWB / 16 ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            if (i < 64 * SB_ROWB / 16) reinterpret_cast<bf16x8*>(Wt)[i] = wv[it];
        }
    } else {
        for (int i = tid; i < 64 * SB_CH; i += 256) {
            const int n = i / SB_CH, ch = i - n * SB_CH;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((ch < 21 && e < 7) ? w[n * 147 + ch * 7 + e] : 0.f);
            *reinterpret_cast<bf16x8*>(Wt + n * SB_ROWB + ch * 16) = o;
        }
    }
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, fhalf = lane >> 5;
    const int col_l = lane & 31;
    const int c8 = (tid & 7) * 8;
    f32x8 sc, sh;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[c8 + e]; sh[e] = shift[c8 + e]; }
    for (int t = 0; t < SB_TPW && oy0 < Ho; ++t, oy0 += SB_TH) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int i = tid + it * 256;
            if (i < SB_PATCH) patch[i] = (__bf16)pv[it];
        }
        __syncthreads();                                       // patch (and, first time, weights) visible; Cs readers of the previous tile done
        if (t + 1 < SB_TPW && oy0 + SB_TH < Ho) load_patch(oy0 + SB_TH);      // in flight under everything below
        // im2col in LDS: item = (k chunk, pixel), pixels fastest: the chunk (channel, ky) is wave-uniform and a
        // pixel's eight taps are eight consecutive bf16 of one patch row (4-byte aligned: four ds_read_b32)
        for (int i = tid; i < SB_CH * 128; i += 256) {
            const int ch = i >> 7, m = i & 127;
            uint32_t o[4] = {0u, 0u, 0u, 0u};
            if (ch < 21) {
                const int c = ch / 7, ky = ch - 7 * c;
                const uint32_t* s2 = reinterpret_cast<const uint32_t*>(
                    patch + ((c * SB_PH) + 2 * (m / SB_TW) + ky) * SB_PWP + 2 * (m % SB_TW));
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = s2[e];
            }
            *reinterpret_cast<uint4*>(At + m * SB_ROWB + ch * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < SB_K / 16; ++s) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Wt + (wn * 32 + frow) * SB_ROWB + (2 * s + fhalf) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(At + (wm * 64 + i * 32 + frow) * SB_ROWB + (2 * s + fhalf) * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();                                       // At is dead: reuse as fp32 C staging
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * 64 + wn * 32 + col_l] = acc[i][r];
        __syncthreads();
        // 128 pixels x 64 channels: 8 lanes x 8 channels per pixel row, 32 rows per pass
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = it * 32 + (tid >> 3);
            const int oy = oy0 + m / SB_TW, ox = ox0 + m % SB_TW;
            if (oy < Ho && ox < Wo) {
                f32x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float tv = Cs[m * 64 + c8 + e] * sc[e] + sh[e];
                    v[e] = (tv > 0.f || !relu) ? tv : 0.f;
                }
                st8(y + (((int64_t)img * Ho + oy) * Wo + ox) * 64 + c8, v);
            }
        }
        // (the next iteration's first barrier comes after its patch writes, which touch neither At / Cs nor Wt; the
        // im2col that overwrites Cs follows that barrier)
    }
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)
#define B16(p) reinterpret_cast<__bf16*>(p)
#define CB16(p) reinterpret_cast<const __bf16*>(p)

extern "C" int grl_cast_bf16(const float* x, void* y, int64_t n, void* stream) {
    GRL_REQUIRE(x && y && n > 0 && n % 8 == 0, "cast_bf16: n % 8");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, B16(y), n / 8);
    return grl_check_launch("grl_cast_bf16");
}

__global__ void stem_pack_weight_b16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // [64][184] bf16 (368-byte rows), k = (c, ky, kx of 8)
    if (i >= 64 * (SB_ROWB / 2)) return;
    const int n = i / (SB_ROWB / 2), k = i - n * (SB_ROWB / 2), ch = k >> 3, kx = k & 7;
    wp[i] = (__bf16)((ch < 21 && kx < 7) ? w[n * 147 + ch * 7 + kx] : 0.f);
}

extern "C" int grl_stem_pack_weight_bf16(const float* w, void* wp, void* stream) {
    GRL_REQUIRE(w && wp, "stem_pack_weight_bf16: null");
    hipLaunchKernelGGL(stem_pack_weight_b16_kernel, dim3(grl_ceil_div(64 * (SB_ROWB / 2), 256)), dim3(256), 0,
                       (hipStream_t)stream, w, B16(wp));
    return grl_check_launch("grl_stem_pack_weight_bf16");
}

static int stem_b16_launch(const float* x, const float* norm, const float* w, const float* scale,
                           const float* shift, void* y, int n, int H, int W, int relu, const void* wp, void* stream);

extern "C" int grl_stem_conv7x7_bf16(const float* x, const float* w, const float* scale, const float* shift,
                                     void* y, int n, int H, int W, int relu, const void* wp, void* stream) {
    return stem_b16_launch(x, nullptr, w, scale, shift, y, n, H, W, relu, wp, stream);
}

extern "C" int grl_stem_conv7x7_u8_bf16(const uint8_t* x, const float* mean_std, const float* w, const float* scale,
                                        const float* shift, void* y, int n, int H, int W, int relu, const void* wp,
                                        void* stream) {
    if (!mean_std) return grl_fail(GRL_EINVAL, "stem_u8_bf16: mean_std is null");
    return stem_b16_launch(reinterpret_cast<const float*>(x), mean_std, w, scale, shift, y, n, H, W, relu, wp,
                           stream);
}

static int stem_b16_launch(const float* x, const float* norm, const float* w, const float* scale,
                           const float* shift, void* y, int n, int H, int W, int relu, const void* wp, void* stream) {
    GRL_REQUIRE(x && w && scale && shift && y && n > 0, "stem_bf16: null/empty");
    GRL_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem_bf16: H and W must be even");
    const int Ho = H / 2, Wo = W / 2;
    const size_t lds = (size_t)(128 + 64) * SB_ROWB + (size_t)SB_PATCH * sizeof(__bf16);
    if (lds > 65536)
        (void)hipFuncSetAttribute((const void*)stem_b16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stem_b16_kernel, dim3(grl_ceil_div(Wo, SB_TW), grl_ceil_div(Ho, SB_TH * SB_TPW), n), dim3(256), lds,
                       (hipStream_t)stream, x, w, scale, shift, B16(y), H, W, relu, CB16(wp), norm);
    return grl_check_launch("grl_stem_conv7x7_bf16");
}

// ---------------------------------------------------------------------------------
// Stem + 3x3 / stride-2 max-pool in ONE launch (round 4; eval, bf16 storage; resnets1.py:101-104, basebranch.py:27-36):
// the post-ReLU stem map (537 MB per 512 frames) is never written nor re-read.  Same MFMA core as stem_b16_kernel with a
// tile of 2 stem rows x 64 columns = the FULL width of a 256 x 128 frame's stem map, so a pooling window never crosses a
// tile's left / right edge; a workgroup walks SP_TPW tiles DOWN its strip and every tile yields exactly one pooled row
// from (the previous tile's last row -- kept in REGISTERS by the lane that will need it --, row 0, row 1).  A strip that
// does not start at the top of the image first runs one warm-up tile (the two rows above it) that only fills the carry.
// Rounding commutes with max (round-to-nearest is monotonic), so the result equals max-pooling the bf16 stem map bit for bit.
constexpr int SP_TH = 2, SP_TW = 64, SP_PH = 2 * SP_TH + 5, SP_PW = 2 * SP_TW + 5, SP_PWP = SP_PW + 1;
constexpr int SP_PATCH = 3 * SP_PH * SP_PWP;
constexpr int SP_TPW = 16;                            // tiles (= pooled rows) per workgroup

__global__ __launch_bounds__(256) void stem_pool_b16_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, __bf16* __restrict__ y,
    int H, int W, const __bf16* __restrict__ wp, const float* __restrict__ norm) {
    extern __shared__ __attribute__((aligned(16))) char smb[];
    char* At = smb;                                            // [128][368 B]
    char* Wt = At + 128 * SB_ROWB;                             // [64][368 B]
    __bf16* patch = reinterpret_cast<__bf16*>(Wt + 64 * SB_ROWB);   // [3][9][134]
    float* Cs = reinterpret_cast<float*>(smb);                 // epilogue staging [128][64] fp32 (32 KB < At)
    const int Ho = H >> 1, Hp = Ho >> 1, Wp = SP_TW / 2;
    const int img = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xi = x + (int64_t)img * 3 * H * W;
    const uint8_t* xu = reinterpret_cast<const uint8_t*>(x) + (int64_t)img * 3 * H * W;
    constexpr int P_IT = (SP_PATCH + 255) / 256, W_IT = (64 * SB_ROWB / 16 + 255) / 256;
    float pv[P_IT];
    // (the u8 / fp32 choice is made OUTSIDE the loop: with `norm ? xu[o] : xi[o]` inside it every iteration was a
    //  branch, and hipcc -- which counts vmcnt per basic block -- waited vmcnt(0) after each load: P_IT exposed
    //  latencies per tile instead of one.  Round 5.)
    auto load_patch = [&](const int oy0) {
        const int iy0 = oy0 * 2 - 3;
        auto body = [&](auto u8_) {
            constexpr bool U8 = decltype(u8_)::value;
#pragma unroll
            for (int it = 0; it < P_IT; ++it) {
                const int i = tid + it * 256;
                const int cr = i / SP_PWP, q = i - cr * SP_PWP, c = cr / SP_PH, r = cr - c * SP_PH;
                const int iy = iy0 + r, ix = q - 3;
                const bool ok = i < SP_PATCH && q < SP_PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const int64_t o = ok ? ((int64_t)c * H + iy) * W + ix : 0;
                float v;
                if constexpr (U8) v = ((float)xu[o] / 255.f - norm[c < 3 ? c : 0]) / norm[3 + (c < 3 ? c : 0)];
                else v = xi[o];
                pv[it] = ok ? v : 0.f;
            }
        };
        if (norm) body(std::true_type{});
        else body(std::false_type{});
    };
    const int strip0 = blockIdx.x * (SP_TPW * SP_TH);           // first stem row whose pooled row this workgroup emits
    int oy0 = strip0 > 0 ? strip0 - SP_TH : 0;                  // (warm-up tile above the strip)
    load_patch(oy0);
    {
        bf16x8 wv[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            wv[it] = reinterpret_cast<const bf16x8*>(wp)[i < 64 * SB_ROWB / 16 ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            if (i < 64 * SB_ROWB / 16) reinterpret_cast<bf16x8*>(Wt)[i] = wv[it];
        }
    }
    const int wm = wave >> 1, wn = wave & 1, frow = lane & 31, fhalf = lane >> 5;
    const int col_l = lane & 31;
    const int c8 = (tid & 7) * 8, ppx = tid >> 3;              // pooling: this thread's pooled column and 8 channels
    f32x8 sc, sh;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[c8 + e]; sh[e] = shift[c8 + e]; }
    f32x8 carry[3];                                             // previous tile's row 1 at stem columns 2 ppx - 1, 2 ppx, 2 ppx + 1
#pragma unroll
    for (int jx = 0; jx < 3; ++jx)
#pragma unroll
        for (int e = 0; e < 8; ++e) carry[jx][e] = 0.f;        // (post-ReLU values are >= 0: a zero never wins over a real element)
    const int oy_end = min(Ho, strip0 + SP_TPW * SP_TH);
    for (; oy0 < oy_end; oy0 += SP_TH) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int i = tid + it * 256;
            if (i < SP_PATCH) patch[i] = (__bf16)pv[it];
        }
        __syncthreads();                                       // patch (and, first time, weights) visible; Cs readers of the previous tile done
        if (oy0 + SP_TH < oy_end) load_patch(oy0 + SP_TH);     // in flight under everything below
        for (int i = tid; i < SB_CH * 128; i += 256) {
            const int ch = i >> 7, m = i & 127;
            uint32_t o[4] = {0u, 0u, 0u, 0u};
            if (ch < 21) {
                const int c = ch / 7, ky = ch - 7 * c;
                const uint32_t* s2 = reinterpret_cast<const uint32_t*>(
                    patch + ((c * SP_PH) + 2 * (m / SP_TW) + ky) * SP_PWP + 2 * (m % SP_TW));
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = s2[e];
            }
            *reinterpret_cast<uint4*>(At + m * SB_ROWB + ch * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int s = 0; s < SB_K / 16; ++s) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Wt + (wn * 32 + frow) * SB_ROWB + (2 * s + fhalf) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(At + (wm * 64 + i * 32 + frow) * SB_ROWB + (2 * s + fhalf) * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();                                       // At is dead: reuse as fp32 C staging
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * 64 + wn * 32 + col_l] = acc[i][r];
        __syncthreads();
        // folded BatchNorm + ReLU in place (a thread rewrites exactly the cells it read)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = it * 32 + (tid >> 3);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float tv = Cs[m * 64 + c8 + e] * sc[e] + sh[e];
                Cs[m * 64 + c8 + e] = tv > 0.f ? tv : 0.f;
            }
        }
        __syncthreads();
        // pooled row oy0 / 2: max over (carry = stem row oy0 - 1, row 0, row 1) x stem columns 2 ppx - 1 .. 2 ppx + 1
        f32x8 mx = carry[0];
#pragma unroll
        for (int jx = 1; jx < 3; ++jx)
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = carry[jx][e] > mx[e] ? carry[jx][e] : mx[e];
#pragma unroll
        for (int jx = 0; jx < 3; ++jx) {
            const int cx = 2 * ppx - 1 + jx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float r0 = cx >= 0 ? Cs[cx * 64 + c8 + e] : 0.f;
                const float r1 = cx >= 0 ? Cs[(SP_TW + cx) * 64 + c8 + e] : 0.f;
                const float m2 = r0 > r1 ? r0 : r1;
                mx[e] = m2 > mx[e] ? m2 : mx[e];
                carry[jx][e] = r1;
            }
        }
        if (oy0 >= strip0) st8(y + (((int64_t)img * Hp + (oy0 >> 1)) * Wp + ppx) * 64 + c8, mx);
        // (the next iteration's first barrier comes after its patch writes, which touch neither At / Cs nor Wt)
    }
}

// ---------------------------------------------------------------------------------
// Stem + max-pool, second form (round 5): no im2col tile, no weight reads, no fp32 staging.  The product is formed as
// D[channel][pixel] = W . patch^T: the 64 x 176 weight image lives in REGISTERS as the A fragments of
// v_mfma_f32_32x32x16_bf16 (2 channel blocks x 11 k-steps, loaded once per wave), and the B fragment of a k-step -- a
// pixel's eight kx taps of one (channel, ky) -- is eight consecutive bf16 of the staged input row, read straight from the
// patch (two ds_read2_b32; 4-byte aligned because a stem pixel is two input pixels).  In that orientation a lane holds
// four CONSECUTIVE channels of one pixel per accumulator quad: folded BatchNorm + ReLU run in registers and the row goes
// to LDS as bf16 [pixel][channel] (8-byte writes, 16-byte chunks XOR-swizzled by the pixel pair), from where the pooling
// threads read 3 x 3 windows as 16-byte vectors and take the maximum with v_pk_max_u16 (post-ReLU bf16 values are
// non-negative: their bit patterns order like unsigned integers; rounding commutes with max).  One iteration = 4 stem rows
// (wave w: row w, both 32-column halves) = 2 pooled rows, two barriers; the input rows live in a 16-row ring per channel
// and the next iteration's 8 new rows are requested before the MFMAs and written to the ring after them.  Same k order
// and the same products as stem_pool_b16_kernel: bit-identical (tested).  512 frames: 414 -> ~150 us.
constexpr int S2_ROWB = SP_PWP * 2;                   // 268 bytes: one input row, columns -3 .. 130 as bf16
constexpr int S2_RING = 16;                           // input-row slots per channel (13 live + 8 incoming - 5 shared)
constexpr int S2_CHB = (S2_RING + 1) * S2_ROWB;       // bytes per channel (+ one row: slot 16 of channel 0 stays zero)
constexpr int S2_PATCHB = (3 * S2_CHB + 15) / 16 * 16;
constexpr int S2_ROWBUF = SP_TW * 128;                // one stem row as bf16 [64 px][64 ch]
constexpr int S2_LDS = 5 * S2_ROWBUF + S2_PATCHB + 512;       // + folded BatchNorm scale / shift (64 floats each)
#ifndef GRL_SP2_KO
#define GRL_SP2_KO 0      // timing-only knock-outs (1: no fragment reads / MFMAs, 2: no epilogue, 4: no pooling reads, 8: no input staging); wrong results
#endif
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_s2 __attribute__((ext_vector_type(4)));
typedef float f32x2_s2 __attribute__((ext_vector_type(2)));

template <bool U8>
__global__ __launch_bounds__(256, 2) void stem_pool2_b16_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, __bf16* __restrict__ y,
    int H, const __bf16* __restrict__ wp, const float* __restrict__ norm, int strip_rows) {
    extern __shared__ __attribute__((aligned(16))) char smb[];
    char* const rowbuf = smb;                                  // [5][64][128 B]
    char* const patch = smb + 5 * S2_ROWBUF;                   // [3][17][268 B]
    float* const scs = reinterpret_cast<float*>(smb + 5 * S2_ROWBUF + S2_PATCHB);     // scale[64], shift[64]
    constexpr int W = 2 * SP_TW;
    const int Ho = H >> 1, Hp = Ho >> 1, Wp = SP_TW / 2;
    const int img = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pxl = lane & 31, hf = lane >> 5;
    const float* xi = x + (int64_t)img * 3 * H * W;
    const uint8_t* xu = reinterpret_cast<const uint8_t*>(x) + (int64_t)img * 3 * H * W;
    // Input rows are addressed by ry = iy + 3 >= 0 (stem row oy, tap ky: ry = 2 oy + ky); ring slot = ry & 15.  A wave
    // stages whole rows: lane l loads the input pixels 2 l, 2 l + 1 (one 8-byte load; the loads are unconditional -- a row
    // outside the image reads row 0 and is zeroed by a select -- so hipcc can count them) and writes two bf16 cells at
    // columns 2 l + 3, 2 l + 4; the six pad cells of a row are zeroed once and never written again.
    // (the RAW load is returned; the out-of-image select and the conversion happen in store_row, after the MFMAs: a select
    //  right behind the load -- or any branch between the load and its use -- makes hipcc wait for the load on the spot)
    auto load_row = [&](int c, int ry) {
        const int iy = ry - 3;
        const bool ok = (unsigned)iy < (unsigned)H;
        const int64_t o = ((int64_t)c * H + (ok ? iy : 0)) * W + 2 * lane;
        f32x2_s2 v;
        if constexpr (U8) {
            const unsigned short u = *reinterpret_cast<const unsigned short*>(xu + o);
            v[0] = __builtin_bit_cast(float, (uint32_t)u);
            v[1] = 0.f;
        } else {
            v = *reinterpret_cast<const f32x2_s2*>(xi + o);
        }
        return v;
    };
    auto store_row = [&](int c, int ry, f32x2_s2 v) {
        const bool ok = (unsigned)(ry - 3) < (unsigned)H;
        if constexpr (U8) {
            const uint32_t u = __builtin_bit_cast(uint32_t, v[0]);
            v[0] = ((float)(u & 255u) / 255.f - norm[c]) / norm[3 + c];
            v[1] = ((float)(u >> 8) / 255.f - norm[c]) / norm[3 + c];
        }
        __bf16* const d = reinterpret_cast<__bf16*>(patch + c * S2_CHB + (ry & (S2_RING - 1)) * S2_ROWB) + 2 * lane + 3;
        d[0] = (__bf16)(ok ? v[0] : 0.f);
        d[1] = (__bf16)(ok ? v[1] : 0.f);
    };
    const int strip0 = blockIdx.x * strip_rows;                 // first stem row whose pooled rows this workgroup emits
    int oyb = strip0 > 0 ? strip0 - 4 : 0;                      // (one warm-up iteration above the strip fills the carry row)
    const int oy_end = min(Ho, strip0 + strip_rows);
    // prologue: zero patch (pads, the zero row) and carry; then the 13 input rows of the first iteration
    for (int i = tid; i < S2_PATCHB / 16; i += 256) reinterpret_cast<uint4*>(patch)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < S2_ROWBUF / 16; i += 256) reinterpret_cast<uint4*>(rowbuf + 4 * S2_ROWBUF)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 128) scs[tid] = tid < 64 ? scale[tid] : shift[tid - 64];      // (read back as broadcast 16-byte vectors: 64 VGPRs otherwise)
    __syncthreads();
    for (int rr = wave; rr < 3 * 13; rr += 4) {
        const int c = rr / 13, r = rr - 13 * c;
        store_row(c, 2 * oyb + r, load_row(c, 2 * oyb + r));
    }
    // weights: A fragments (lane: channel 32 j + pxl, k chunk 2 s + hf) of the packed image [64][368 B]
    bf16x8 wf[2][SB_CH / 2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < SB_CH / 2; ++s)
        {
            wf[j][s] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(wp) + (32 * j + pxl) * SB_ROWB + (2 * s + hf) * 16);
            asm volatile("" : "+v"(wf[j][s]));                 // (pinned: hipcc re-loaded all 22 fragments inside the row loop)
        }
    constexpr int P_IT = 6;                                     // rows per wave of the 3 x 8 new input rows of an iteration
    f32x2_s2 pv[P_IT];
    const int c8 = tid & 7, ppx = tid >> 3;                     // pooling: this thread's pooled column and 8 channels
    __syncthreads();
    for (int it = 0; oyb < oy_end; oyb += 4, ++it) {
        // the next iteration's 8 new rows x 3 channels, 6 per wave: requested here, written to the ring behind the MFMAs.
        // No branch between the two (past the last iteration the rows are loaded and stored all the same: nobody reads them)
#pragma unroll
        for (int k = 0; k < P_IT; ++k) {
            const int rr = wave + 4 * k;                       // (channel rr >> 3, new row rr & 7)
            if (!(GRL_SP2_KO & 8)) pv[k] = load_row(rr >> 3, 2 * oyb + 13 + (rr & 7));
        }
        __builtin_amdgcn_sched_barrier(0);                     // (hipcc otherwise sinks the loads below the MFMAs, next to their use)
        // row slots of this iteration: rows 0..2 -> slots 0..2, row 3 -> slot 3 (even) / 4 (odd); the carry (previous
        // iteration's row 3) is the other one of 3 / 4
        const int slot3 = 3 + (it & 1), carry = 4 - (it & 1);
        const int oy = oyb + wave;
        const int myslot = wave < 3 ? wave : slot3;
        // both 32-column halves of the row together: four independent accumulator chains, and the folded BatchNorm
        // vectors of a channel quad are read once for both
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][j][r] = 0.f;
        // fragments double-buffered in registers: the reads of k-step s + 1 go out in front of the MFMAs of step s (hipcc's own
        // order -- read, wait, four MFMAs, next read into the same registers -- exposed the LDS latency eleven times per row)
        bf16x8 bfr[2][2];
        auto read_frag = [&](int s, bf16x8 (&dst)[2]) {
            const int ch0 = 2 * s, ch1 = 2 * s + 1;
            const int off0 = (ch0 / 7) * S2_CHB + ((2 * oy + ch0 % 7) & (S2_RING - 1)) * S2_ROWB;
            const int off1 = ch1 < 21 ? (ch1 / 7) * S2_CHB + ((2 * oy + ch1 % 7) & (S2_RING - 1)) * S2_ROWB : S2_RING * S2_ROWB;
            const uint32_t* s2 = reinterpret_cast<const uint32_t*>(patch + (hf ? off1 : off0) + 4 * pxl);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = s2[32 * cb + e];
                dst[cb] = __builtin_bit_cast(bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
            }
        };
        if (!(GRL_SP2_KO & 1)) read_frag(0, bfr[0]);
#pragma unroll
        for (int s = 0; s < ((GRL_SP2_KO & 1) ? 0 : SB_CH / 2); ++s) {
            if (s + 1 < SB_CH / 2) read_frag(s + 1, bfr[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[cb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][s], bfr[s & 1][cb], acc[cb][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < ((GRL_SP2_KO & 2) ? 0 : 2); ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (this lane's channels 32 j + 8 q + 4 hf + (0..3))
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scs + 32 * j + 8 * q + 4 * hf);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(scs + 64 + 32 * j + 8 * q + 4 * hf);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int px = 32 * cb + pxl;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tv = acc[cb][j][4 * q + e] * sc[e] + sh[e];
                        v[e] = tv > 0.f ? tv : 0.f;
                    }
                    *reinterpret_cast<bf16x4_s2*>(rowbuf + myslot * S2_ROWBUF + px * 128 + 8 * hf + (((4 * j + q) ^ ((px >> 1) & 7)) << 4)) =
                        __builtin_convertvector(v, bf16x4_s2);
                }
            }
        __syncthreads();                                       // the four stem rows are in rowbuf; nobody reads the patch any more
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < P_IT; ++k) {
            const int rr = wave + 4 * k;
            if (!(GRL_SP2_KO & 8)) store_row(rr >> 3, 2 * oyb + 13 + (rr & 7), pv[k]);
        }
        // two pooled rows: (carry, row 0, row 1) and (row 1, row 2, row 3) x stem columns 2 ppx - 1 .. 2 ppx + 1
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            u16x8 mx = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rr = 0; rr < ((GRL_SP2_KO & 4) ? 0 : 3); ++rr) {
                const int slot = pr == 0 ? (rr == 0 ? carry : rr - 1) : (rr == 2 ? slot3 : rr + 1);
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int cx = 2 * ppx + dx;
                    if (cx >= 0) {
                        const u16x8 v = *reinterpret_cast<const u16x8*>(rowbuf + slot * S2_ROWBUF + cx * 128 + ((c8 ^ ((cx >> 1) & 7)) << 4));
                        mx = __builtin_elementwise_max(mx, v);
                    }
                }
            }
            if (oyb >= strip0)
                *reinterpret_cast<u16x8*>(y + (((int64_t)img * Hp + (oyb >> 1) + pr) * Wp + ppx) * 64 + c8 * 8) = mx;
        }
        __syncthreads();                                       // rowbuf rows 0..2 (+ the old carry) free, the ring holds the next rows
    }
}

extern "C" int grl_stem_pool_bf16(const void* x, int x_is_u8, const float* mean_std, const float* scale, const float* shift,
                                  void* y, int n, int H, int W, const void* wp, void* stream) {
    GRL_REQUIRE(x && scale && shift && y && wp && n > 0, "stem_pool_bf16: null/empty");
    GRL_REQUIRE(W == 2 * SP_TW && H % 4 == 0, "stem_pool_bf16: needs W == 128 and H % 4 == 0");
    GRL_REQUIRE(!x_is_u8 || mean_std, "stem_pool_bf16: u8 input needs mean_std");
    GRL_REQUIRE(!x_is_u8 || ((uintptr_t)x & 1) == 0, "stem_pool_bf16: u8 input must be 2-byte aligned (2-byte row loads)");
    GRL_REQUIRE(n <= 65535, "stem_pool_bf16: at most 65535 frames per launch (gridDim.y)");
    const int Ho = H / 2;
    const size_t lds = (size_t)(128 + 64) * SB_ROWB + (size_t)SP_PATCH * sizeof(__bf16);
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)stem_pool_b16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
    }();
    (void)attr;
    static const bool form2 = [] { const char* e = getenv("GRL_STEM_POOL2"); return !e || atoi(e) != 0; }();       // (0: A/B and tests)
    if (form2 && Ho % 4 == 0) {
        // strips of whole iterations (4 stem rows); enough workgroups for two per CU, as few warm-up iterations as possible
        int strips = 1;
        while (strips * 2 <= Ho / 8 && (int64_t)n * strips < 512) strips *= 2;
        const int strip_rows = (Ho / 4 + strips - 1) / strips * 4;
        if (x_is_u8)
            hipLaunchKernelGGL(stem_pool2_b16_kernel<true>, dim3(grl_ceil_div(Ho, strip_rows), n), dim3(256), (size_t)S2_LDS,
                               (hipStream_t)stream, reinterpret_cast<const float*>(x), scale, shift, B16(y), H, CB16(wp), mean_std, strip_rows);
        else
            hipLaunchKernelGGL(stem_pool2_b16_kernel<false>, dim3(grl_ceil_div(Ho, strip_rows), n), dim3(256), (size_t)S2_LDS,
                               (hipStream_t)stream, reinterpret_cast<const float*>(x), scale, shift, B16(y), H, CB16(wp),
                               (const float*)nullptr, strip_rows);
        return grl_check_launch("grl_stem_pool_bf16 (form 2)");
    }
    hipLaunchKernelGGL(stem_pool_b16_kernel, dim3(grl_ceil_div(Ho, SP_TPW * SP_TH), n), dim3(256), lds, (hipStream_t)stream,
                       reinterpret_cast<const float*>(x), scale, shift, B16(y), H, W, CB16(wp), x_is_u8 ? mean_std : nullptr);
    return grl_check_launch("grl_stem_pool_bf16");
}

extern "C" int grl_maxpool3x3s2_bf16(const void* x, void* y, int n, int H, int W, int C, void* stream) {
    GRL_REQUIRE(x && y && n > 0 && C % 8 == 0, "maxpool_bf16: bad args");
    const int64_t total = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool_b16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, CB16(x), B16(y),
                       n, H, W, C);
    return grl_check_launch("grl_maxpool3x3s2_bf16");
}

extern "C" int grl_group_mean_bf16(const void* x, float* y, int groups, int rows, int C, int ldy, float out_scale,
                                   int accumulate, void* stream) {
    GRL_REQUIRE(x && y && groups > 0 && rows > 0 && C % 8 == 0, "group_mean_bf16: bad args");
    hipLaunchKernelGGL(group_mean_b16_kernel, dim3(grl_ceil_div(C, 256), groups), dim3(256), 0, (hipStream_t)stream,
                       CB16(x), y, rows, C, ldy, out_scale / rows, accumulate);
    return grl_check_launch("grl_group_mean_bf16");
}

extern "C" int grl_sqdiff_mean_bf16(const void* f1, const void* f2, float* d, int b, int rows, int C,
                                    int64_t f2_clip_stride, void* stream) {
    GRL_REQUIRE(f1 && f2 && d && b > 0 && rows > 0 && C % 8 == 0 && f2_clip_stride % 8 == 0, "sqdiff_mean_bf16: bad args");
    hipLaunchKernelGGL(sqdiff_mean_b16_kernel, dim3(grl_ceil_div(C, 256), b), dim3(256), 0, (hipStream_t)stream,
                       CB16(f1), CB16(f2), d, rows, C, f2_clip_stride);
    return grl_check_launch("grl_sqdiff_mean_bf16");
}

extern "C" int grl_gce_gate_bf16(const void* h, const float* w3, const float* bn_scale, const float* bn_shift,
                                 const void* x, float* corr_map, void* x_corr, void* x_uncorr, int M, int Ch, int C,
                                 void* stream) {
    GRL_REQUIRE(h && w3 && bn_scale && bn_shift && x && x_corr && x_uncorr, "gce_gate_bf16: null");
    GRL_REQUIRE(M > 0 && Ch % 8 == 0 && C % 8 == 0, "gce_gate_bf16: bad shape");
    hipLaunchKernelGGL(gce_gate_b16_kernel, dim3(grl_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, CB16(h), w3,
                       bn_scale, bn_shift, CB16(x), corr_map, B16(x_corr), B16(x_uncorr), M, Ch, C);
    return grl_check_launch("grl_gce_gate_bf16");
}

extern "C" int grl_temporal_mean_bf16(const void* x, void* y, int b, int T, int64_t inner, void* stream) {
    GRL_REQUIRE(x && y && b > 0 && T > 0 && inner % 8 == 0, "temporal_mean_bf16: bad args");
    const int64_t total8 = (int64_t)b * inner / 8;
    hipLaunchKernelGGL(temporal_mean_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, CB16(x),
                       B16(y), T, inner / 8, total8);
    return grl_check_launch("grl_temporal_mean_bf16");
}

extern "C" int grl_add_strided_bf16(const void* a, const void* b, void* y, int nb, int64_t inner,
                                    int64_t b_clip_stride, void* stream) {
    GRL_REQUIRE(a && b && y && nb > 0 && inner % 8 == 0 && b_clip_stride % 8 == 0, "add_strided_bf16: bad args");
    const int64_t total8 = (int64_t)nb * inner / 8;
    hipLaunchKernelGGL(add_strided_b16_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, CB16(a),
                       CB16(b), B16(y), inner / 8, b_clip_stride / 8, total8);
    return grl_check_launch("grl_add_strided_bf16");
}
