// Bandwidth-bound kernels of the GRL path for gfx950: stem conv, pooling, the GCE
// gate, the TRL reductions / channel attention, BN folding, L2-normalisation and the
// Siamese temporal attention.  All tensors are channels-last fp32; every kernel moves
// 16 bytes per lane per access (float4) with consecutive lanes on consecutive
// addresses.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// ---------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out,
                                        int N, int C, int taps) {
    // out[n][t][c] = w[n][c][t]
    const int64_t total = (int64_t)N * C * taps;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % C;
        const int t = (i / C) % taps;
        const int n = i / ((int64_t)C * taps);
        out[i] = w[((int64_t)n * C + c) * taps + t];
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean,
                               const float* var, const float* bias, float eps, float* scale,
                               float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float mu = mean ? mean[c] : 0.f;
    // same operation order as ATen's eval batch_norm: invstd = 1/sqrt(var+eps)
    const float s = var ? g * (1.f / sqrtf(var[c] + eps)) : g;
    if (scale) scale[c] = s;
    shift[c] = b - mu * s + (bias ? bias[c] * s : 0.f);
}

// ---------------------------------------------------------------------------------
// Stem: 7x7/s2/p3 conv, Cin = 3 (K = 147 is too thin and too ragged for the MFMA
// path), folded BN + ReLU.  One workgroup = 16x16 output pixels x 64 channels; the
// 37x37x3 input patch is staged in LDS, each lane owns one pixel and keeps its 64
// channel accumulators in VGPRs; weights/scale/shift are wave-uniform and come
// through the scalar cache (s_load), so the inner loop is one LDS read per 64 FMAs.
constexpr int ST = 16;                 // output tile edge
constexpr int SP = 2 * ST + 5;         // input patch edge (37)

__global__ __launch_bounds__(256) void stem_conv7x7_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ y, int H, int W, int relu) {
    __shared__ float patch[3][SP][SP + 1];
    const int Ho = H >> 1, Wo = W >> 1;
    const int img = blockIdx.z, oy0 = blockIdx.y * ST, ox0 = blockIdx.x * ST;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const float* xi = x + (int64_t)img * 3 * H * W;
    for (int i = threadIdx.x; i < 3 * SP * SP; i += 256) {
        const int c = i / (SP * SP), r = (i / SP) % SP, q = i % SP;
        const int iy = iy0 + r, ix = ix0 + q;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = xi[((int64_t)c * H + iy) * W + ix];
        patch[c][r][q] = v;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    float acc[64];
#pragma unroll
    for (int o = 0; o < 64; ++o) acc[o] = 0.f;
    for (int c = 0; c < 3; ++c)
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float v = patch[c][2 * ty + ky][2 * tx + kx];
                const float* wk = w + (c * 7 + ky) * 7 + kx;      // w[o][c][ky][kx], o-stride 147
#pragma unroll
                for (int o = 0; o < 64; ++o) acc[o] = fmaf(v, wk[o * 147], acc[o]);
            }
        }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < Ho && ox < Wo) {
        float* yo = y + (((int64_t)img * Ho + oy) * Wo + ox) * 64;
#pragma unroll
        for (int o = 0; o < 64; o += 4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = acc[o + e] * scale[o + e] + shift[o + e];
                v[e] = (t > 0.f || !relu) ? t : 0.f;
            }
            *reinterpret_cast<f32x4*>(yo + o) = v;
        }
    }
}


// ---------------------------------------------------------------------------------
// Stem as an implicit GEMM on the fp32 MFMA: one workgroup = 8 x 16 output pixels (M = 128)
// x 64 channels, K = 3*7*7 = 147 padded to 160.  The input patch (3 x 21 x 37) and the
// whole weight matrix live in LDS; the A operand A[m][k] = patch[c][2py+ky][2px+kx] is
// read with ds_read_b32 through a k -> patch-offset table, the B operand with
// ds_read_b128 (rows padded to 164 floats: conflict-free).  Same k-permutation as the
// GEMM kernel: lane half h supplies k = 8q + 4h + s at step s of chunk q.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SM_TH = 8, SM_TW = 16;                 // output tile
constexpr int SM_PH = 2 * SM_TH + 5, SM_PW = 2 * SM_TW + 5, SM_PWP = SM_PW + 1;   // 21 x 37 (+1)
constexpr int SM_K = 160, SM_WLD = 164;
constexpr int SM_PATCH = 3 * SM_PH * SM_PWP;         // floats; cells SM_PATCH..+3 are zeros
constexpr int SM_KOFF = (SM_PATCH + 4 + 3) / 4 * 4;  // offset of the k -> patch-offset table

__global__ __launch_bounds__(256) void stem_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ y, int H, int W, int relu,
    const float* __restrict__ wp, const float* __restrict__ norm) {
    // norm != NULL: x holds raw u8 pixels, normalised while the patch is staged:
    // (u/255 - mean[c]) / std[c], the same fp32 operations as ToTensor + Normalize.
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ws = sm;                                   // [64][164]
    float* patch = Ws + 64 * SM_WLD;                  // [3][21][38] + zero cell (+pad)
    int* koff = reinterpret_cast<int*>(patch + SM_KOFF);         // [160], 16-byte aligned
    float* Cs = sm;                                   // epilogue staging [128][64] (reuses Ws/patch)
    const int Ho = H >> 1, Wo = W >> 1;
    const int img = blockIdx.z, oy0 = blockIdx.y * SM_TH, ox0 = blockIdx.x * SM_TW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const float* xi = x + (int64_t)img * 3 * H * W;
    const uint8_t* xu = reinterpret_cast<const uint8_t*>(x) + (int64_t)img * 3 * H * W;
    // staging: every global load is issued before the first is waited for (as plain `i += 256` loops hipcc
    // emitted load / s_waitcnt vmcnt(0) / ds_write per iteration -- ~20 exposed latencies per workgroup)
    constexpr int P_N = 3 * SM_PH * SM_PW, P_IT = (P_N + 255) / 256, W_N = 64 * SM_WLD / 4, W_IT = (W_N + 255) / 256;
    float pv[P_IT];
    f32x4 wv[W_IT];
    // (round 5: the u8 / fp32 choice is made OUTSIDE the loop -- with `norm ? xu[o] : xi[o]` inside it every iteration
    //  was a branch again and hipcc, which counts vmcnt per basic block, waited vmcnt(0) after each load)
    auto load_patch = [&](auto u8_) {
        constexpr bool U8 = decltype(u8_)::value;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int i = tid + it * 256;
            const int c = i / (SM_PH * SM_PW), r = (i / SM_PW) % SM_PH, q = i % SM_PW;
            const int iy = iy0 + r, ix = ix0 + q;
            const bool ok = i < P_N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int cc = c < 3 ? c : 0;
            const int64_t o = ok ? ((int64_t)cc * H + iy) * W + ix : 0;
            float v;
            if constexpr (U8) v = ((float)xu[o] / 255.f - norm[cc]) / norm[3 + cc];
            else v = xi[o];
            pv[it] = ok ? v : 0.f;
        }
    };
    if (norm) load_patch(std::true_type{});
    else load_patch(std::false_type{});
    if (wp) {                                          // LDS image made once by grl_stem_pack_weight
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            wv[it] = reinterpret_cast<const f32x4*>(wp)[i < W_N ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            if (i < W_N) reinterpret_cast<f32x4*>(Ws)[i] = wv[it];
        }
    } else {
        for (int i = tid; i < 64 * SM_K; i += 256) {
            const int n = i / SM_K, k = i - n * SM_K;
            Ws[n * SM_WLD + k] = k < 147 ? w[n * 147 + k] : 0.f;
        }
    }
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int i = tid + it * 256;
        if (i < P_N) {
            const int c = i / (SM_PH * SM_PW), r = (i / SM_PW) % SM_PH, q = i % SM_PW;
            patch[(c * SM_PH + r) * SM_PWP + q] = pv[it];
        }
    }
    if (tid < 8) patch[SM_PATCH + (tid & 3)] = 0.f;
    if (tid < SM_K) {
        const int k = tid;
        koff[k] = k < 147 ? ((k / 49) * SM_PH + (k / 7) % 7) * SM_PWP + k % 7 : SM_PATCH;
    }
    __syncthreads();
    // wave tile: 64 pixels (2 MFMA row tiles) x 32 channels
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + frow;
        abase[i] = (2 * (m / SM_TW)) * SM_PWP + 2 * (m % SM_TW);
    }
    const float* brow = Ws + (wn * 32 + frow) * SM_WLD;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 4
    for (int q = 0; q < SM_K / 8; ++q) {
        const int kb = 8 * q + 4 * fhalf;
        const f32x4 bf = *reinterpret_cast<const f32x4*>(brow + kb);
        const int4 ko = *reinterpret_cast<const int4*>(koff + kb);
        const int kov[4] = {ko.x, ko.y, ko.z, ko.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // padded k (>= 147) points at the zero cell for every pixel
            const float a0 = patch[kov[s] == SM_PATCH ? SM_PATCH : abase[0] + kov[s]];
            const float a1 = patch[kov[s] == SM_PATCH ? SM_PATCH : abase[1] + kov[s]];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[s], acc[1], 0, 0, 0);
        }
    }
    __syncthreads();                                   // Ws / patch are dead: reuse as C staging
    const int col_l = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * 64 + wn * 32 + col_l] = acc[i][r];
    __syncthreads();
    // 128 pixels x 64 channels = 2048 float4: 8 per thread, 16 lanes per pixel row
    const int c4 = (tid & 15) * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4), sh = *reinterpret_cast<const f32x4*>(shift + c4);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int m = it * 16 + (tid >> 4);
        const int oy = oy0 + m / SM_TW, ox = ox0 + m % SM_TW;
        if (oy < Ho && ox < Wo) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + m * 64 + c4) * sc + sh;
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(y + (((int64_t)img * Ho + oy) * Wo + ox) * 64 + c4) = v;
        }
    }
}

__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                    int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C >> 2;
    const int64_t total = (int64_t)n * Ho * Wo * C4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = i % C4;
        int64_t r = i / C4;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int img = r / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(
                    x + (((int64_t)img * H + iy) * W + ix) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
    }
}

// ---------------------------------------------------------------------------------
// y[g][c] (+)= out_scale/rows * sum_r x[g][r][c].  One workgroup per (group, 256-channel
// slab): 64 lanes x float4 cover the slab, the 4 waves split the rows.
// 16 waves per workgroup share the rows of one (group, 256-column) strip: the grid of a GAP over
// 32 clips is only 256 workgroups, so the loads in flight per CU come from the waves, not the grid.
constexpr int RED_WAVES = 16;
__device__ __forceinline__ f32x4 red16(const f32x4 (*red)[64], int l) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < RED_WAVES; w += 4)
        t += (red[w][l] + red[w + 1][l]) + (red[w + 2][l] + red[w + 3][l]);
    return t;
}

__global__ __launch_bounds__(1024) void group_mean_kernel(const float* __restrict__ x,
                                                          float* __restrict__ y, int rows, int C,
                                                          int ldy, float mul, int accumulate) {
    __shared__ f32x4 red[RED_WAVES][64];
    const int g = blockIdx.y, c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int wave = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float* xp = x + (int64_t)g * rows * C + c;
#pragma unroll 8
        for (int r = wave; r < rows; r += RED_WAVES) s += *reinterpret_cast<const f32x4*>(xp + (int64_t)r * C);
    }
    red[wave][threadIdx.x & 63] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        f32x4 t = red16(red, threadIdx.x);
        t *= mul;
        float* yp = y + (int64_t)g * ldy + c;
        if (accumulate) t += *reinterpret_cast<const f32x4*>(yp);
        *reinterpret_cast<f32x4*>(yp) = t;
    }
}

// d[b][c] = mean_r (f1[b][r][c] - f2[b*stride + r*C + c])^2
__global__ __launch_bounds__(1024) void sqdiff_mean_kernel(const float* __restrict__ f1,
                                                           const float* __restrict__ f2,
                                                           float* __restrict__ d, int rows, int C,
                                                           int64_t f2_stride) {
    __shared__ f32x4 red[RED_WAVES][64];
    const int g = blockIdx.y, c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int wave = threadIdx.x >> 6;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float* p1 = f1 + (int64_t)g * rows * C + c;
        const float* p2 = f2 + (int64_t)g * f2_stride + c;
#pragma unroll 8
        for (int r = wave; r < rows; r += RED_WAVES) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p1 + (int64_t)r * C) -
                            *reinterpret_cast<const f32x4*>(p2 + (int64_t)r * C);
            s += t * t;
        }
    }
    red[wave][threadIdx.x & 63] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        f32x4 t = red16(red, threadIdx.x);
        t *= 1.f / rows;
        *reinterpret_cast<f32x4*>(d + (int64_t)g * C + c) = t;
    }
}

// ---------------------------------------------------------------------------------
// GCE gate: one wave per pixel row.
__global__ __launch_bounds__(256) void gce_gate_kernel(
    const float* __restrict__ h, const float* __restrict__ w3, const float* __restrict__ bsc,
    const float* __restrict__ bsh, const float* __restrict__ x, float* __restrict__ cmap,
    float* __restrict__ xc, float* __restrict__ xu, int M, int Ch, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float s = 0.f;
    for (int k = lane * 4; k < Ch; k += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(h + (int64_t)m * Ch + k);
        const f32x4 b = *reinterpret_cast<const f32x4*>(w3 + k);
        s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    s = wave_sum(s);
    const float g = sigmoidf_(s * bsc[0] + bsh[0]);
    if (lane == 0 && cmap) cmap[m] = g;
    const float gu = 1.f - g;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)m * C + c);
        *reinterpret_cast<f32x4*>(xc + (int64_t)m * C + c) = v * g;
        *reinterpret_cast<f32x4*>(xu + (int64_t)m * C + c) = v * gu;
    }
}

__global__ void temporal_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                     int64_t inner4, int64_t total4) {
    const float inv = 1.f / T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / inner4, r = i - b * inner4;
        const f32x4* xp = reinterpret_cast<const f32x4*>(x) + b * T * inner4 + r;
        f32x4 s = xp[0];
        for (int t = 1; t < T; ++t) s += xp[t * inner4];
        reinterpret_cast<f32x4*>(y)[i] = s * inv;
    }
}

__global__ void add_strided_kernel(const float* __restrict__ a, const float* __restrict__ bsrc,
                                   float* __restrict__ y, int64_t inner4, int64_t bstride4,
                                   int64_t total4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / inner4, r = i - b * inner4;
        reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(a)[i] +
                                         reinterpret_cast<const f32x4*>(bsrc)[b * bstride4 + r];
    }
}

// ---------------------------------------------------------------------------------
// Channel attention of one TRL step (b clips, C channels, Hd hidden units), two
// launches so that the 2 x 1 MB of MLP weights are streamed by the whole chip instead
// of by one workgroup per clip:
//   hid[b][j] = relu(W1[j] . d[b])           one wave per (clip, hidden unit)
//   c[b][n]   = sigmoid(sum_j W2T[j][n] hid[b][j]);  fstep (+)= (1 + c) * gap
__global__ __launch_bounds__(256) void channel_hidden_kernel(const float* __restrict__ d,
                                                             const float* __restrict__ w1,
                                                             float* __restrict__ hid, int C, int Hd,
                                                             int total) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= total) return;
    const int b = idx / Hd, j = idx - b * Hd;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(w1 + (int64_t)j * C + c);
        const f32x4 v = *reinterpret_cast<const f32x4*>(d + (int64_t)b * C + c);
        s += a[0] * v[0] + a[1] * v[1] + a[2] * v[2] + a[3] * v[3];
    }
    s = wave_sum(s);
    if (lane == 0) hid[idx] = s > 0.f ? s : 0.f;
}

__global__ __launch_bounds__(256) void channel_atte_out_kernel(
    const float* __restrict__ hid, const float* __restrict__ w2t, const float* __restrict__ gap,
    int64_t gap_stride, float* __restrict__ catte, float* __restrict__ fstep, int64_t fstep_stride,
    int accumulate, int C, int Hd) {
    extern __shared__ __attribute__((aligned(16))) float hs[];      // [Hd]
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < Hd; j += 256) hs[j] = hid[(int64_t)b * Hd + j];
    __syncthreads();
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (int j = 0; j < Hd; ++j) s += *reinterpret_cast<const f32x4*>(w2t + (int64_t)j * C + c) * hs[j];   // (latency: 64 workgroups, 128 dependent-free loads each)
    f32x4 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = sigmoidf_(s[e]);
    if (catte) *reinterpret_cast<f32x4*>(catte + (int64_t)b * C + c) = a;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gap + (int64_t)b * gap_stride + c);
    f32x4 o = g * a + g;                                   // reference: mean(x*c + x) == gap*c + gap
    float* fp = fstep + (int64_t)b * fstep_stride + c;
    if (accumulate) o += *reinterpret_cast<const f32x4*>(fp);
    *reinterpret_cast<f32x4*>(fp) = o;
}

// y[row] = v / max(|v|, 1e-12), v = x[row]*scale + shift   (one workgroup per row)
__global__ __launch_bounds__(256) void affine_l2norm_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            float* __restrict__ y, int C,
                                                            int64_t ldy) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    float ss = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)row * C + c);
        if (scale) v = v * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = block_sum(ss, red);
    const float nrm = sqrtf(ss);
    const float inv = 1.f / (nrm > 1e-12f ? nrm : 1e-12f);
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)row * C + c);
        if (scale) v = v * *reinterpret_cast<const f32x4*>(scale + c) + *reinterpret_cast<const f32x4*>(shift + c);
        *reinterpret_cast<f32x4*>(y + (int64_t)row * ldy + c) = v * inv;
    }
}

// ---------------------------------------------------------------------------------
// Siamese temporal attention, one workgroup (4 waves) per clip.  qk rows hold the
// BN-folded Q (first D) and K (next D) projections of the T frames.
//   q_i, k_j L2-normalised; S = q k^T (T x T); P = softmax_j(S);
//   pooled = sum_i sum_j P_ij x_j = sum_j (sum_i P_ij) x_j ; pooled /= |pooled|
constexpr int ATT_TMAX = 16;
__global__ __launch_bounds__(256) void siamese_attn_kernel(const float* __restrict__ qk,
                                                           const float* __restrict__ x,
                                                           float* __restrict__ pooled, int T,
                                                           int D, int C, int64_t ldy) {
    __shared__ float inv_norm[2 * ATT_TMAX];
    __shared__ float S[ATT_TMAX][ATT_TMAX];
    __shared__ float colw[ATT_TMAX];
    __shared__ float red[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qk + (int64_t)b * T * 2 * D;
    // row norms of Q_i (index i) and K_j (index T + j)
    for (int r = wave; r < 2 * T; r += 4) {
        const float* p = base + (int64_t)(r % T) * 2 * D + (r / T) * D;
        float s = 0.f;
        for (int k = lane * 4; k < D; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + k);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        s = wave_sum(s);
        if (lane == 0) inv_norm[r] = 1.f / sqrtf(s);
    }
    __syncthreads();
    for (int ij = wave; ij < T * T; ij += 4) {
        const int i = ij / T, j = ij - i * T;
        const float* q = base + (int64_t)i * 2 * D;
        const float* k = base + (int64_t)j * 2 * D + D;
        float s = 0.f;
        for (int e = lane * 4; e < D; e += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(q + e);
            const f32x4 c = *reinterpret_cast<const f32x4*>(k + e);
            s += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
        s = wave_sum(s);
        if (lane == 0) S[i][j] = s * inv_norm[i] * inv_norm[T + j];
    }
    __syncthreads();
    if (threadIdx.x < T) {                      // softmax of row i, in place
        const int i = threadIdx.x;
        float mx = S[i][0];
        for (int j = 1; j < T; ++j) mx = S[i][j] > mx ? S[i][j] : mx;
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { const float e = expf(S[i][j] - mx); S[i][j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < T; ++j) S[i][j] *= inv;
    }
    __syncthreads();
    if (threadIdx.x < T) {
        float s = 0.f;
        for (int i = 0; i < T; ++i) s += S[i][threadIdx.x];
        colw[threadIdx.x] = s;
    }
    __syncthreads();
    const float* xb = x + (int64_t)b * T * C;
    float ss = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < T; ++j) acc += *reinterpret_cast<const f32x4*>(xb + (int64_t)j * C + c) * colw[j];
        ss += acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
        *reinterpret_cast<f32x4*>(pooled + (int64_t)b * ldy + c) = acc;
    }
    ss = block_sum(ss, red);
    const float inv = 1.f / sqrtf(ss);
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float* p = pooled + (int64_t)b * ldy + c;
        *reinterpret_cast<f32x4*>(p) = *reinterpret_cast<const f32x4*>(p) * inv;   // own element
    }
}

__global__ void mean_T_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int C,
                              int64_t ldy, int64_t total4) {
    const int C4 = C >> 2;
    const float inv = 1.f / T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / C4;
        const int c = (i - b * C4) * 4;
        const float* xp = x + b * T * C + c;
        f32x4 s = *reinterpret_cast<const f32x4*>(xp);
        for (int t = 1; t < T; ++t) s += *reinterpret_cast<const f32x4*>(xp + (int64_t)t * C);
        *reinterpret_cast<f32x4*>(y + b * ldy + c) = s * inv;
    }
}

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x,
                                                         float* __restrict__ out, int K, int ld) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    float ss = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)row * ld + k);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) out[row] = ss;
}

// Pair verification head (Siamese.py:127-140), eval mode:
//   out[i][j][c] = bias[c] + sum_k W[c][k] * (scale[k]*(p[i][k]-g[j][k])^2 + shift[k])
// one wave per (i, j) pair; ncls <= 4.
__global__ __launch_bounds__(256) void pair_verify_kernel(
    const float* __restrict__ p, const float* __restrict__ g, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int np, int ng, int K, int ncls) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= np * ng) return;
    const int i = pair / ng, j = pair - i * ng;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + (int64_t)i * K + k);
        const f32x4 b = *reinterpret_cast<const f32x4*>(g + (int64_t)j * K + k);
        f32x4 d = a - b;
        d = d * d;
        if (scale) d = d * *reinterpret_cast<const f32x4*>(scale + k) + *reinterpret_cast<const f32x4*>(shift + k);
        for (int c = 0; c < ncls; ++c) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)c * K + k);
            acc[c] += d[0] * wv[0] + d[1] * wv[1] + d[2] * wv[2] + d[3] * wv[3];
        }
    }
    for (int c = 0; c < ncls; ++c) {
        const float s = wave_sum(acc[c]);
        if (lane == 0) out[(int64_t)pair * ncls + c] = s + (bias ? bias[c] : 0.f);
    }
}

// Row-wise argsort of the distance matrix (eva_functions.py:139 `np.argsort(distmat, axis=1)`)
// as one LDS bitonic network per row: (key, index) pairs, ascending, ties broken by the
// smaller index (np.argsort(kind='stable') order; numpy's default introsort leaves ties
// unspecified).  One workgroup of 1024 lanes per row, n <= 16384 (128 KiB of LDS).
// Keys are compared as order-preserving unsigned integers so that the comparator is a strict
// total order for every input: -0 == +0, NaN (any sign) sorts after +inf as in numpy, and the
// padding of the network after every real element -- the output is always a permutation of 0..n-1.
constexpr int SORT_MAX = 16384;
__device__ __forceinline__ unsigned sort_key(float v) {
    unsigned u = __float_as_uint(v);
    if (v != v) u = 0x7fc00000u;                       // canonical NaN
    else if (v == 0.f) u = 0u;                         // -0 -> +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(1024) void row_argsort_kernel(const float* __restrict__ d, int64_t ld,
                                                           int n, int P, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm_sort[];
    unsigned* key = sm_sort;                                // [P]
    int* idx = reinterpret_cast<int*>(sm_sort + P);         // [P]
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* dr = d + (int64_t)row * ld;
    for (int i = tid; i < P; i += 1024) {
        key[i] = i < n ? sort_key(dr[i]) : 0xffffffffu;
        idx[i] = i < n ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += 1024) {
                const int i = 2 * j * (t / j) + (t % j), l = i + j;
                const bool asc = (i & k) == 0;
                const unsigned ki = key[i], kl = key[l];
                const int ii = idx[i], il = idx[l];
                const bool gt = ki > kl || (ki == kl && ii > il);          // element i after element l?
                if (gt == asc) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += 1024) out[(int64_t)row * n + i] = idx[i];
}

// ---- rows wider than one LDS network (n > 16384): the same bitonic network, cut at the LDS size.
// Keys / indices of the padded row (P = 2^ceil(log2 n) entries) live in a caller workspace; chunks of
// CH = 16384 entries are sorted in LDS (`k_lo`..`k_hi` stages; the direction bit of stage k comes from
// the GLOBAL position, so neighbouring chunks come out ascending / descending as the network needs),
// compare-exchange steps at distance j >= CH run as plain global passes.  Same total order as the
// one-workgroup kernel: (key, index), ties to the smaller index = numpy's stable argsort.
constexpr int SORT_CH = 16384;

// stages k = k_lo .. k_hi (each with j = min(k/2, CH/2) .. 1) of the network inside every chunk
__global__ __launch_bounds__(1024) void sort_chunk_kernel(const float* __restrict__ d, int64_t ld, int n, int P,
                                                          unsigned* __restrict__ gkey, int* __restrict__ gidx,
                                                          int k_lo, int k_hi, int load_from_d,
                                                          int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm_sort[];
    unsigned* key = sm_sort;
    int* idx = reinterpret_cast<int*>(sm_sort + SORT_CH);
    const int row = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int g0 = chunk * SORT_CH;
    unsigned* rk = gkey + (int64_t)row * P + g0;
    int* ri = gidx + (int64_t)row * P + g0;
    if (load_from_d) {
        const float* dr = d + (int64_t)row * ld;
        for (int i = tid; i < SORT_CH; i += 1024) {
            const int gi = g0 + i;
            key[i] = gi < n ? sort_key(dr[gi]) : 0xffffffffu;
            idx[i] = gi < n ? gi : 0x7fffffff;
        }
    } else {
        for (int i = tid; i < SORT_CH; i += 1024) { key[i] = rk[i]; idx[i] = ri[i]; }
    }
    __syncthreads();
    for (int k = k_lo; k <= k_hi; k <<= 1) {
        for (int j = min(k >> 1, SORT_CH >> 1); j > 0; j >>= 1) {
            for (int t = tid; t < (SORT_CH >> 1); t += 1024) {
                const int i = 2 * j * (t / j) + (t % j), l = i + j;
                const bool asc = ((g0 + i) & k) == 0;
                const unsigned ki = key[i], kl = key[l];
                const int ii = idx[i], il = idx[l];
                const bool gt = ki > kl || (ki == kl && ii > il);
                if (gt == asc) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
            }
            __syncthreads();
        }
    }
    if (out) {                                             // last pass: the row is sorted
        for (int i = tid; i < SORT_CH; i += 1024)
            if (g0 + i < n) out[(int64_t)row * n + g0 + i] = idx[i];
    } else {
        for (int i = tid; i < SORT_CH; i += 1024) { rk[i] = key[i]; ri[i] = idx[i]; }
    }
}

// one compare-exchange step of stage k at distance j >= CH, in global memory
__global__ void sort_global_step_kernel(unsigned* __restrict__ gkey, int* __restrict__ gidx, int P, int k, int j) {
    const int row = blockIdx.y;
    unsigned* rk = gkey + (int64_t)row * P;
    int* ri = gidx + (int64_t)row * P;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (P >> 1); t += gridDim.x * blockDim.x) {
        const int i = 2 * j * (t / j) + (t % j), l = i + j;
        const bool asc = (i & k) == 0;
        const unsigned ki = rk[i], kl = rk[l];
        const int ii = ri[i], il = ri[l];
        const bool gt = ki > kl || (ki == kl && ii > il);
        if (gt == asc) { rk[i] = kl; rk[l] = ki; ri[i] = il; ri[l] = ii; }
    }
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

namespace {
// y = (u/255 - mean[c]) / std[c] over [n][3][plane] (ToTensor + Normalize, seqtransforms.py:190,212-213)
__global__ void normalize_u8_kernel(const uint8_t* __restrict__ x, const float* __restrict__ norm,
                                    float* __restrict__ y, int64_t plane, int64_t total) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((t / plane) % 3);
        y[t] = ((float)x[t] / 255.f - norm[c]) / norm[3 + c];
    }
}

// Training augmentation + ToTensor + Normalize in one pass over raw uint8 clips [n][T][3][H][W]:
//   RandomHorizontalFlip (whole clip), RandomSizedEarser (per frame, a constant-colour patch
//   pasted AFTER the flip, seqtransforms.py:92-151), then ((float)u / 255 - mean) / std.
// params[clip] = {flip, T x {erase, left, top, w, h, R, G, B}} drawn on the host
// (grl_amd/reid/data/augment.py).  One thread per 4 output pixels of a row.
__global__ void augment_normalize_u8_kernel(const uint8_t* __restrict__ x, const int* __restrict__ params,
                                            const float* __restrict__ norm, float* __restrict__ y, int T,
                                            int H, int W, int64_t total4) {
    const int W4 = W >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(i % W4);
        int64_t r = i / W4;
        const int yy = (int)(r % H); r /= H;
        const int c = (int)(r % 3); r /= 3;
        const int t = (int)(r % T);
        const int64_t clip = r / T;
        const int* pc = params + clip * (1 + 8 * T);
        const int flip = pc[0];
        const int* pf = pc + 1 + 8 * t;
        const int erase = pf[0], ex0 = pf[1], ey0 = pf[2], ex1 = pf[1] + pf[3], ey1 = pf[2] + pf[4];
        const uint8_t colour = (uint8_t)pf[5 + c];
        const uint8_t* row = x + (((clip * T + t) * 3 + c) * H + yy) * (int64_t)W;
        const bool in_y = erase && yy >= ey0 && yy < ey1;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = xq * 4 + e;
            uint8_t u = row[flip ? W - 1 - xx : xx];
            if (in_y && xx >= ex0 && xx < ex1) u = colour;
            o[e] = ((float)u / 255.f - norm[c]) / norm[3 + c];
        }
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

// RectScale (seqtransforms.py:30-47) = PIL's `resize(size, BILINEAR)` on 8-bit frames, bit for bit:
// Pillow resamples horizontally, rounds to uint8, then vertically, with 22-bit fixed-point taps
// (libImaging/Resample.c).  The host builds the per-axis tap tables with Pillow's own arithmetic
// (grl_amd/reid/data/augment.py:pil_bilinear_coeffs); one thread per output pixel recomputes the
// (at most kv) horizontally resampled, rounded pixels its vertical taps need.
__global__ void resize_bilinear_u8_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y,
                                          const int* __restrict__ bh, const int* __restrict__ ch, int kh,
                                          const int* __restrict__ bv, const int* __restrict__ cv, int kv,
                                          int Hin, int Win, int Hout, int Wout, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wout);
        const int yo = (int)((i / Wout) % Hout);
        const int64_t plane = i / ((int64_t)Wout * Hout);
        const uint8_t* src = x + plane * (int64_t)Hin * Win;
        const int x0 = bh[2 * xo], nx = bh[2 * xo + 1], y0 = bv[2 * yo], ny = bv[2 * yo + 1];
        int acc_v = 1 << 21;
        for (int r = 0; r < ny; ++r) {
            const uint8_t* row = src + (int64_t)(y0 + r) * Win + x0;
            int acc = 1 << 21;
            for (int t = 0; t < nx; ++t) acc += (int)row[t] * ch[xo * kh + t];
            int h8 = acc >> 22;
            h8 = h8 < 0 ? 0 : (h8 > 255 ? 255 : h8);
            acc_v += h8 * cv[yo * kv + r];
        }
        int o = acc_v >> 22;
        y[i] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
}
}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_resize_bilinear_u8(const uint8_t* x, uint8_t* y, const int* bounds_h, const int* coefs_h, int kh,
                                      const int* bounds_v, const int* coefs_v, int kv, int64_t planes, int Hin,
                                      int Win, int Hout, int Wout, void* stream) {
    GRL_REQUIRE(x && y && bounds_h && coefs_h && bounds_v && coefs_v && planes > 0, "resize_bilinear_u8: null");
    GRL_REQUIRE(kh > 0 && kv > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resize_bilinear_u8: bad shape");
    const int64_t total = planes * Hout * Wout;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, y, bounds_h,
                       coefs_h, kh, bounds_v, coefs_v, kv, Hin, Win, Hout, Wout, total);
    return grl_check_launch("grl_resize_bilinear_u8");
}

extern "C" int grl_augment_normalize_u8(const uint8_t* x, const int* params, const float* mean_std, float* y,
                                        int n_clips, int T, int H, int W, void* stream) {
    GRL_REQUIRE(x && params && mean_std && y && n_clips > 0 && T > 0 && H > 0 && W > 0 && W % 4 == 0,
                "augment_normalize_u8: bad args (W must be a multiple of 4)");
    const int64_t total4 = (int64_t)n_clips * T * 3 * H * (W / 4);
    int64_t blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(augment_normalize_u8_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, params,
                       mean_std, y, T, H, W, total4);
    return grl_check_launch("grl_augment_normalize_u8");
}

extern "C" int grl_normalize_u8(const uint8_t* x, const float* mean_std, float* y, int n, int64_t plane,
                                void* stream) {
    GRL_REQUIRE(x && mean_std && y && n > 0 && plane > 0, "normalize_u8: bad args");
    const int64_t total = (int64_t)n * 3 * plane;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, mean_std, y, plane,
                       total);
    return grl_check_launch("grl_normalize_u8");
}

extern "C" int grl_pack_conv_weight(const float* w, float* out, int N, int C, int kh, int kw, void* stream) {
    GRL_REQUIRE(w && out && N > 0 && C > 0 && kh > 0 && kw > 0, "pack_conv_weight: bad args");
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for((int64_t)N * C * kh * kw)), dim3(256), 0,
                       (hipStream_t)stream, w, out, N, C, kh * kw);
    return grl_check_launch("grl_pack_conv_weight");
}

extern "C" int grl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                           const float* bias, float eps, float* scale, float* shift, int C, void* stream) {
    GRL_REQUIRE(shift && C > 0, "bn_fold: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(grl_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, mean, var, bias, eps, scale, shift, C);
    return grl_check_launch("grl_bn_fold");
}

__global__ void stem_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // [64][SM_WLD]
    if (i >= 64 * SM_WLD) return;
    const int n = i / SM_WLD, k = i - n * SM_WLD;
    wp[i] = k < 147 ? w[n * 147 + k] : 0.f;
}

extern "C" int grl_stem_pack_weight(const float* w, float* wp, void* stream) {
    GRL_REQUIRE(w && wp, "stem_pack_weight: null");
    hipLaunchKernelGGL(stem_pack_weight_kernel, dim3(grl_ceil_div(64 * SM_WLD, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wp);
    return grl_check_launch("grl_stem_pack_weight");
}

static int stem_launch(const float* x, const float* norm, const float* w, const float* scale, const float* shift,
                       float* y, int n, int H, int W, int relu, const float* wp, void* stream);

extern "C" int grl_stem_conv7x7(const float* x, const float* w, const float* scale, const float* shift,
                                float* y, int n, int H, int W, int relu, const float* wp, void* stream) {
    return stem_launch(x, nullptr, w, scale, shift, y, n, H, W, relu, wp, stream);
}

extern "C" int grl_stem_conv7x7_u8(const uint8_t* x, const float* mean_std, const float* w, const float* scale,
                                   const float* shift, float* y, int n, int H, int W, int relu, const float* wp,
                                   void* stream) {
    if (!mean_std) return grl_fail(GRL_EINVAL, "stem_u8: mean_std is null");
    return stem_launch(reinterpret_cast<const float*>(x), mean_std, w, scale, shift, y, n, H, W, relu, wp, stream);
}

static int stem_launch(const float* x, const float* norm, const float* w, const float* scale, const float* shift,
                       float* y, int n, int H, int W, int relu, const float* wp, void* stream) {
    GRL_REQUIRE(x && w && scale && shift && y && n > 0, "stem: null/empty");
    GRL_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H and W must be even");
    const int Ho = H / 2, Wo = W / 2;
    static const int use_valu = getenv("GRL_STEM_VALU") ? 1 : 0;     // kernel tuning only
    if (use_valu && !norm) {
        hipLaunchKernelGGL(stem_conv7x7_kernel, dim3(grl_ceil_div(Wo, ST), grl_ceil_div(Ho, ST), n), dim3(256), 0,
                           (hipStream_t)stream, x, w, scale, shift, y, H, W, relu);
    } else {
        const size_t lds = (size_t)(64 * SM_WLD + SM_KOFF + SM_K) * sizeof(float);
        hipLaunchKernelGGL(stem_mfma_kernel, dim3(grl_ceil_div(Wo, SM_TW), grl_ceil_div(Ho, SM_TH), n), dim3(256), lds,
                           (hipStream_t)stream, x, w, scale, shift, y, H, W, relu, wp, norm);
    }
    return grl_check_launch("grl_stem_conv7x7");
}

// ---------------------------------------------------------------------------------
// Stem + 3x3 / stride-2 max-pool in ONE launch, exact fp32 (round 5; eval: resnets1.py:101-104, basebranch.py:27-36):
// the post-ReLU stem map (268 MB per 128 frames, written and read back by the two-launch form) never reaches HBM.
// Orientation D[channel][pixel] = W . patch^T on v_mfma_f32_32x32x2_f32: a wave owns ONE 32-channel block, its 84
// weight values per lane (k = (c, ky, kx) with kx padded to 8; lane half h takes kx = 4 h + u in k-step u of a
// (c, ky) chunk) stay in registers for the whole launch, and a chunk's four B operands are four consecutive floats of
// the staged input row (8-byte aligned: a stem pixel is two input pixels).  Input rows live in a 16-row ring per channel
// (4 new rows per iteration, written under the MFMAs); one iteration = 2 stem rows (wave: row w >> 1, channel block
// w & 1, both 32-column halves, interleaved accumulators) = 1 pooled row.  Folded BatchNorm + ReLU in registers, the two
// rows go to LDS as [pixel][channel] fp32 (16-byte chunks XOR-swizzled), the pooling threads read 3 x 3 windows
// against the carried previous row.  The k order differs from stem_mfma_kernel's (c, ky, kx) walk: same products,
// another fp32 summation order (tested against torch at 1e-5 like the stem itself).  128 frames: 239 + 75 us -> ~170.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16p __attribute__((ext_vector_type(16)));
constexpr int FP_TW = 64;                              // stem columns = the full width of a 128-pixel-wide frame
constexpr int FP_ROWP = 2 * FP_TW + 6;                 // cells per staged input row: columns -3 .. 130
constexpr int FP_ROWB = FP_ROWP * 4;                   // 536 bytes
constexpr int FP_RING = 16;
constexpr int FP_CHB = FP_RING * FP_ROWB;
constexpr int FP_PATCHB = 3 * FP_CHB;                  // 25728
constexpr int FP_ROWBUF = FP_TW * 256;                 // one stem row [64 px][64 ch] fp32
constexpr int FP_LDS = 3 * FP_ROWBUF + FP_PATCHB;      // 74880: two workgroups per CU
constexpr int FP_WK = 84;                              // weight values per (channel, lane half): 21 chunks x 4 k-steps

__global__ void stem_pack_weight_pool_kernel(const float* __restrict__ w, float* __restrict__ wq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // [64][2][84]: (channel, half h, chunk (c, ky), k-step u) <- kx = 4 h + u
    if (i >= 64 * 2 * FP_WK) return;
    const int ch = i / (2 * FP_WK), r = i - ch * 2 * FP_WK, h = r / FP_WK, t = r - h * FP_WK, chunk = t >> 2, kx = 4 * h + (t & 3);
    wq[i] = kx < 7 ? w[ch * 147 + chunk * 7 + kx] : 0.f;
}

template <bool U8>
__global__ __launch_bounds__(256, 2) void stem_pool_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ y,
    int H, const float* __restrict__ wq, const float* __restrict__ norm, int strip_prows) {
    extern __shared__ __attribute__((aligned(16))) char smf[];
    char* const rowbuf = smf;                                  // [3][64][256 B]
    char* const patch = smf + 3 * FP_ROWBUF;                   // [3][16][536 B]
    constexpr int W = 2 * FP_TW;
    const int Ho = H >> 1, Hp = Ho >> 1, Wp = FP_TW / 2;
    const int img = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pxl = lane & 31, hf = lane >> 5;
    const int jb = wave & 1, wrow = wave >> 1;                  // this wave's channel block and stem row of the iteration
    const float* xi = x + (int64_t)img * 3 * H * W;
    const uint8_t* xu = reinterpret_cast<const uint8_t*>(x) + (int64_t)img * 3 * H * W;
    // input rows by ry = iy + 3 (stem row oy, tap ky: ry = 2 oy + ky); lane l stages the input pixels 2 l, 2 l + 1 of a row
    // (unconditional 8-byte load, a row outside the image is zeroed by a select) at cells 2 l + 3, 2 l + 4
    // (the RAW load is returned; the out-of-image select and the u8 conversion happen in store_row, after the MFMAs: a
    //  select right behind the load -- or any branch between the load and its use -- makes hipcc wait for the load on the spot)
    auto load_row = [&](int c, int ry) {
        const int iy = ry - 3;
        const bool ok = (unsigned)iy < (unsigned)H;
        const int64_t o = ((int64_t)c * H + (ok ? iy : 0)) * W + 2 * lane;
        f32x2 v;
        if constexpr (U8) {
            const unsigned short u = *reinterpret_cast<const unsigned short*>(xu + o);
            v[0] = __builtin_bit_cast(float, (uint32_t)u);
            v[1] = 0.f;
        } else {
            v = *reinterpret_cast<const f32x2*>(xi + o);
        }
        return v;
    };
    auto store_row = [&](int c, int ry, f32x2 v) {
        const bool ok = (unsigned)(ry - 3) < (unsigned)H;
        if constexpr (U8) {
            const uint32_t u = __builtin_bit_cast(uint32_t, v[0]);
            v[0] = ((float)(u & 255u) / 255.f - norm[c]) / norm[3 + c];
            v[1] = ((float)(u >> 8) / 255.f - norm[c]) / norm[3 + c];
        }
        float* const d = reinterpret_cast<float*>(patch + c * FP_CHB + (ry & (FP_RING - 1)) * FP_ROWB) + 2 * lane + 3;
        d[0] = ok ? v[0] : 0.f;
        d[1] = ok ? v[1] : 0.f;
    };
    const int strip0 = blockIdx.x * strip_prows;                // first pooled row of this workgroup
    int py = strip0 > 0 ? strip0 - 1 : 0;                       // (one warm-up iteration above the strip fills the carry row)
    const int py_end = min(Hp, strip0 + strip_prows);
    for (int i = tid; i < FP_PATCHB / 16; i += 256) reinterpret_cast<uint4*>(patch)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 3 * FP_ROWBUF / 16; i += 256) reinterpret_cast<uint4*>(rowbuf)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int rr = wave; rr < 3 * 9; rr += 4) {                 // the 9 input rows of the first iteration
        const int c = rr / 9, r = rr - 9 * c;
        store_row(c, 4 * py + r, load_row(c, 4 * py + r));
    }
    // weights: lane (channel 32 jb + pxl, half hf): 21 chunks x 4 k-steps
    f32x4 wr[FP_WK / 4];
#pragma unroll
    for (int k = 0; k < FP_WK / 4; ++k) {
        wr[k] = *reinterpret_cast<const f32x4*>(wq + ((32 * jb + pxl) * 2 + hf) * FP_WK + 4 * k);
        asm volatile("" : "+v"(wr[k]));                        // (pinned in registers)
    }
    f32x4 sc[4], sh[4];                                         // this lane's channels 32 jb + 8 q + 4 hf + (0..3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sc[q] = *reinterpret_cast<const f32x4*>(scale + 32 * jb + 8 * q + 4 * hf);
        sh[q] = *reinterpret_cast<const f32x4*>(shift + 32 * jb + 8 * q + 4 * hf);
    }
    const int c8 = tid & 7, ppx = tid >> 3;                     // pooling: this thread's pooled column and 8 channels
    int cs = 0;                                                 // row-buffer slot of the carried row (zeros at the top)
    __syncthreads();
    for (; py < py_end; ++py) {
        const int oy = 2 * py + wrow;
        // the 4 new input rows x 3 channels of the next iteration, 3 per wave: requested here, written to the ring behind the
        // MFMAs, no branch in between (past the last iteration they are loaded and stored all the same: nobody reads them)
        f32x2 pv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int rr = wave + 4 * k;                       // (channel rr >> 2, new row rr & 3)
            pv[k] = load_row(rr >> 2, 4 * py + 9 + (rr & 3));
        }
        __builtin_amdgcn_sched_barrier(0);                     // (hipcc otherwise sinks the loads below the MFMAs, next to their use)
        f32x16p acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
        // B operands double-buffered in registers: the reads of chunk + 1 go out in front of the MFMAs of this chunk
        f32x2 bb[2][2][2];                                      // [buffer][column half][float pair]
        auto read_chunk = [&](int chunk, f32x2 (&dst)[2][2]) {
            const int c = chunk / 7, ky = chunk - 7 * c;
            const char* const rowp = patch + c * FP_CHB + ((2 * oy + ky) & (FP_RING - 1)) * FP_ROWB + 16 * hf + 8 * pxl;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                dst[cb][0] = *reinterpret_cast<const f32x2*>(rowp + 256 * cb);
                dst[cb][1] = *reinterpret_cast<const f32x2*>(rowp + 256 * cb + 8);
            }
        };
        read_chunk(0, bb[0]);
#pragma unroll
        for (int chunk = 0; chunk < 21; ++chunk) {
            if (chunk + 1 < 21) read_chunk(chunk + 1, bb[(chunk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[chunk][u], bb[chunk & 1][cb][u >> 1][u & 1], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);                     // (... and hoists the ring stores, with their wait, up between the MFMAs)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int rr = wave + 4 * k;
            store_row(rr >> 2, 4 * py + 9 + (rr & 3), pv[k]);
        }
        __syncthreads();                                       // everyone has pooled the previous iteration and read this one's patch rows
        const int s0 = cs == 2 ? 0 : cs + 1, s1 = s0 == 2 ? 0 : s0 + 1;        // slots of this iteration's rows 0 / 1
        const int myslot = wrow == 0 ? s0 : s1;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int px = 32 * cb + pxl;
            char* const dst = rowbuf + myslot * FP_ROWBUF + px * 256;
            const int sw = (px >> 1) & 15;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tv = acc[cb][4 * q + e] * sc[q][e] + sh[q][e];
                    v[e] = tv > 0.f ? tv : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + (((8 * jb + 2 * q + hf) ^ sw) << 4)) = v;
            }
        }
        __syncthreads();
        // pooled row py: (carry, row 0, row 1) x stem columns 2 ppx - 1 .. 2 ppx + 1; post-ReLU values are >= 0
        f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int slot = rr == 0 ? cs : (rr == 1 ? s0 : s1);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int cx = 2 * ppx + dx;
                if (cx >= 0) {
                    const char* const src = rowbuf + slot * FP_ROWBUF + cx * 256;
                    const int sw = (cx >> 1) & 15;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (((2 * c8) ^ sw) << 4));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (((2 * c8 + 1) ^ sw) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        m0[e] = v0[e] > m0[e] ? v0[e] : m0[e];
                        m1[e] = v1[e] > m1[e] ? v1[e] : m1[e];
                    }
                }
            }
        }
        if (py >= strip0) {
            float* const o = y + (((int64_t)img * Hp + py) * Wp + ppx) * 64 + c8 * 8;
            *reinterpret_cast<f32x4*>(o) = m0;
            *reinterpret_cast<f32x4*>(o + 4) = m1;
        }
        cs = s1;
    }
}

extern "C" int grl_stem_pack_weight_pool(const float* w, float* wq, void* stream) {
    GRL_REQUIRE(w && wq, "stem_pack_weight_pool: null");
    hipLaunchKernelGGL(stem_pack_weight_pool_kernel, dim3(grl_ceil_div(64 * 2 * FP_WK, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wq);
    return grl_check_launch("grl_stem_pack_weight_pool");
}

extern "C" int grl_stem_pool_f32(const void* x, int x_is_u8, const float* mean_std, const float* scale, const float* shift,
                                 float* y, int n, int H, int W, const float* wq, void* stream) {
    GRL_REQUIRE(x && scale && shift && y && wq && n > 0, "stem_pool_f32: null/empty");
    GRL_REQUIRE(W == 2 * FP_TW && H % 4 == 0, "stem_pool_f32: needs W == 128 and H % 4 == 0");
    GRL_REQUIRE(!x_is_u8 || mean_std, "stem_pool_f32: u8 input needs mean_std");
    GRL_REQUIRE(((uintptr_t)x & 7) == 0 || x_is_u8, "stem_pool_f32: x must be 8-byte aligned");
    GRL_REQUIRE(!x_is_u8 || ((uintptr_t)x & 1) == 0, "stem_pool_f32: u8 input must be 2-byte aligned (2-byte row loads)");
    GRL_REQUIRE(n <= 65535, "stem_pool_f32: at most 65535 frames per launch (gridDim.y)");
    const int Hp = H / 4;
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)stem_pool_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FP_LDS);
        (void)hipFuncSetAttribute((const void*)stem_pool_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FP_LDS);
        return true;
    }();
    (void)attr;
    // strips of pooled rows: enough workgroups for two per CU, as few warm-up iterations as possible
    int strips = 1;
    while (strips * 2 <= Hp / 4 && (int64_t)n * strips < 512) strips *= 2;
    const int strip_prows = (Hp + strips - 1) / strips;
    if (x_is_u8)
        hipLaunchKernelGGL(stem_pool_f32_kernel<true>, dim3(grl_ceil_div(Hp, strip_prows), n), dim3(256), (size_t)FP_LDS, (hipStream_t)stream,
                           reinterpret_cast<const float*>(x), scale, shift, y, H, wq, mean_std, strip_prows);
    else
        hipLaunchKernelGGL(stem_pool_f32_kernel<false>, dim3(grl_ceil_div(Hp, strip_prows), n), dim3(256), (size_t)FP_LDS, (hipStream_t)stream,
                           reinterpret_cast<const float*>(x), scale, shift, y, H, wq, (const float*)nullptr, strip_prows);
    return grl_check_launch("grl_stem_pool_f32");
}

extern "C" int grl_maxpool3x3s2(const float* x, float* y, int n, int H, int W, int C, void* stream) {
    GRL_REQUIRE(x && y && n > 0 && C % 4 == 0, "maxpool: bad args");
    const int64_t total = (int64_t)n * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, n, H, W, C);
    return grl_check_launch("grl_maxpool3x3s2");
}

extern "C" int grl_group_mean(const float* x, float* y, int groups, int rows, int C, int ldy, float out_scale,
                              int accumulate, void* stream) {
    GRL_REQUIRE(x && y && groups > 0 && rows > 0 && C % 4 == 0 && ldy % 4 == 0, "group_mean: bad args");
    hipLaunchKernelGGL(group_mean_kernel, dim3(grl_ceil_div(C, 256), groups), dim3(1024), 0, (hipStream_t)stream,
                       x, y, rows, C, ldy, out_scale / rows, accumulate);
    return grl_check_launch("grl_group_mean");
}

extern "C" int grl_gce_gate(const float* h, const float* w3, const float* bn_scale, const float* bn_shift,
                            const float* x, float* corr_map, float* x_corr, float* x_uncorr, int M, int Ch, int C,
                            void* stream) {
    GRL_REQUIRE(h && w3 && bn_scale && bn_shift && x && x_corr && x_uncorr, "gce_gate: null");
    GRL_REQUIRE(M > 0 && Ch % 4 == 0 && C % 4 == 0, "gce_gate: bad shape");
    hipLaunchKernelGGL(gce_gate_kernel, dim3(grl_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, h, w3, bn_scale,
                       bn_shift, x, corr_map, x_corr, x_uncorr, M, Ch, C);
    return grl_check_launch("grl_gce_gate");
}

extern "C" int grl_temporal_mean(const float* x, float* y, int b, int T, int64_t inner, void* stream) {
    GRL_REQUIRE(x && y && b > 0 && T > 0 && inner % 4 == 0, "temporal_mean: bad args");
    const int64_t total4 = (int64_t)b * inner / 4;
    hipLaunchKernelGGL(temporal_mean_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, T,
                       inner / 4, total4);
    return grl_check_launch("grl_temporal_mean");
}

extern "C" int grl_sqdiff_mean(const float* f1, const float* f2, float* d, int b, int rows, int C,
                               int64_t f2_clip_stride, void* stream) {
    GRL_REQUIRE(f1 && f2 && d && b > 0 && rows > 0 && C % 4 == 0 && f2_clip_stride % 4 == 0, "sqdiff_mean: bad args");
    hipLaunchKernelGGL(sqdiff_mean_kernel, dim3(grl_ceil_div(C, 256), b), dim3(1024), 0, (hipStream_t)stream, f1, f2,
                       d, rows, C, f2_clip_stride);
    return grl_check_launch("grl_sqdiff_mean");
}

extern "C" int grl_channel_atte(const float* d, const float* w1, const float* w2t, const float* gap,
                                int64_t gap_stride, float* catte, float* fstep, int64_t fstep_stride, int accumulate,
                                int b, int C, int Hd, float* hid_ws, void* stream) {
    GRL_REQUIRE(d && w1 && w2t && gap && fstep && hid_ws && b > 0 && Hd > 0, "channel_atte: null");
    GRL_REQUIRE(C % 4 == 0 && gap_stride % 4 == 0 && fstep_stride % 4 == 0, "channel_atte: alignment");
    hipLaunchKernelGGL(channel_hidden_kernel, dim3(grl_ceil_div((int64_t)b * Hd, 4)), dim3(256), 0,
                       (hipStream_t)stream, d, w1, hid_ws, C, Hd, b * Hd);
    hipLaunchKernelGGL(channel_atte_out_kernel, dim3(grl_ceil_div(C, 1024), b), dim3(256), (size_t)Hd * sizeof(float),
                       (hipStream_t)stream, hid_ws, w2t, gap, gap_stride, catte, fstep, fstep_stride, accumulate, C, Hd);
    return grl_check_launch("grl_channel_atte");
}

extern "C" int grl_add_strided(const float* a, const float* b, float* y, int nb, int64_t inner, int64_t b_clip_stride,
                               void* stream) {
    GRL_REQUIRE(a && b && y && nb > 0 && inner % 4 == 0 && b_clip_stride % 4 == 0, "add_strided: bad args");
    const int64_t total4 = (int64_t)nb * inner / 4;
    hipLaunchKernelGGL(add_strided_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, a, b, y,
                       inner / 4, b_clip_stride / 4, total4);
    return grl_check_launch("grl_add_strided");
}

extern "C" int grl_affine_l2norm(const float* x, const float* scale, const float* shift, float* y, int rows, int C,
                                 int64_t ldy, void* stream) {
    GRL_REQUIRE(x && y && rows > 0 && C % 4 == 0 && ldy % 4 == 0, "affine_l2norm: bad args");
    GRL_REQUIRE((scale == nullptr) == (shift == nullptr), "affine_l2norm: scale/shift come together");
    hipLaunchKernelGGL(affine_l2norm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, C, ldy);
    return grl_check_launch("grl_affine_l2norm");
}

extern "C" int grl_siamese_attn(const float* qk, const float* x, float* pooled, int b, int T, int D, int C,
                                int64_t ldy, void* stream) {
    GRL_REQUIRE(qk && x && pooled && b > 0, "siamese_attn: null");
    GRL_REQUIRE(T >= 1 && T <= ATT_TMAX && D % 4 == 0 && C % 4 == 0 && ldy % 4 == 0, "siamese_attn: bad shape (T<=16)");
    hipLaunchKernelGGL(siamese_attn_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, qk, x, pooled, T, D, C, ldy);
    return grl_check_launch("grl_siamese_attn");
}

extern "C" int grl_mean_T(const float* x, float* y, int b, int T, int C, int64_t ldy, void* stream) {
    GRL_REQUIRE(x && y && b > 0 && T > 0 && C % 4 == 0 && ldy % 4 == 0, "mean_T: bad args");
    const int64_t total4 = (int64_t)b * C / 4;
    hipLaunchKernelGGL(mean_T_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, T, C, ldy, total4);
    return grl_check_launch("grl_mean_T");
}

extern "C" int grl_row_sqnorm(const float* x, float* out, int rows, int K, int ld, void* stream) {
    GRL_REQUIRE(x && out && rows > 0 && K % 4 == 0 && ld % 4 == 0, "row_sqnorm: bad args");
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, out, K, ld);
    return grl_check_launch("grl_row_sqnorm");
}

extern "C" int grl_pair_verify(const float* p, const float* g, const float* scale, const float* shift,
                               const float* w, const float* bias, float* out, int np, int ng, int K, int ncls,
                               void* stream) {
    GRL_REQUIRE(p && g && w && out && np > 0 && ng > 0, "pair_verify: null/empty");
    GRL_REQUIRE(K % 4 == 0 && ncls >= 1 && ncls <= 4, "pair_verify: K % 4, ncls <= 4");
    GRL_REQUIRE((scale == nullptr) == (shift == nullptr), "pair_verify: scale/shift come together");
    hipLaunchKernelGGL(pair_verify_kernel, dim3(grl_ceil_div((int64_t)np * ng, 4)), dim3(256), 0, (hipStream_t)stream,
                       p, g, scale, shift, w, bias, out, np, ng, K, ncls);
    return grl_check_launch("grl_pair_verify");
}

extern "C" int64_t grl_row_argsort_workspace_bytes(int rows, int n) {
    if (rows <= 0 || n <= SORT_MAX) return 0;
    int64_t P = 2;
    while (P < n) P <<= 1;
    return (int64_t)rows * P * 8;
}

extern "C" int grl_row_argsort_wide(const float* d, int64_t ld, int rows, int n, int32_t* idx, void* workspace,
                                    void* stream) {
    GRL_REQUIRE(d && idx && workspace && rows > 0 && n > SORT_MAX && ld >= n, "row_argsort_wide: bad args");
    GRL_REQUIRE(n <= (1 << 24) && rows <= 65535, "row_argsort_wide: at most 2^24 columns, 65535 rows per call");
    int P = 2;
    while (P < n) P <<= 1;
    hipStream_t s = (hipStream_t)stream;
    unsigned* gkey = reinterpret_cast<unsigned*>(workspace);
    int* gidx = reinterpret_cast<int*>(gkey + (int64_t)rows * P);
    const size_t lds = (size_t)SORT_CH * 8;
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)sort_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SORT_CH * 8);
        return true;
    }();
    (void)attr;
    const dim3 cgrid(P / SORT_CH, rows);
    // stages 2 .. CH entirely inside the chunks
    hipLaunchKernelGGL(sort_chunk_kernel, cgrid, dim3(1024), lds, s, d, ld, n, P, gkey, gidx, 2, SORT_CH, 1,
                       (int*)nullptr);
    for (int k = 2 * SORT_CH; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= SORT_CH; j >>= 1)
            hipLaunchKernelGGL(sort_global_step_kernel, dim3(grid_for(P / 2), rows), dim3(256), 0, s, gkey, gidx, P, k, j);
        // the remaining steps of stage k (j = CH/2 .. 1) stay inside a chunk
        hipLaunchKernelGGL(sort_chunk_kernel, cgrid, dim3(1024), lds, s, d, ld, n, P, gkey, gidx, k, k, 0,
                           k == P ? idx : (int*)nullptr);
    }
    return grl_check_launch("grl_row_argsort_wide");
}

extern "C" int grl_row_argsort(const float* d, int64_t ld, int rows, int n, int32_t* idx, void* stream) {
    GRL_REQUIRE(d && idx && rows > 0 && n > 0 && ld >= n, "row_argsort: bad args");
    GRL_REQUIRE(n <= SORT_MAX, "row_argsort: at most 16384 columns per LDS network (wider rows: grl_row_argsort_wide)");
    int P = 2;
    while (P < n) P <<= 1;
    const size_t lds = (size_t)P * 8;
    if (lds > 65536)
        (void)hipFuncSetAttribute((const void*)row_argsort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(row_argsort_kernel, dim3(rows), dim3(1024), lds, (hipStream_t)stream, d, ld, n, P, idx);
    return grl_check_launch("grl_row_argsort");
}
