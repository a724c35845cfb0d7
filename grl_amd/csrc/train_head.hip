// Backward (and the few train-only forward) kernels of the GCE gate, the TRL
// reductions / channel attention, the L2-normalised tail and the Siamese heads.
// Reference: autograd of reid/models/basebranch.py:58-66, grl_model.py:137-178,222-226,
// Siamese.py:85-140.  Channels-last fp32, float4 per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// map = sigmoid(y[m*ldy]) ; xc = x*map ; xu = x*(1-map)        (one wave per pixel row)
__global__ __launch_bounds__(256) void gate_apply_kernel(const float* __restrict__ y, int ldy,
                                                         const float* __restrict__ x,
                                                         float* __restrict__ cmap,
                                                         float* __restrict__ xc,
                                                         float* __restrict__ xu, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float g = sigmoidf_(y[(int64_t)m * ldy]);
    if (lane == 0) cmap[m] = g;
    const float gu = 1.f - g;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)m * C + c);
        *reinterpret_cast<f32x4*>(xc + (int64_t)m * C + c) = v * g;
        *reinterpret_cast<f32x4*>(xu + (int64_t)m * C + c) = v * gu;
    }
}

// dx (+)= dxc*map + dxu*(1-map);  dy[m*ldy] = map(1-map) * sum_c (dxc-dxu)*x  (other dy cols untouched)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dxc,
                                                       const float* __restrict__ dxu,
                                                       const float* __restrict__ x,
                                                       const float* __restrict__ cmap,
                                                       float* __restrict__ dx, int accumulate,
                                                       float* __restrict__ dy, int ldy, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float g = cmap[m], gu = 1.f - g;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const int64_t o = (int64_t)m * C + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(dxc + o);
        const f32x4 b = *reinterpret_cast<const f32x4*>(dxu + o);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
        s += dot4(a - b, v);
        f32x4 d = a * g + b * gu;
        if (accumulate) d += *reinterpret_cast<const f32x4*>(dx + o);
        *reinterpret_cast<f32x4*>(dx + o) = d;
    }
    s = wave_sum(s);
    if (lane == 0) dy[(int64_t)m * ldy] = s * g * gu;
}

// dst[m][c] (+)= v[m / rpg][c] * scale      (C may be a whole frame: temporal-mean backward)
__global__ void add_rowbcast_kernel(float* __restrict__ dst, const float* __restrict__ v,
                                    int64_t C4, int64_t rpg, float scale, int accumulate,
                                    int64_t total4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / C4, c = i - m * C4;
        f32x4 d = reinterpret_cast<const f32x4*>(v)[(m / rpg) * C4 + c] * scale;
        if (accumulate) d += reinterpret_cast<const f32x4*>(dst)[i];
        reinterpret_cast<f32x4*>(dst)[i] = d;
    }
}

// d = mean_px (f1-f2)^2 backward: df1[b][r][c] = 2 (f1-f2) dd[b][c] / rows ; df2 (+)= -df1
__global__ void sqdiff_bwd_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                  const float* __restrict__ dd, float* __restrict__ df1,
                                  float* __restrict__ df2, int rows, int C4, int64_t f2_stride4,
                                  int acc2, int64_t total4) {
    const float k = 2.f / rows;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % C4, r = (i / C4) % rows, b = i / ((int64_t)C4 * rows);
        const int64_t j = b * f2_stride4 + r * C4 + c;
        const f32x4 g = (reinterpret_cast<const f32x4*>(f1)[i] - reinterpret_cast<const f32x4*>(f2)[j]) *
                        reinterpret_cast<const f32x4*>(dd)[b * C4 + c] * k;
        reinterpret_cast<f32x4*>(df1)[i] = g;
        f32x4 h = -g;
        if (acc2) h += reinterpret_cast<const f32x4*>(df2)[j];
        reinterpret_cast<f32x4*>(df2)[j] = h;
    }
}

// fstep = gap*c + gap backward: ds = dfs*gap*c(1-c) ; dgap (+)= dfs*(1+c)
__global__ void catte_bwd_kernel(const float* __restrict__ dfs, int64_t dfs_stride,
                                 const float* __restrict__ gap, int64_t gap_stride,
                                 const float* __restrict__ catte, float* __restrict__ ds,
                                 float* __restrict__ dgap, int64_t dgap_stride, int acc, int C4,
                                 int64_t total4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / C4, c = (i - b * C4) * 4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dfs + b * dfs_stride + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gap + b * gap_stride + c);
        const f32x4 a = reinterpret_cast<const f32x4*>(catte)[i];
        reinterpret_cast<f32x4*>(ds)[i] = d * g * a * (1.f - a);
        f32x4 o = d * (1.f + a);
        float* gp = dgap + b * dgap_stride + c;
        if (acc) o += *reinterpret_cast<const f32x4*>(gp);
        *reinterpret_cast<f32x4*>(gp) = o;
    }
}

// y = v/|v| backward: dv = (dy - y (y.dy)) / |v|          (one workgroup per row)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                         const float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ v,
                                                         float* __restrict__ dv, int C) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    float dot = 0.f, ss = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(dy + row * lddy + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(y + row * ldy + c);
        const f32x4 w = *reinterpret_cast<const f32x4*>(v + (int64_t)row * C + c);
        dot += dot4(a, b); ss += dot4(w, w);
    }
    dot = block_sum(dot, red);
    ss = block_sum(ss, red);
    const float nrm = sqrtf(ss);
    const float inv = 1.f / (nrm > 1e-12f ? nrm : 1e-12f);
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(dy + row * lddy + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(y + row * ldy + c);
        *reinterpret_cast<f32x4*>(dv + (int64_t)row * C + c) = (a - b * dot) * inv;
    }
}

// ---------------------------------------------------------------------------------
// Siamese temporal attention backward, one workgroup per clip (T <= 16).
//   forward: qh_i = q_i/|q_i|, kh_j = k_j/|k_j|, S = qh kh^T, P = softmax_j S,
//            w_j = sum_i P_ij, raw = sum_j w_j x_j, out = raw/|raw|
constexpr int ATT_TMAX = 16;
__global__ __launch_bounds__(256) void siamese_attn_bwd_kernel(
    const float* __restrict__ qk, const float* __restrict__ x, const float* __restrict__ out,
    int64_t ldo, const float* __restrict__ dout, int64_t lddo, float* __restrict__ dqk,
    float* __restrict__ dx, int dx_acc, int T, int D, int C) {
    __shared__ float inv_norm[2 * ATT_TMAX];
    __shared__ float S[ATT_TMAX][ATT_TMAX];     // P after softmax
    __shared__ float dS[ATT_TMAX][ATT_TMAX];
    __shared__ float colw[ATT_TMAX], dw[ATT_TMAX];
    __shared__ float red[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = qk + (int64_t)b * T * 2 * D;
    for (int r = wave; r < 2 * T; r += 4) {
        const float* p = base + (int64_t)(r % T) * 2 * D + (r / T) * D;
        float s = 0.f;
        for (int k = lane * 4; k < D; k += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + k); s += dot4(v, v); }
        s = wave_sum(s);
        if (lane == 0) inv_norm[r] = 1.f / sqrtf(s);
    }
    __syncthreads();
    for (int ij = wave; ij < T * T; ij += 4) {
        const int i = ij / T, j = ij - i * T;
        const float* q = base + (int64_t)i * 2 * D;
        const float* k = base + (int64_t)j * 2 * D + D;
        float s = 0.f;
        for (int e = lane * 4; e < D; e += 256)
            s += dot4(*reinterpret_cast<const f32x4*>(q + e), *reinterpret_cast<const f32x4*>(k + e));
        s = wave_sum(s);
        if (lane == 0) S[i][j] = s * inv_norm[i] * inv_norm[T + j];
    }
    __syncthreads();
    if (threadIdx.x < T) {
        const int i = threadIdx.x;
        float mx = S[i][0];
        for (int j = 1; j < T; ++j) mx = S[i][j] > mx ? S[i][j] : mx;
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { const float e = expf(S[i][j] - mx); dS[i][j] = e; sum += e; }
        const float inv = 1.f / sum;
        // keep raw scores in S for the normalisation backward? not needed: store P in S
        for (int j = 0; j < T; ++j) S[i][j] = dS[i][j] * inv;
    }
    __syncthreads();
    if (threadIdx.x < T) {
        float s = 0.f;
        for (int i = 0; i < T; ++i) s += S[i][threadIdx.x];
        colw[threadIdx.x] = s;
    }
    __syncthreads();
    // raw norm and (out . dout)
    const float* xb = x + (int64_t)b * T * C;
    float ss = 0.f, od = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < T; ++j) acc += *reinterpret_cast<const f32x4*>(xb + (int64_t)j * C + c) * colw[j];
        ss += dot4(acc, acc);
        od += dot4(*reinterpret_cast<const f32x4*>(out + b * ldo + c), *reinterpret_cast<const f32x4*>(dout + b * lddo + c));
    }
    ss = block_sum(ss, red);
    od = block_sum(od, red);
    const float inv_raw = 1.f / sqrtf(ss);
    // draw = (dout - out*(out.dout)) / |raw| ; dx_j (+)= w_j draw ; dw_j = draw . x_j
    float dwl[ATT_TMAX];
#pragma unroll
    for (int j = 0; j < ATT_TMAX; ++j) dwl[j] = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const f32x4 dr = (*reinterpret_cast<const f32x4*>(dout + b * lddo + c) -
                          *reinterpret_cast<const f32x4*>(out + b * ldo + c) * od) * inv_raw;
#pragma unroll
        for (int j = 0; j < ATT_TMAX; ++j) {
            if (j < T) {
                const int64_t o = ((int64_t)b * T + j) * C + c;
                dwl[j] += dot4(dr, *reinterpret_cast<const f32x4*>(x + o));
                f32x4 g = dr * colw[j];
                if (dx_acc) g += *reinterpret_cast<const f32x4*>(dx + o);
                *reinterpret_cast<f32x4*>(dx + o) = g;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ATT_TMAX; ++j) {
        if (j < T) {                             // T is uniform: every thread joins the barrier
            const float t = block_sum(dwl[j], red);
            if (threadIdx.x == 0) dw[j] = t;
        }
    }
    __syncthreads();
    // softmax backward: dP_ij = dw_j ; dS_ij = P_ij (dP_ij - sum_k P_ik dP_ik)
    if (threadIdx.x < T) {
        const int i = threadIdx.x;
        float s = 0.f;
        for (int k = 0; k < T; ++k) s += S[i][k] * dw[k];
        for (int j = 0; j < T; ++j) dS[i][j] = S[i][j] * (dw[j] - s);
    }
    __syncthreads();
    // dqh_i = sum_j dS_ij kh_j ; dkh_j = sum_i dS_ij qh_i ; then the row normalisation backward
    //   dq = (dqh - qh (qh.dqh)) / |q|.  Each wave owns rows r = wave, wave+4, ...
    for (int r = wave; r < 2 * T; r += 4) {
        const int idx = r % T, isk = r / T;
        const float* self = base + (int64_t)idx * 2 * D + isk * D;
        float dot = 0.f;
        for (int e = lane * 4; e < D; e += 256) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            for (int o = 0; o < T; ++o) {
                const float coef = isk ? dS[o][idx] * inv_norm[o] : dS[idx][o] * inv_norm[T + o];
                const float* other = base + (int64_t)o * 2 * D + (isk ? 0 : D);
                g += *reinterpret_cast<const f32x4*>(other + e) * coef;
            }
            dot += dot4(g, *reinterpret_cast<const f32x4*>(self + e)) * inv_norm[r];
            *reinterpret_cast<f32x4*>(dqk + ((int64_t)b * T + idx) * 2 * D + isk * D + e) = g;   // dqh for now
        }
        dot = wave_sum(dot);                     // qh . dqh
        for (int e = lane * 4; e < D; e += 256) {
            float* dp = dqk + ((int64_t)b * T + idx) * 2 * D + isk * D + e;
            const f32x4 g = *reinterpret_cast<const f32x4*>(dp);
            const f32x4 qh = *reinterpret_cast<const f32x4*>(self + e) * inv_norm[r];
            *reinterpret_cast<f32x4*>(dp) = (g - qh * dot) * inv_norm[r];
        }
    }
}

// diff[i*ng + j][k] = (p[i][k] - g[j][k])^2
__global__ void pair_sqdiff_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                   float* __restrict__ diff, int ng, int K4, int64_t total4) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total4;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t % K4, pair = t / K4;
        const int64_t i = pair / ng, j = pair - i * ng;
        const f32x4 d = reinterpret_cast<const f32x4*>(p)[i * K4 + k] - reinterpret_cast<const f32x4*>(g)[j * K4 + k];
        reinterpret_cast<f32x4*>(diff)[t] = d * d;
    }
}

// dp[i][k] = sum_j 2 (p_i - g_j) ddiff_ij ; dg[j][k] = -sum_i 2 (p_i - g_j) ddiff_ij
__global__ void pair_sqdiff_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                       const float* __restrict__ dd, float* __restrict__ dp,
                                       float* __restrict__ dg, int np, int ng, int K4) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)(np + ng) * K4;
    if (t >= total) return;
    const int64_t k = t % K4, row = t / K4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (row < np) {
        const f32x4 pv = reinterpret_cast<const f32x4*>(p)[row * K4 + k];
        for (int j = 0; j < ng; ++j)
            s += (pv - reinterpret_cast<const f32x4*>(g)[(int64_t)j * K4 + k]) *
                 reinterpret_cast<const f32x4*>(dd)[(row * ng + j) * K4 + k];
        reinterpret_cast<f32x4*>(dp)[row * K4 + k] = s * 2.f;
    } else {
        const int64_t j = row - np;
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[j * K4 + k];
        for (int i = 0; i < np; ++i)
            s += (reinterpret_cast<const f32x4*>(p)[(int64_t)i * K4 + k] - gv) *
                 reinterpret_cast<const f32x4*>(dd)[((int64_t)i * ng + j) * K4 + k];
        reinterpret_cast<f32x4*>(dg)[j * K4 + k] = s * -2.f;
    }
}

// OIM look-up-table update (reid/loss/oim.py:24-26): for every sample, in batch order,
//   lut[y] = m*lut[y] + (1-m)*x ; lut[y] /= |lut[y]|.
// Updates of different labels commute, updates of one label do not: one workgroup per
// label (the workgroup of the label's FIRST sample) replays that label's samples in order.
__global__ __launch_bounds__(256) void oim_update_kernel(float* __restrict__ lut,
                                                         const float* __restrict__ x,
                                                         const int64_t* __restrict__ labels, int n,
                                                         int D, int num_classes, float m) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    const int64_t y = labels[i];
    if (y < 0 || y >= num_classes) return;                // a label outside the table updates nothing
    for (int j = 0; j < i; ++j)
        if (labels[j] == y) return;                       // not the first sample of this label
    float* row = lut + y * (int64_t)D;
    for (int j = i; j < n; ++j) {
        if (labels[j] != y) continue;                     // uniform across the workgroup
        float ss = 0.f;
        for (int c = threadIdx.x * 4; c < D; c += 1024) {
            f32x4 r = *reinterpret_cast<const f32x4*>(row + c) * m +
                      *reinterpret_cast<const f32x4*>(x + (int64_t)j * D + c) * (1.f - m);
            *reinterpret_cast<f32x4*>(row + c) = r;
            ss += dot4(r, r);
        }
        ss = block_sum(ss, red);
        const float inv = 1.f / sqrtf(ss);
        for (int c = threadIdx.x * 4; c < D; c += 1024)
            *reinterpret_cast<f32x4*>(row + c) = *reinterpret_cast<const f32x4*>(row + c) * inv;
        __syncthreads();
    }
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_gate_apply(const float* y, int ldy, const float* x, float* cmap, float* xc, float* xu, int M,
                              int C, void* stream) {
    GRL_REQUIRE(y && x && cmap && xc && xu && M > 0 && C % 4 == 0 && ldy > 0, "gate_apply: bad args");
    hipLaunchKernelGGL(gate_apply_kernel, dim3(grl_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, y, ldy, x,
                       cmap, xc, xu, M, C);
    return grl_check_launch("grl_gate_apply");
}

extern "C" int grl_gate_bwd(const float* dxc, const float* dxu, const float* x, const float* cmap, float* dx,
                            int accumulate, float* dy, int ldy, int M, int C, void* stream) {
    GRL_REQUIRE(dxc && dxu && x && cmap && dx && dy && M > 0 && C % 4 == 0 && ldy > 0, "gate_bwd: bad args");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grl_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, dxc, dxu, x,
                       cmap, dx, accumulate, dy, ldy, M, C);
    return grl_check_launch("grl_gate_bwd");
}

extern "C" int grl_add_rowbcast(float* dst, const float* v, int64_t M, int64_t C, int64_t rows_per_group,
                                float scale, int accumulate, void* stream) {
    GRL_REQUIRE(dst && v && M > 0 && C % 4 == 0 && rows_per_group > 0, "add_rowbcast: bad args");
    const int64_t total4 = M * C / 4;
    hipLaunchKernelGGL(add_rowbcast_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dst, v, C / 4,
                       rows_per_group, scale, accumulate, total4);
    return grl_check_launch("grl_add_rowbcast");
}

extern "C" int grl_sqdiff_bwd(const float* f1, const float* f2, const float* dd, float* df1, float* df2, int b,
                              int rows, int C, int64_t f2_clip_stride, int accumulate_df2, void* stream) {
    GRL_REQUIRE(f1 && f2 && dd && df1 && df2 && b > 0 && rows > 0 && C % 4 == 0 && f2_clip_stride % 4 == 0,
                "sqdiff_bwd: bad args");
    const int64_t total4 = (int64_t)b * rows * C / 4;
    hipLaunchKernelGGL(sqdiff_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, f1, f2, dd, df1,
                       df2, rows, C / 4, f2_clip_stride / 4, accumulate_df2, total4);
    return grl_check_launch("grl_sqdiff_bwd");
}

extern "C" int grl_catte_bwd(const float* dfs, int64_t dfs_stride, const float* gap, int64_t gap_stride,
                             const float* catte, float* ds, float* dgap, int64_t dgap_stride, int accumulate,
                             int b, int C, void* stream) {
    GRL_REQUIRE(dfs && gap && catte && ds && dgap && b > 0 && C % 4 == 0, "catte_bwd: bad args");
    GRL_REQUIRE(dfs_stride % 4 == 0 && gap_stride % 4 == 0 && dgap_stride % 4 == 0, "catte_bwd: strides % 4");
    const int64_t total4 = (int64_t)b * C / 4;
    hipLaunchKernelGGL(catte_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dfs, dfs_stride,
                       gap, gap_stride, catte, ds, dgap, dgap_stride, accumulate, C / 4, total4);
    return grl_check_launch("grl_catte_bwd");
}

extern "C" int grl_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* v, float* dv,
                              int rows, int C, void* stream) {
    GRL_REQUIRE(dy && y && v && dv && rows > 0 && C % 4 == 0 && lddy % 4 == 0 && ldy % 4 == 0, "l2norm_bwd: bad args");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy, v, dv, C);
    return grl_check_launch("grl_l2norm_bwd");
}

extern "C" int grl_siamese_attn_bwd(const float* qk, const float* x, const float* out, int64_t ldo,
                                    const float* dout, int64_t lddo, float* dqk, float* dx, int dx_accumulate, int b,
                                    int T, int D, int C, void* stream) {
    GRL_REQUIRE(qk && x && out && dout && dqk && dx && b > 0, "siamese_attn_bwd: null");
    GRL_REQUIRE(T >= 1 && T <= ATT_TMAX && D % 4 == 0 && C % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0,
                "siamese_attn_bwd: bad shape (T<=16)");
    hipLaunchKernelGGL(siamese_attn_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, qk, x, out, ldo, dout,
                       lddo, dqk, dx, dx_accumulate, T, D, C);
    return grl_check_launch("grl_siamese_attn_bwd");
}

extern "C" int grl_pair_sqdiff(const float* p, const float* g, float* diff, int np, int ng, int K, void* stream) {
    GRL_REQUIRE(p && g && diff && np > 0 && ng > 0 && K % 4 == 0, "pair_sqdiff: bad args");
    const int64_t total4 = (int64_t)np * ng * K / 4;
    hipLaunchKernelGGL(pair_sqdiff_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, p, g, diff, ng,
                       K / 4, total4);
    return grl_check_launch("grl_pair_sqdiff");
}

extern "C" int grl_pair_sqdiff_bwd(const float* p, const float* g, const float* ddiff, float* dp, float* dg, int np,
                                   int ng, int K, void* stream) {
    GRL_REQUIRE(p && g && ddiff && dp && dg && np > 0 && ng > 0 && K % 4 == 0, "pair_sqdiff_bwd: bad args");
    const int64_t total = (int64_t)(np + ng) * K / 4;
    hipLaunchKernelGGL(pair_sqdiff_bwd_kernel, dim3(grl_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, p,
                       g, ddiff, dp, dg, np, ng, K / 4);
    return grl_check_launch("grl_pair_sqdiff_bwd");
}

extern "C" int grl_oim_update(float* lut, const float* x, const int64_t* labels, int n, int D, int num_classes,
                              float momentum, void* stream) {
    GRL_REQUIRE(lut && x && labels && n > 0 && D % 4 == 0 && num_classes > 0, "oim_update: bad args");
    hipLaunchKernelGGL(oim_update_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, lut, x, labels, n, D, num_classes,
                       momentum);
    return grl_check_launch("grl_oim_update");
}
