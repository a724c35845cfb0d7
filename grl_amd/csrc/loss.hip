// The trainer's loss block on the device: forward AND backward of
//   * OIMLoss' cross entropy over scalar * x . LUT^T          (reid/loss/oim.py:14-27,46-53)
//   * the batch-hard soft-margin triplet loss                  (reid/loss/triplet.py:16-90)
//   * the pair-verification softmax + BCE + top-1 precision    (reid/loss/pairloss.py:18-45,
//                                                               reid/train/trainer.py:146-149)
// The batches are tiny (<= a few hundred rows), so every kernel is latency bound: one
// workgroup per row, fixed-order reductions (deterministic), no atomics, and upstream
// gradients arrive as DEVICE scalars so that nothing forces a host sync.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// block-wide max for <= 16 waves; all threads receive it.
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

__device__ __forceinline__ bool label_ok(int64_t y, int c) { return y >= 0 && y < c; }

// One workgroup per row: log-softmax, weighted row loss, gradient of the MEAN loss, arg-max hit.
// Every workgroup re-derives sum_i w[y_i] in the same order, so the normaliser is identical in
// all of them without a second pass.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ z, int64_t ld,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ weight, int n, int c,
                                                         float* __restrict__ rows,     // [2][n]: loss, hit
                                                         float* __restrict__ dz, int64_t ldd) {
    __shared__ float red[16];
    __shared__ int arg_s[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* zi = z + (int64_t)i * ld;
    float wsum = 0.f;                                  // serial, fixed order: n is small
    for (int r = 0; r < n; ++r) {
        const int64_t y = labels[r];
        wsum += label_ok(y, c) ? (weight ? weight[y] : 1.f) : 0.f;
    }
    const int64_t y = labels[i];
    const bool ok = label_ok(y, c);
    const float wy = ok ? (weight ? weight[y] : 1.f) : 0.f;

    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int j = tid; j < c; j += 256) {
        const float v = zi[j];
        if (v > m) { m = v; am = j; }
    }
    const float mx = block_max(m, red);
    // first index attaining the maximum (torch.topk / max tie rule)
    int cand = (m == mx) ? am : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
    if ((tid & 63) == 0) arg_s[tid >> 6] = cand;
    float s = 0.f;
    for (int j = tid; j < c; j += 256) s += expf(zi[j] - mx);
    s = block_sum(s, red);                             // (its barriers also publish arg_s)
    const int arg = min(min(arg_s[0], arg_s[1]), min(arg_s[2], arg_s[3]));
    const float lse = logf(s);                         // of the shifted row
    const float gscale = wsum > 0.f ? wy / wsum : 0.f;
    if (dz) {
        float* di = dz + (int64_t)i * ldd;
        for (int j = tid; j < c; j += 256) {
            const float p = expf(zi[j] - mx - lse);
            di[j] = gscale * (p - ((ok && j == (int)y) ? 1.f : 0.f));
        }
    }
    if (tid == 0) {
        rows[i] = ok ? gscale * (lse - (zi[y] - mx)) : 0.f;
        rows[n + i] = (ok && arg == (int)y) ? 1.f : 0.f;
    }
}

// out[k] = sum_i rows[k][i] in index order (one thread per output; n is small)
__global__ void rows_sum_kernel(const float* __restrict__ rows, int n, int nout, float* __restrict__ o0,
                                float* __restrict__ o1) {
    const int k = threadIdx.x;
    if (k >= nout) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += rows[(int64_t)k * n + i];
    float* o = k == 0 ? o0 : o1;
    if (o) o[0] = s;
}

// dx[i][k] = alpha*g * sum_j dl[i][j] * lut[j][k]: 8 rows x 256 columns per workgroup, the 8
// gradient rows in LDS, the LUT streamed once per row block (coalesced along k).
constexpr int OG_ROWS = 8;
__global__ __launch_bounds__(256) void oim_grad_kernel(const float* __restrict__ dl, int64_t ldd,
                                                       const float* __restrict__ lut,
                                                       const float* __restrict__ g, float alpha,
                                                       float* __restrict__ dx, int n, int c, int D) {
    extern __shared__ __attribute__((aligned(16))) float sl[];     // [OG_ROWS][cp], cp = c rounded up to 4, zero padded
    const int cp = (c + 3) & ~3;
    const int r0 = blockIdx.y * OG_ROWS, k = blockIdx.x * 256 + threadIdx.x;
    for (int t = threadIdx.x; t < OG_ROWS * cp; t += 256) {
        const int r = t / cp, j = t - r * cp;
        sl[t] = (r0 + r < n && j < c) ? dl[(int64_t)(r0 + r) * ldd + j] : 0.f;
    }
    __syncthreads();
    if (k >= D) return;
    float acc[OG_ROWS];
#pragma unroll
    for (int r = 0; r < OG_ROWS; ++r) acc[r] = 0.f;
    // four classes per step: one 16-byte LDS read per gradient row instead of four scalar ones (the kernel was bound by
    // LDS instruction issue: 8 broadcast reads per class and lane), j ascending within a row as before
#pragma unroll 8
    for (int j = 0; j < cp; j += 4) {                  // 32 LUT loads in flight per lane: the loop is L2-latency bound
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = j + e < c ? lut[(int64_t)(j + e) * D + k] : 0.f;
#pragma unroll
        for (int r = 0; r < OG_ROWS; ++r) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sl + r * cp + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r] = fmaf(d4[e], w[e], acc[r]);
        }
    }
    const float a = alpha * (g ? g[0] : 1.f);
#pragma unroll
    for (int r = 0; r < OG_ROWS; ++r)
        if (r0 + r < n) dx[(int64_t)(r0 + r) * D + k] = a * acc[r];
}

// ---- triplet ---------------------------------------------------------------------------
// One workgroup per anchor i: the n distances of row i (one wave per j, lanes over the
// features), then the hardest positive / negative with torch's value formulas and first-index
// tie rule:  pos value = dist * [same id, j != i],  neg value = dist + 1e5 * [same id].
__global__ __launch_bounds__(256) void triplet_fwd_kernel(const float* __restrict__ f,
                                                          const int64_t* __restrict__ ids, int n, int D,
                                                          int soft, float margin,
                                                          float* __restrict__ loss, float* __restrict__ dist,
                                                          int32_t* __restrict__ sel, float* __restrict__ zout) {
    extern __shared__ float drow[];                    // [n]
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* fi = reinterpret_cast<const f32x4*>(f + (int64_t)i * D);
    const int D4 = D >> 2;
    for (int j = wave; j < n; j += 4) {
        const f32x4* fj = reinterpret_cast<const f32x4*>(f + (int64_t)j * D);
        float s = 0.f;
        for (int k = lane; k < D4; k += 64) {
            const f32x4 d = fi[k] - fj[k];
            s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float d = sqrtf(s + 1e-12f);
            drow[j] = d;
            dist[(int64_t)i * n + j] = d;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t yi = ids[i];
        float vp = -INFINITY, vn = INFINITY;
        int jp = 0, jn = 0;
        for (int j = 0; j < n; ++j) {
            const bool same = ids[j] == yi;
            const float p = drow[j] * ((same && j != i) ? 1.f : 0.f);
            const float q = drow[j] + 1e5f * (same ? 1.f : 0.f);
            if (p > vp) { vp = p; jp = j; }
            if (q < vn) { vn = q; jn = j; }
        }
        const float z = vp - vn;
        zout[i] = z;
        loss[i] = soft ? logf(1.f + expf(z)) : fmaxf(z + margin, 0.f);
        // a masked maximum (no positive in the batch) carries no gradient
        sel[2 * i] = (ids[jp] == yi && jp != i) ? jp : -1;
        sel[2 * i + 1] = jn;
    }
}

// dfeat[r] = sum over anchors i (index order) of the three ways row r enters z_i = d(i,p_i) - d(i,n_i)
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const float* __restrict__ f,
                                                          const float* __restrict__ dist,
                                                          const int32_t* __restrict__ sel,
                                                          const float* __restrict__ z,
                                                          const float* __restrict__ dloss, int soft,
                                                          float margin, float* __restrict__ df, int n,
                                                          int D4) {
    const int r = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= D4) return;
    const f32x4* F = reinterpret_cast<const f32x4*>(f);
    const f32x4 fr = F[(int64_t)r * D4 + k];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) {
        const int p = sel[2 * i], q = sel[2 * i + 1];
        if (i != r && p != r && q != r) continue;      // workgroup-uniform
        const float ez = expf(z[i]);
        const float gz = dloss[i] * (soft ? ez / (1.f + ez) : (z[i] + margin > 0.f ? 1.f : 0.f));
        const f32x4 fi = F[(int64_t)i * D4 + k];
        if (i == r) {
            if (p >= 0) acc += (fr - F[(int64_t)p * D4 + k]) * (gz / dist[(int64_t)i * n + p]);
            acc -= (fr - F[(int64_t)q * D4 + k]) * (gz / dist[(int64_t)i * n + q]);
        }
        if (p == r && i != r) acc -= (fi - fr) * (gz / dist[(int64_t)i * n + r]);
        if (q == r && i != r) acc += (fi - fr) * (gz / dist[(int64_t)i * n + r]);
    }
    reinterpret_cast<f32x4*>(df)[(int64_t)r * D4 + k] = acc;
}

// ---- pair verification -------------------------------------------------------------------
__global__ void softmax2_kernel(const float* __restrict__ s, float* __restrict__ p, float* __restrict__ p0,
                                int64_t m) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= m) return;
    const float a = s[2 * t], b = s[2 * t + 1], mx = fmaxf(a, b);
    const float ea = expf(a - mx), eb = expf(b - mx);
    p[t] = eb / (ea + eb);
    if (p0) p0[t] = ea / (ea + eb);
}

// torch's softmax backward, term for term: d s_c = (g_c - sum_k g_k p_k) * p_c with g = (0, dp):
// the class-0 term keeps the accurately computed p0 (1 - p1 cancels when p1 -> 1).
__global__ void softmax2_bwd_kernel(const float* __restrict__ p, const float* __restrict__ p0,
                                    const float* __restrict__ dp, float* __restrict__ ds, int64_t m) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= m) return;
    const float g = dp[t], gp = g * p[t];
    ds[2 * t] = (0.f - gp) * p0[t];
    ds[2 * t + 1] = (g - gp) * p[t];
}

// single workgroup: BCE(mean) of prob[a][b] against [tar_probe[b] == tar_gallery[a]], the
// top-1 precision of (1-s, s) (ties -> class 0) and d loss / d prob.  torch's BCELoss clamps
// the logs at -100 and the backward denominator at 1e-12.
__global__ __launch_bounds__(256) void pair_bce_kernel(const float* __restrict__ prob,
                                                       const int64_t* __restrict__ tp,
                                                       const int64_t* __restrict__ tg, int n,
                                                       float* __restrict__ loss, float* __restrict__ prec,
                                                       float* __restrict__ dprob) {
    __shared__ float red[16];
    const int N = n * n;
    const float inv = 1.f / (float)N;
    float l = 0.f, hit = 0.f;
    for (int t = threadIdx.x; t < N; t += 256) {
        const int a = t / n, b = t - a * n;
        const float y = tp[b] == tg[a] ? 1.f : 0.f;
        const float s = prob[t];
        l -= y * fmaxf(logf(s), -100.f) + (1.f - y) * fmaxf(logf(1.f - s), -100.f);
        hit += ((s > 1.f - s) ? 1.f : 0.f) == y ? 1.f : 0.f;
        if (dprob) dprob[t] = (s - y) / fmaxf((1.f - s) * s, 1e-12f) * inv;
    }
    l = block_sum(l, red);
    hit = block_sum(hit, red);
    if (threadIdx.x == 0) {
        loss[0] = l * inv;
        if (prec) prec[0] = hit * inv;
    }
}

__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ g, float alpha,
                                 float* __restrict__ y, int64_t n) {
    const float a = alpha * g[0];
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        y[t] = a * x[t];
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_softmax_ce(const float* logits, int64_t ld, const int64_t* labels, const float* weight, int n,
                              int c, float* loss, float* correct, float* dlogits, int64_t ldd, float* ws,
                              void* stream) {
    GRL_REQUIRE(logits && labels && loss && ws && n > 0 && c > 0 && ld >= c, "softmax_ce: bad args");
    GRL_REQUIRE(!dlogits || ldd >= c, "softmax_ce: ldd < c");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(n), dim3(256), 0, s, logits, ld, labels, weight, n, c, ws, dlogits,
                       ldd);
    hipLaunchKernelGGL(rows_sum_kernel, dim3(1), dim3(64), 0, s, ws, n, 2, loss, correct);
    return grl_check_launch("grl_softmax_ce");
}

extern "C" int grl_oim_grad(const float* dlogits, int64_t ldd, const float* lut, const float* g, float alpha,
                            float* dx, int n, int c, int D, void* stream) {
    GRL_REQUIRE(dlogits && lut && dx && n > 0 && c > 0 && D > 0 && ldd >= c, "oim_grad: bad args");
    const size_t lds = (size_t)OG_ROWS * ((c + 3) & ~3) * sizeof(float);
    GRL_REQUIRE(lds <= 160 * 1024, "oim_grad: more than 5120 classes");
    if (lds > 65536)
        (void)hipFuncSetAttribute((const void*)oim_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(oim_grad_kernel, dim3(grl_ceil_div(D, 256), grl_ceil_div(n, OG_ROWS)), dim3(256), lds,
                       (hipStream_t)stream, dlogits, ldd, lut, g, alpha, dx, n, c, D);
    return grl_check_launch("grl_oim_grad");
}

extern "C" int grl_triplet_fwd(const float* feat, const int64_t* ids, int n, int D, int soft, float margin,
                               float* loss, float* dist, int32_t* sel, float* z, void* stream) {
    GRL_REQUIRE(feat && ids && loss && dist && sel && z && n > 0 && n <= 16384 && D > 0 && D % 4 == 0,
                "triplet_fwd: bad args (D % 4, n <= 16384)");
    hipLaunchKernelGGL(triplet_fwd_kernel, dim3(n), dim3(256), (size_t)n * sizeof(float), (hipStream_t)stream, feat,
                       ids, n, D, soft, margin, loss, dist, sel, z);
    return grl_check_launch("grl_triplet_fwd");
}

extern "C" int grl_triplet_bwd(const float* feat, const float* dist, const int32_t* sel, const float* z,
                               const float* dloss, int soft, float margin, float* dfeat, int n, int D,
                               void* stream) {
    GRL_REQUIRE(feat && dist && sel && z && dloss && dfeat && n > 0 && D > 0 && D % 4 == 0, "triplet_bwd: bad args");
    hipLaunchKernelGGL(triplet_bwd_kernel, dim3(grl_ceil_div(D / 4, 256), n), dim3(256), 0, (hipStream_t)stream, feat,
                       dist, sel, z, dloss, soft, margin, dfeat, n, D / 4);
    return grl_check_launch("grl_triplet_bwd");
}

extern "C" int grl_softmax2(const float* scores, float* prob, float* prob0, int64_t m, void* stream) {
    GRL_REQUIRE(scores && prob && m > 0, "softmax2: bad args");
    hipLaunchKernelGGL(softmax2_kernel, dim3(grl_ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream, scores, prob,
                       prob0, m);
    return grl_check_launch("grl_softmax2");
}

extern "C" int grl_softmax2_bwd(const float* prob, const float* prob0, const float* dprob, float* dscores, int64_t m,
                                void* stream) {
    GRL_REQUIRE(prob && prob0 && dprob && dscores && m > 0, "softmax2_bwd: bad args");
    hipLaunchKernelGGL(softmax2_bwd_kernel, dim3(grl_ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream, prob, prob0,
                       dprob, dscores, m);
    return grl_check_launch("grl_softmax2_bwd");
}

extern "C" int grl_pair_bce(const float* prob, const int64_t* tar_probe, const int64_t* tar_gallery, int n,
                            float* loss, float* prec, float* dprob, void* stream) {
    GRL_REQUIRE(prob && tar_probe && tar_gallery && loss && n > 0 && n <= 4096, "pair_bce: bad args");
    hipLaunchKernelGGL(pair_bce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, prob, tar_probe, tar_gallery, n,
                       loss, prec, dprob);
    return grl_check_launch("grl_pair_bce");
}

extern "C" int grl_scale_dev(const float* x, const float* g, float alpha, float* y, int64_t n, void* stream) {
    GRL_REQUIRE(x && g && y && n > 0, "scale_dev: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_dev_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, g, alpha, y, n);
    return grl_check_launch("grl_scale_dev");
}
