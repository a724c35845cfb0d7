// k-reciprocal re-ranking (Zhong et al., CVPR'17) on the device -- SURVEY.md 8(f) rank 3, the
// second half: the reference post-processes the MARS distance matrices with O(N^2) numpy loops
// (reid/evaluator/rerank.py:37-104).  Same algorithm, same fp32 arithmetic, laid out for HBM:
//
//   build    D[i][j] = S[j][i] / max_r S[r][i],  S = [[qq, qg], [qg^T, gg]]^2        (:41-47)
//   (grl_row_argsort ranks every row of D)                                            (:49)
//   krecip   per sample: k-reciprocal set, its 2/3-overlap expansion, Gaussian weights (:55-75)
//   expand   local query expansion: mean of the k2 nearest samples' weight rows       (:77-83)
//   jaccard  per query: sum_k min(V[i][k], V[j][k]) over the query's non-zero k, then
//            (1-lambda) * jaccard + lambda * D                                        (:86-104)
//
// The weight matrix V is a dense N x N fp32 array in HBM (706 MB at MARS size, trivial next to
// 288 GB) plus a short sorted index list per row; the Jaccard pass streams rows of V^T, so a
// lane owns fixed gallery columns, accumulates in registers in ascending-k order (the order of
// the reference's loop) and needs neither atomics nor an inverted index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grl_hip.h"
#include "common.h"

namespace {

constexpr int RR_LMAX = 256;          // k-reciprocal expansion list: (k1+1) + (k1+1)*(k1/2+1) <= 256 for k1 <= 20
constexpr int RR_K1MAX = 20;
constexpr int RR_K2MAX = 8;
constexpr int RR_SLOTS = 64;          // gallery columns per lane in the Jaccard pass: N <= 256 * 64

__device__ __forceinline__ float block_elem(const float* qg, const float* qq, const float* gg, int nq, int ng,
                                            int r, int c) {
    float v;
    if (r < nq) v = c < nq ? qq[(int64_t)r * nq + c] : qg[(int64_t)r * ng + (c - nq)];
    else v = c < nq ? qg[(int64_t)c * ng + (r - nq)] : gg[(int64_t)(r - nq) * ng + (c - nq)];
    return v * v;
}

// colmax[c] = max_r S[r][c]; 64 columns per workgroup, 4 row phases
__global__ __launch_bounds__(256) void rr_colmax_kernel(const float* __restrict__ qg, const float* __restrict__ qq,
                                                        const float* __restrict__ gg, int nq, int ng,
                                                        float* __restrict__ colmax) {
    __shared__ float red[4][64];
    const int N = nq + ng, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float m = -INFINITY;
    if (c < N)
        for (int r = ph; r < N; r += 4) m = fmaxf(m, block_elem(qg, qq, gg, nq, ng, r, c));
    red[ph][threadIdx.x & 63] = m;
    __syncthreads();
    if (ph == 0 && c < N) colmax[c] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]),
                                            fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

// D[i][j] = S[j][i] / colmax[i], 32 x 32 tiles transposed through LDS
__global__ __launch_bounds__(256) void rr_build_kernel(const float* __restrict__ qg, const float* __restrict__ qq,
                                                       const float* __restrict__ gg, int nq, int ng,
                                                       const float* __restrict__ colmax, float* __restrict__ D) {
    __shared__ float tile[32][33];
    const int N = nq + ng;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int rr = ty; rr < 32; rr += 8) {                      // read S[j0+rr][i0+tx] (coalesced in i)
        const int r = j0 + rr, c = i0 + tx;
        tile[rr][tx] = (r < N && c < N) ? block_elem(qg, qq, gg, nq, ng, r, c) : 0.f;
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {                      // write D[i0+rr][j0+tx]
        const int i = i0 + rr, j = j0 + tx;
        if (i < N && j < N) D[(int64_t)i * N + j] = tile[tx][rr] / colmax[i];
    }
}

// numpy's pairwise summation of a contiguous float32 array (np.sum, n <= 256 here)
__device__ float np_pairwise_sum(const float* a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// k-reciprocal neighbours of `s` among its top (k+1): members f of rank[s][0..k] whose own top (k+1)
// contains s.  One wave; returns the ballot mask over the first k+1 lanes (order = rank order).
__device__ __forceinline__ unsigned long long krecip_mask(const int32_t* __restrict__ rank, int64_t ld, int s, int k,
                                                         int lane, int& f_out) {
    bool member = false;
    int f = -1;
    if (lane <= k) {
        f = rank[(int64_t)s * ld + lane];
        const int32_t* rf = rank + (int64_t)f * ld;
        for (int c = 0; c <= k; ++c) member |= (rf[c] == s);
    }
    f_out = f;
    return __ballot(member);
}

// One wave per sample: expansion list (sorted, unique), weights exp(-D) / sum, dense V row + list.
__global__ __launch_bounds__(64) void rr_krecip_kernel(const float* __restrict__ D, const int32_t* __restrict__ rank,
                                                       int N, int k1, int half, float* __restrict__ V,
                                                       int32_t* __restrict__ lcnt, int32_t* __restrict__ lidx) {
    __shared__ int base[RR_K1MAX + 1];
    __shared__ int raw[RR_LMAX], srt[RR_LMAX];
    __shared__ float w[RR_LMAX];
    __shared__ int n_raw;
    const int i = blockIdx.x, lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    int f;
    const unsigned long long mb = krecip_mask(rank, N, i, k1, lane, f);
    const int nb = __popcll(mb);
    if ((mb >> lane) & 1ull) {
        const int p = __popcll(mb & below);
        base[p] = f;
        raw[p] = f;
    }
    if (lane == 0) n_raw = nb;
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int cand = base[b];
        int f2;
        const unsigned long long mc = krecip_mask(rank, N, cand, half, lane, f2);
        const int nc = __popcll(mc);
        bool in_base = false;
        if ((mc >> lane) & 1ull)
            for (int t = 0; t < nb; ++t) in_base |= (base[t] == f2);
        const int inter = __popcll(__ballot(in_base));
        // len(intersect1d(cand_set, base)) > 2./3 * len(cand_set), evaluated in double as numpy does
        if ((double)inter > 2.0 / 3.0 * (double)nc) {
            const int off = n_raw;
            if ((mc >> lane) & 1ull) raw[off + __popcll(mc & below)] = f2;
            __syncthreads();
            if (lane == 0) n_raw = off + nc;
        }
        __syncthreads();
    }
    const int L = n_raw;
    // np.unique: sort by (value, position), drop repeats
    for (int a = lane; a < L; a += 64) {
        const int va = raw[a];
        int p = 0;
        for (int b = 0; b < L; ++b) p += (raw[b] < va || (raw[b] == va && b < a)) ? 1 : 0;
        srt[p] = va;
    }
    __syncthreads();
    int n_u = 0;                                   // ordered compaction, 64 entries per round
    for (int a0 = 0; a0 < L; a0 += 64) {
        const int a = a0 + lane;
        const bool keep = a < L && (a == 0 || srt[a] != srt[a - 1]);
        const unsigned long long mk = __ballot(keep);
        const int v = a < L ? srt[a] : 0;
        __syncthreads();                            // every lane has read srt[a], srt[a-1] of this round
        if (keep) raw[n_u + __popcll(mk & below)] = v;
        n_u += __popcll(mk);
    }
    __syncthreads();
    const float* Di = D + (int64_t)i * N;
    for (int a = lane; a < n_u; a += 64) w[a] = expf(-Di[raw[a]]);
    __syncthreads();
    float total = 0.f;
    if (lane == 0) total = np_pairwise_sum(w, n_u);
    total = __shfl(total, 0);
    float* Vi = V + (int64_t)i * N;
    int32_t* li = lidx + (int64_t)i * RR_LMAX;
    for (int a = lane; a < n_u; a += 64) {
        Vi[raw[a]] = w[a] / total;
        li[a] = raw[a];
    }
    if (lane == 0) lcnt[i] = n_u;
}

// Local query expansion: V2[i] = mean_t V[rank[i][t]] (t < k2; row i itself when k2 == 1), written
// transposed (V2T[e][i]) for the Jaccard pass and, for query rows, as a dense row V2q[i][e].
// Only the union of the k2 index lists can be non-zero.
__global__ __launch_bounds__(256) void rr_expand_kernel(const float* __restrict__ V, const int32_t* __restrict__ rank,
                                                        const int32_t* __restrict__ lcnt,
                                                        const int32_t* __restrict__ lidx, int N, int nq, int k2,
                                                        float* __restrict__ V2T, float* __restrict__ V2q) {
    __shared__ int rows[RR_K2MAX];
    __shared__ int offs[RR_K2MAX + 1];
    const int i = blockIdx.x;
    if (threadIdx.x == 0) {
        int o = 0;
        for (int t = 0; t < k2; ++t) {
            rows[t] = k2 == 1 ? i : rank[(int64_t)i * N + t];
            offs[t] = o;
            o += lcnt[rows[t]];
        }
        offs[k2] = o;
    }
    __syncthreads();
    const int total = offs[k2];
    for (int a = threadIdx.x; a < total; a += 256) {
        int t = 0;
        while (a >= offs[t + 1]) ++t;
        const int e = lidx[(int64_t)rows[t] * RR_LMAX + (a - offs[t])];
        float s = V[(int64_t)rows[0] * N + e];                // np.add.reduce over axis 0: row order
        for (int u = 1; u < k2; ++u) s += V[(int64_t)rows[u] * N + e];
        const float v = k2 == 1 ? s : s / (float)k2;
        V2T[(int64_t)e * N + i] = v;                          // repeats of e write the same value
        if (i < nq) V2q[(int64_t)i * N + e] = v;
    }
}

// One workgroup per query: compact the non-zero k of V2q[i] in ascending order, then every lane
// walks them for its own gallery columns j = tid + 256 s:  acc_j += min(V2q[i][k], V2T[k][j]).
__global__ __launch_bounds__(256) void rr_jaccard_kernel(const float* __restrict__ V2q, const float* __restrict__ V2T,
                                                         const float* __restrict__ D, int N, int nq, float lam,
                                                         float one_minus, float* __restrict__ out) {
    extern __shared__ int sm_i[];                     // [cap] k indices, then [cap] values
    __shared__ int wcnt[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cap = RR_K2MAX * RR_LMAX;
    int* kidx = sm_i;
    float* kval = reinterpret_cast<float*>(sm_i + cap);
    const float* vi = V2q + (int64_t)i * N;
    const unsigned long long below = (1ull << lane) - 1ull;
    int nk = 0;
    for (int base = 0; base < N; base += 256) {
        const int k = base + tid;
        const float v = k < N ? vi[k] : 0.f;
        const bool nz = v != 0.f;
        const unsigned long long m = __ballot(nz);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int p = nk;
        for (int w = 0; w < wave; ++w) p += wcnt[w];
        if (nz) {
            p += __popcll(m & below);
            kidx[p] = k;
            kval[p] = v;
        }
        nk += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    float acc[RR_SLOTS];
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) acc[s] = 0.f;
    for (int a = 0; a < nk; ++a) {
        const float vik = kval[a];
        const float* row = V2T + (int64_t)kidx[a] * N;
#pragma unroll
        for (int s = 0; s < RR_SLOTS; ++s) {
            const int j = tid + 256 * s;
            if (j < N) acc[s] = acc[s] + fminf(vik, row[j]);
        }
    }
    const int ng = N - nq;
    // one_minus = float32(1 - lambda) evaluated in double on the host, as numpy's weak python scalar
    const float* Di = D + (int64_t)i * N;
#pragma unroll
    for (int s = 0; s < RR_SLOTS; ++s) {
        const int j = tid + 256 * s;
        if (j >= nq && j < N) {
            const float jac = 1.f - acc[s] / (2.f - acc[s]);
            out[(int64_t)i * ng + (j - nq)] = jac * one_minus + Di[j] * lam;
        }
    }
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_rerank_build(const float* q_g, const float* q_q, const float* g_g, int nq, int ng, float* D,
                                float* colmax_ws, void* stream) {
    GRL_REQUIRE(q_g && q_q && g_g && D && colmax_ws && nq > 0 && ng > 0, "rerank_build: bad args");
    const int N = nq + ng;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rr_colmax_kernel, dim3(grl_ceil_div(N, 64)), dim3(256), 0, s, q_g, q_q, g_g, nq, ng, colmax_ws);
    hipLaunchKernelGGL(rr_build_kernel, dim3(grl_ceil_div(N, 32), grl_ceil_div(N, 32)), dim3(256), 0, s, q_g, q_q, g_g,
                       nq, ng, colmax_ws, D);
    return grl_check_launch("grl_rerank_build");
}

extern "C" int grl_rerank_krecip(const float* D, const int32_t* rank, int N, int k1, float* V, int32_t* lcnt,
                                 int32_t* lidx, void* stream) {
    GRL_REQUIRE(D && rank && V && lcnt && lidx && N > 0, "rerank_krecip: bad args");
    GRL_REQUIRE(k1 >= 1 && k1 <= RR_K1MAX && k1 < N, "rerank_krecip: 1 <= k1 <= 20 and k1 < N");
    // int(np.around(k1 / 2.)): round half to even
    int half = k1 / 2;
    if (k1 % 2 == 1 && (half % 2 == 1)) half += 1;
    hipLaunchKernelGGL(rr_krecip_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, D, rank, N, k1, half, V, lcnt,
                       lidx);
    return grl_check_launch("grl_rerank_krecip");
}

extern "C" int grl_rerank_expand(const float* V, const int32_t* rank, const int32_t* lcnt, const int32_t* lidx, int N,
                                 int nq, int k2, float* V2T, float* V2q, void* stream) {
    GRL_REQUIRE(V && rank && lcnt && lidx && V2T && V2q && N > 0 && nq > 0 && nq <= N, "rerank_expand: bad args");
    GRL_REQUIRE(k2 >= 1 && k2 <= RR_K2MAX && k2 <= N, "rerank_expand: 1 <= k2 <= 8");
    hipLaunchKernelGGL(rr_expand_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, V, rank, lcnt, lidx, N, nq, k2,
                       V2T, V2q);
    return grl_check_launch("grl_rerank_expand");
}

extern "C" int grl_rerank_jaccard(const float* V2q, const float* V2T, const float* D, int N, int nq,
                                  float lambda_value, float one_minus_lambda, float* out, void* stream) {
    GRL_REQUIRE(V2q && V2T && D && out && N > 0 && nq > 0 && nq < N, "rerank_jaccard: bad args");
    GRL_REQUIRE(N <= 256 * RR_SLOTS, "rerank_jaccard: at most 16384 samples (query + gallery)");
    const size_t lds = (size_t)2 * RR_K2MAX * RR_LMAX * sizeof(int);
    hipLaunchKernelGGL(rr_jaccard_kernel, dim3(nq), dim3(256), lds, (hipStream_t)stream, V2q, V2T, D, N, nq,
                       lambda_value, one_minus_lambda, out);
    return grl_check_launch("grl_rerank_jaccard");
}
