// Ranking metrics on the device: the per-query part of
// eva_functions.evaluate (reid/evaluator/eva_functions.py:134-184) over the row-wise argsort
// that grl_row_argsort leaves in HBM -- drop the gallery entries that share pid AND camera
// with the query, find the rank of the first match (CMC) and the average precision.  The
// 89.6 MB index matrix of the MARS protocol never leaves the GPU; the host receives three
// numbers per query.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grl_hip.h"
#include "common.h"

namespace {

// One workgroup per query.  The row is walked in 256-entry chunks; inside a chunk a lane's
// position among the kept entries / among the hits comes from wave ballots + popcounts, the
// carries across waves and chunks are workgroup-uniform integers.  AP terms are exact
// integer ratios evaluated in fp64 and summed lane-serially then in wave order (fixed).
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int32_t* __restrict__ idx, int64_t ld,
                                                           const int32_t* __restrict__ q_pids,
                                                           const int32_t* __restrict__ q_cams,
                                                           const int32_t* __restrict__ g_pids,
                                                           const int32_t* __restrict__ g_cams, int ng,
                                                           int32_t* __restrict__ first_hit,
                                                           int32_t* __restrict__ n_hits,
                                                           double* __restrict__ ap) {
    __shared__ int wk[4], wh[4];
    __shared__ double wsum[4];
    __shared__ int wfirst[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t qp = q_pids[q], qc = q_cams[q];
    const int32_t* row = idx + (int64_t)q * ld;
    int kept_base = 0, hits_base = 0;                  // carries (uniform)
    int first = 0x7fffffff;
    double sum = 0.0;
    for (int base = 0; base < ng; base += 256) {
        const int t = base + tid;
        bool keep = false, hit = false;
        if (t < ng && (unsigned)row[t] < (unsigned)ng) {          // (a corrupt index is dropped, never read through)
            const int g = row[t];
            const bool same = g_pids[g] == qp;
            keep = !(same && g_cams[g] == qc);
            hit = same && keep;
        }
        const unsigned long long mk = __ballot(keep), mh = __ballot(hit);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { wk[wave] = __popcll(mk); wh[wave] = __popcll(mh); }
        __syncthreads();
        int pk = kept_base, ph = hits_base;
        for (int w = 0; w < wave; ++w) { pk += wk[w]; ph += wh[w]; }
        if (hit) {
            const int pos = pk + __popcll(mk & below);             // 0-based rank among kept entries
            const int nh = ph + __popcll(mh & below) + 1;          // hits up to and including this one
            sum += (double)nh / (double)(pos + 1);
            first = min(first, pos);
        }
        kept_base += wk[0] + wk[1] + wk[2] + wk[3];
        hits_base += wh[0] + wh[1] + wh[2] + wh[3];
        __syncthreads();
    }
    // fixed-order reduction: lanes of a wave serially through shuffles from lane 0 up, then waves
    double ws = 0.0;
    for (int l = 0; l < 64; ++l) ws += __shfl(sum, l);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (lane == 0) { wsum[wave] = ws; wfirst[wave] = first; }
    __syncthreads();
    if (tid == 0) {
        const double s = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
        const int f = min(min(wfirst[0], wfirst[1]), min(wfirst[2], wfirst[3]));
        n_hits[q] = hits_base;
        first_hit[q] = hits_base > 0 ? f : -1;
        ap[q] = hits_base > 0 ? s / (double)hits_base : 0.0;
    }
}

}  // namespace

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_rank_metrics(const int32_t* idx, int64_t ld, const int32_t* q_pids, const int32_t* q_cams,
                                const int32_t* g_pids, const int32_t* g_cams, int nq, int ng, int32_t* first_hit,
                                int32_t* n_hits, double* ap, void* stream) {
    GRL_REQUIRE(idx && q_pids && q_cams && g_pids && g_cams && first_hit && n_hits && ap, "rank_metrics: null");
    GRL_REQUIRE(nq > 0 && ng > 0 && ld >= ng, "rank_metrics: bad shape");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, idx, ld, q_pids, q_cams,
                       g_pids, g_cams, ng, first_hit, n_hits, ap);
    return grl_check_launch("grl_rank_metrics");
}
