"""k-reciprocal re-ranking (Zhong et al., CVPR'17) with the reference's call signature
and numerics (/root/reference/reid/evaluator/rerank.py:37-104).

Two entries behind one function:
  * DEVICE tensors in -> the whole post-process runs on MI355X (grl_amd/csrc/rerank.hip:
    build / row argsort / k-reciprocal sets / query expansion / Jaccard) and a device tensor
    comes back; nothing of the (q+g)^2 problem touches the host;
  * numpy arrays in -> the host numpy restatement below, as the reference runs it.
Both are fed by the matrices grl_amd.engine computes on the GPU.

Written from the algorithm, vectorised where the reference loops in Python:
  1. stack the four blocks into one (q+g)^2 matrix, square it, column-normalise by the
     column maximum and transpose;
  2. k-reciprocal neighbour set R(i, k1) of every sample, expanded by the k1/2-reciprocal
     sets of its members that overlap them by more than 2/3;
  3. Gaussian-kernel membership vector V_i over the expanded set, averaged over the k2
     nearest neighbours (local query expansion);
  4. Jaccard distance from min/max overlaps of the sparse V rows;
  5. final = (1 - lambda) * jaccard + lambda * original, query rows x gallery columns.
"""
import numpy as np

__all__ = ['re_ranking']


def _k_reciprocal(initial_rank, i, k):
    fwd = initial_rank[i, :k + 1]
    back = initial_rank[fwd, :k + 1]
    return fwd[np.where(back == i)[0]]


def _re_ranking_device(q_g, q_q, g_g, k1, k2, lambda_value):
    """The five launches of grl_amd/csrc/rerank.hip; every buffer is a torch device tensor."""
    import ctypes as C
    import torch
    from grl_amd import _lib, engine
    from grl_amd._lib import ptr
    for t in (q_g, q_q, g_g):
        _lib.require_device(t, 'distance matrix')
    q_g, q_q, g_g = q_g.contiguous(), q_q.contiguous(), g_g.contiguous()
    nq, ng = q_g.shape
    if tuple(q_q.shape) != (nq, nq) or tuple(g_g.shape) != (ng, ng):
        raise ValueError('re_ranking: q_q must be [%d,%d] and g_g [%d,%d]' % (nq, nq, ng, ng))
    N, dev = nq + ng, q_g.device
    if N > 16384:
        raise _lib.GrlHipError('device re_ranking holds one row per LDS sort network: q + g <= 16384 (got %d)' % N)

    def call(name, *args):
        _lib.check(getattr(_lib.load(), name)(*args, _lib.stream()), name)
    D = torch.empty((N, N), dtype=torch.float32, device=dev)
    colmax = torch.empty(N, dtype=torch.float32, device=dev)
    call('grl_rerank_build', ptr(q_g), ptr(q_q), ptr(g_g), nq, ng, ptr(D), ptr(colmax))
    rank = engine.rank_rows(D)
    V = torch.zeros((N, N), dtype=torch.float32, device=dev)
    lcnt = torch.empty(N, dtype=torch.int32, device=dev)
    lidx = torch.empty((N, 256), dtype=torch.int32, device=dev)
    call('grl_rerank_krecip', ptr(D), ptr(rank), N, int(k1), ptr(V), ptr(lcnt), ptr(lidx))
    V2T = torch.zeros((N, N), dtype=torch.float32, device=dev)
    V2q = torch.zeros((nq, N), dtype=torch.float32, device=dev)
    call('grl_rerank_expand', ptr(V), ptr(rank), ptr(lcnt), ptr(lidx), N, nq, int(k2), ptr(V2T), ptr(V2q))
    del V, rank
    out = torch.empty((nq, ng), dtype=torch.float32, device=dev)
    call('grl_rerank_jaccard', ptr(V2q), ptr(V2T), ptr(D), N, nq, C.c_float(lambda_value),
         C.c_float(1 - lambda_value), ptr(out))
    return out


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    try:
        import torch
        if all(torch.is_tensor(t) and t.is_cuda for t in (q_g_dist, q_q_dist, g_g_dist)):
            return _re_ranking_device(q_g_dist, q_q_dist, g_g_dist, k1, k2, lambda_value)
    except ImportError:                                   # pragma: no cover
        pass
    q_g_dist, q_q_dist, g_g_dist = (np.asarray(a, dtype=np.float32) for a in (q_g_dist, q_q_dist, g_g_dist))
    query_num = q_g_dist.shape[0]
    all_num = query_num + q_g_dist.shape[1]
    original = np.concatenate([np.concatenate([q_q_dist, q_g_dist], axis=1),
                               np.concatenate([q_g_dist.T, g_g_dist], axis=1)], axis=0)
    original = np.power(original, 2).astype(np.float32)
    original = np.transpose(1. * original / np.max(original, axis=0))
    initial_rank = np.argsort(original).astype(np.int32)
    V = np.zeros_like(original, dtype=np.float32)
    half = int(np.around(k1 / 2.))
    for i in range(all_num):
        base = _k_reciprocal(initial_rank, i, k1)
        expanded = base
        for cand in base:
            cand_set = _k_reciprocal(initial_rank, cand, half)
            if len(np.intersect1d(cand_set, base)) > 2. / 3 * len(cand_set):
                expanded = np.append(expanded, cand_set)
        expanded = np.unique(expanded)
        weight = np.exp(-original[i, expanded])
        V[i, expanded] = 1. * weight / np.sum(weight)
    original = original[:query_num]
    if k2 != 1:
        V = np.stack([V[initial_rank[i, :k2]].mean(axis=0) for i in range(all_num)]).astype(np.float32)
    inv_index = [np.where(V[:, j] != 0)[0] for j in range(all_num)]
    jaccard = np.zeros_like(original, dtype=np.float32)
    for i in range(query_num):
        temp_min = np.zeros((1, all_num), dtype=np.float32)
        nz = np.where(V[i] != 0)[0]
        for j in nz:
            rows = inv_index[j]
            temp_min[0, rows] = temp_min[0, rows] + np.minimum(V[i, j], V[rows, j])
        jaccard[i] = 1 - temp_min / (2. - temp_min)
    final = jaccard * (1 - lambda_value) + original * lambda_value
    return final[:query_num, query_num:]
