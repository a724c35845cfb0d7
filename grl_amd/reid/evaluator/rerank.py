"""k-reciprocal re-ranking (Zhong et al., CVPR'17) with the reference's call signature
and numerics (/root/reference/reid/evaluator/rerank.py:37-104): host-side numpy post-
process of the q-g distance matrix, fed by the Euclidean q-q / g-g matrices that
grl_amd.engine.pairwise_distance_tensor computes on the GPU.

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


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
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
