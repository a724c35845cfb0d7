"""k-reciprocal re-ranking (reference: reid/evaluator/rerank.py:37-104) is an
opt-in host-side numpy post-process outside the hot-path scope of this round
(SURVEY.md section 8(f), rank 3).  The Euclidean q-q / g-g matrices it consumes are
provided on the GPU by grl_amd.engine.pairwise_distance_tensor."""

__all__ = ['re_ranking']


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    raise NotImplementedError(
        're_ranking is not part of the MI355X hot path yet; run the evaluator with rerank=False')
