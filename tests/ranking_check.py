"""Ranking contract against the REFERENCE's own index fixtures (tests/golden/evaluator_q40_g400.npz:
``indices`` / ``idx_dense`` / ``idx_dense_euclid`` = np.argsort of the matrices that
/root/reference/reid/evaluator/attevaluator.py:33-46 produced, as eva_functions.py:139 ranks them).

A GEMM with another fp32 summation order than the reference's BLAS cannot reproduce the order of two
gallery entries whose reference distances differ by less than fp32 rounding noise; everything else
must be identical.  ``compare`` asserts exactly that and returns the numbers a test prints:

* every row of ``ours`` is a permutation of the gallery;
* at every position where the two rankings differ, the entry we put there and the entry the
  reference put there have REFERENCE distances within ``tol`` of each other (a swap inside a
  noise-level run of neighbours), and the position moved by at most ``max_shift`` places;
* no difference changes what the metric sees: per query, the match / drop pattern along the ranking
  (hence the first-match rank, the CMC row and the AP) is identical.
"""
import numpy as np


def compare(ours, ref_idx, ref_dist, q_pids=None, g_pids=None, q_camids=None, g_camids=None,
            tol=3e-5, max_shift=3):
    ours = np.asarray(ours).astype(np.int64)
    ref_idx = np.asarray(ref_idx).astype(np.int64)
    ref_dist = np.asarray(ref_dist)
    nq, ng = ref_idx.shape
    assert ours.shape == (nq, ng)
    assert np.array_equal(np.sort(ours, axis=1), np.broadcast_to(np.arange(ng), (nq, ng))), 'not a permutation'
    diff = ours != ref_idx
    n_diff = int(diff.sum())
    d_ours = np.take_along_axis(ref_dist, ours, 1)
    d_ref = np.take_along_axis(ref_dist, ref_idx, 1)
    gap = np.abs(d_ours - d_ref)[diff]
    worst = float(gap.max()) if n_diff else 0.0
    assert worst <= tol, 'ranking differs from the reference beyond fp32 noise: %g' % worst
    # how far did an entry move?  position of every gallery id in both rankings
    pos_o = np.empty_like(ours); pos_r = np.empty_like(ref_idx)
    rows = np.arange(nq)[:, None]
    pos_o[rows, ours] = np.arange(ng)[None, :]
    pos_r[rows, ref_idx] = np.arange(ng)[None, :]
    shift = int(np.abs(pos_o - pos_r).max())
    assert shift <= max_shift, 'an entry moved %d places' % shift
    metric_changes = 0
    if q_pids is not None:
        q_pids, g_pids = np.asarray(q_pids), np.asarray(g_pids)
        q_camids, g_camids = np.asarray(q_camids), np.asarray(g_camids)
        m_o = g_pids[ours] == q_pids[:, None]
        m_r = g_pids[ref_idx] == q_pids[:, None]
        k_o = ~(m_o & (g_camids[ours] == q_camids[:, None]))
        k_r = ~(m_r & (g_camids[ref_idx] == q_camids[:, None]))
        for qi in range(nq):
            if not np.array_equal(m_o[qi][k_o[qi]], m_r[qi][k_r[qi]]):
                metric_changes += 1
        assert metric_changes == 0, '%d queries whose CMC / AP input differs' % metric_changes
    return {'positions': nq * ng, 'differ': n_diff, 'rows_touched': int(diff.any(1).sum()),
            'worst_ref_gap': worst, 'max_shift': shift, 'metric_changes': metric_changes}


def dense_case(qf, gf, qp, qc, gp, gc):
    """The non-unit-norm (dense test_all.py mode) case of the fixture: make_golden.py:620-623."""
    import torch
    qd = qf.view(20, 2, -1).mean(1)
    gd = torch.cat((qd, gf[40:240]), 0)
    qpd, qcd = np.asarray(qp)[::2], np.asarray(qc)[::2]
    return qd, gd, qpd, qcd, np.concatenate((qpd, np.asarray(gp)[40:240])), np.concatenate((qcd, np.asarray(gc)[40:240]))
