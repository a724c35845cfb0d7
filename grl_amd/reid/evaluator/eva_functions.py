"""Ranking metrics on the host (numpy), as in the reference: the distance
matrix comes back from the GPU and the argsort/CMC/AP stay on CPU
(/root/reference/reid/evaluator/eva_functions.py:118-184)."""
import numpy as np
import torch

__all__ = ['accuracy', 'cmc', 'mean_ap', 'evaluate']


def accuracy(output, target, topk=(1,)):
    """top-k precision (eva_functions.py:118-131)."""
    if not torch.is_tensor(output):
        output = torch.from_numpy(np.asarray(output))
    if not torch.is_tensor(target):
        target = torch.from_numpy(np.asarray(target))
    maxk = max(topk)
    n = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand(maxk, n))
    return [correct[:k].reshape(-1).float().sum(0).mul_(1. / n) for k in topk]


def evaluate(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=100, indices=None):
    """CMC curve and mAP (eva_functions.py:134-184).  Rows are ranked with
    ``np.argsort`` exactly as the reference does; gallery entries that share
    both pid and camid with the query are dropped; queries whose identity never
    appears are skipped.  Returns (cmc[max_rank] float32, mAP).  ``indices`` may carry a
    precomputed row-wise argsort; when it is the DEVICE tensor of grl_amd.engine.rank_rows the
    per-query work also runs on the GPU (engine.rank_metrics) and ``distmat`` is not touched."""
    if torch.is_tensor(indices) and indices.is_cuda:
        from grl_amd import engine
        return engine.rank_metrics(indices, q_pids, g_pids, q_camids, g_camids, max_rank)
    distmat = np.asarray(distmat)
    q_pids, g_pids = np.asarray(q_pids), np.asarray(g_pids)
    q_camids, g_camids = np.asarray(q_camids), np.asarray(g_camids)
    num_q, num_g = distmat.shape
    if num_g < max_rank:
        max_rank = num_g
        print("Note: number of gallery samples is quite small, got {}".format(num_g))
    if indices is None:
        indices = np.argsort(distmat, axis=1)
    ranked_pid = g_pids[indices]
    matches = (ranked_pid == q_pids[:, None])
    drop = matches & (g_camids[indices] == q_camids[:, None])
    all_cmc, all_ap = [], []
    for qi in range(num_q):
        hits = matches[qi][~drop[qi]].astype(np.int32)
        if not hits.any():
            continue
        first = hits.cumsum()
        first[first > 1] = 1
        first = first[:max_rank]
        if first.size < max_rank:       # fewer than max_rank gallery entries left for this query: the
            first = np.concatenate([first, np.full(max_rank - first.size, first[-1])])  # curve stays flat
        all_cmc.append(first)           # (the reference builds a ragged array and raises here)
        prec = hits.cumsum() / (np.arange(hits.size) + 1.0)
        all_ap.append((prec * hits).sum() / hits.sum())
    assert len(all_cmc) > 0, "Error: all query identities do not appear in gallery"
    cmc_curve = np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_cmc))
    return cmc_curve, np.mean(all_ap)


def cmc(distmat, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None,
        topk=100, separate_camera_set=False, single_gallery_shot=False, first_match_break=False):
    """CMC curve of eva_functions.py:18-79 with the reference's defaults: per valid query either
    the first match only (``first_match_break=True``) or, by default, every match i (0-based, at
    filtered rank k_i) adding 1/#matches at rank k_i - i.  The gallery-resampling protocols
    (``separate_camera_set`` / ``single_gallery_shot``) are outside the GRL path and raise."""
    if separate_camera_set or single_gallery_shot:
        raise NotImplementedError('separate_camera_set / single_gallery_shot CMC protocols are not provided')
    d = distmat.cpu().numpy() if torch.is_tensor(distmat) else np.asarray(distmat)
    m, n = d.shape
    query_ids = np.arange(m) if query_ids is None else np.asarray(query_ids)
    gallery_ids = np.arange(n) if gallery_ids is None else np.asarray(gallery_ids)
    query_cams = np.zeros(m, np.int32) if query_cams is None else np.asarray(query_cams)
    gallery_cams = np.ones(n, np.int32) if gallery_cams is None else np.asarray(gallery_cams)
    indices = np.argsort(d, axis=1)
    hist, valid_q = np.zeros(topk), 0
    for i in range(m):
        order = indices[i]
        keep = (gallery_ids[order] != query_ids[i]) | (gallery_cams[order] != query_cams[i])
        ranks = np.flatnonzero(gallery_ids[order][keep] == query_ids[i])   # filtered ranks of the matches
        if ranks.size == 0:
            continue
        valid_q += 1
        if first_match_break:
            if ranks[0] < topk:
                hist[ranks[0]] += 1
            continue
        slot = ranks - np.arange(ranks.size)          # non-decreasing: stop at the first slot >= topk
        slot = slot[:np.searchsorted(slot, topk)]
        np.add.at(hist, slot, 1.0 / ranks.size)
    if valid_q == 0:
        raise RuntimeError("No valid query")
    return hist.cumsum() / valid_q


def mean_ap(distmat, query_ids=None, gallery_ids=None, query_cams=None, gallery_cams=None):
    """mAP via sklearn's average_precision_score (eva_functions.py:82-115)."""
    from sklearn.metrics import average_precision_score
    d = distmat.cpu().numpy() if torch.is_tensor(distmat) else np.asarray(distmat)
    m, n = d.shape
    query_ids = np.arange(m) if query_ids is None else np.asarray(query_ids)
    gallery_ids = np.arange(n) if gallery_ids is None else np.asarray(gallery_ids)
    query_cams = np.zeros(m, np.int32) if query_cams is None else np.asarray(query_cams)
    gallery_cams = np.ones(n, np.int32) if gallery_cams is None else np.asarray(gallery_cams)
    indices = np.argsort(d, axis=1)
    matches = gallery_ids[indices] == query_ids[:, None]
    aps = []
    for i in range(m):
        valid = (gallery_ids[indices[i]] != query_ids[i]) | (gallery_cams[indices[i]] != query_cams[i])
        y_true = matches[i, valid]
        if not np.any(y_true):
            continue
        aps.append(average_precision_score(y_true, -d[i][indices[i]][valid]))
    if not aps:
        raise RuntimeError("No valid query")
    return np.mean(aps)
