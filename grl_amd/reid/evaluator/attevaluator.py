"""ATTEvaluator with the reference's constructor and methods
(/root/reference/reid/evaluator/attevaluator.py:49-163).  Feature extraction and
the query x gallery distance matrix run on the GPU through grl_amd.engine; the
ranking metrics stay on the host."""
import math

import numpy as np
import torch

from grl_amd import engine
from grl_amd import dist as grl_dist
from .eva_functions import evaluate
from .rerank import re_ranking

__all__ = ['ATTEvaluator', 'evaluate_seq', 'cosin_dist', 'pairwise_distance_tensor']


def evaluate_seq(distmat, query_pids, query_camids, gallery_pids, gallery_camids, path,
                 cmc_topk=(1, 5, 10, 20), indices=None):
    """Prints mAP / Rank-k in the reference's format and returns Rank-1
    (attevaluator.py:15-30)."""
    cmc_scores, mAP = evaluate(distmat, np.array(query_pids), np.array(gallery_pids),
                               np.array(query_camids), np.array(gallery_camids), indices=indices)
    print('Mean AP: {:4.1%}'.format(mAP))
    for r in cmc_topk:
        print("Rank-{:<3}: {:.1%}".format(r, cmc_scores[r - 1]))
    print("------------------")
    return cmc_scores[0]


def cosin_dist(qf, gf):
    return engine.cosin_dist(qf, gf)


def pairwise_distance_tensor(query_x, gallery_x):
    return engine.pairwise_distance_tensor(query_x, gallery_x)


class ATTEvaluator(object):
    def __init__(self, cnn_model, Siamese_model, only_eval):
        self.cnn_model = cnn_model
        self.siamese_model = Siamese_model
        self.only_eval = only_eval
        # clips per forward in dense mode.  The reference uses 8 (attevaluator.py:72-76) to fit
        # its GPU; clip features are independent (eval BN is folded), so any chunking gives the
        # same rows -- 32 keeps the MI355X busy (1900 vs 1400 clip-features/s at 8).
        self.chunk = 32
        # dense mode: tracklets are collected until this many clips are pending, then extracted together in
        # `chunk`-sized forwards -- a MARS tracklet has ~15 clips, one forward per tracklet would leave every
        # forward half empty.  Rows do not depend on what else is in the batch (tested), so the per-tracklet means
        # are the ones a tracklet-by-tracklet run gives, bit for bit.
        self.group = 128

    def _device(self):
        return next(self.cnn_model.parameters()).device

    @torch.no_grad()
    def extract_feature(self, data_loader):
        self.cnn_model.eval()
        self.siamese_model.eval()
        dev = self._device()
        rank, world = grl_dist._rank_world(None, None)
        # data parallel (one process per GPU): batch i is extracted by rank i % world -- clips are
        # independent, there is no collective on the data path -- and the rows are all-gathered once
        # (sharded at the batch SAMPLER: a rank's loader workers decode only its own clips)
        own = [i for i in range(len(data_loader)) if i % world == rank]
        batches = grl_dist.shard_loader_batches(data_loader, rank, world)
        mine = []
        pending, n_pending = [], 0          # dense mode: (batch index, clips [n,s,c,h,w], pids, camids)

        def flush():
            clips = torch.cat([p[1] for p in pending], 0) if len(pending) > 1 else pending[0][1]
            rows = torch.cat([engine.extract_features(self.cnn_model, self.siamese_model, clips[y:y + self.chunk])
                              for y in range(0, clips.size(0), self.chunk)], 0)
            off = 0
            for i, c, pids, camids in pending:
                mine.append((i, engine.rows_mean(rows[off:off + c.size(0)]), pids, camids))
                off += c.size(0)
            del pending[:]
        # the next batch's host->device copy overlaps this batch's kernels (side HIP stream)
        for i, (imgs, pids, camids) in zip(own, engine.DevicePrefetcher(batches, dev)):
            pids, camids = [int(x) for x in pids], [int(x) for x in camids]
            if self.only_eval:
                # dense mode: one tracklet per item, all its clips; features are averaged
                # over clips (attevaluator.py:68-98)
                b, n, s, c, h, w = imgs.size()
                pending.append((i, imgs.view(b * n, s, c, h, w), pids, camids))
                n_pending += b * n
                if n_pending >= self.group:
                    flush()
                    n_pending = 0
            else:
                mine.append((i, engine.extract_features(self.cnn_model, self.siamese_model, imgs), pids, camids))
        if pending:
            flush()
        feat, pids_all, cams_all = grl_dist.gather_feature_batches(mine, len(data_loader))
        return feat, np.asarray(pids_all), np.asarray(cams_all)

    def evaluate(self, query, gallery, query_loader, gallery_loader, path, visual, rerank):
        if visual:
            raise NotImplementedError('ranked-result visualisation is outside the GRL hot path')
        qf, q_pids, q_camids = self.extract_feature(query_loader)
        print('Done, obtained {}-by-{} matrix'.format(qf.size(0), qf.size(1)))
        gf, g_pids, g_camids = self.extract_feature(gallery_loader)
        gf = torch.cat((qf, gf), 0)               # query is prepended (attevaluator.py:143-145)
        g_pids = np.append(q_pids, g_pids)
        g_camids = np.append(q_camids, g_camids)
        print('Done, obtained {}-by-{} matrix'.format(gf.size(0), gf.size(1)))
        print("Computing distance matrix")
        dist_dev = grl_dist.sharded_distmat(qf, gf, cosin_dist)     # gallery rows sharded over the ranks
        # ranking AND the per-query CMC / AP work on the device (one LDS sort network per row up to
        # 16384 gallery entries -- MARS: 11310 -- the chunked network beyond): neither the distance nor
        # the index matrix leaves HBM
        if not rerank:
            return evaluate_seq(None, q_pids, q_camids, g_pids, g_camids, path,
                                indices=engine.rank_rows(dist_dev))
        if rerank and qf.size(0) + gf.size(0) <= 16384:
            print('Applying person re-ranking ...')            # entirely on the device
            dist_dev = re_ranking(dist_dev, pairwise_distance_tensor(qf, qf), pairwise_distance_tensor(gf, gf))
            return evaluate_seq(None, q_pids, q_camids, g_pids, g_camids, path,
                                indices=engine.rank_rows(dist_dev))
        distmat = dist_dev.cpu().numpy()
        if rerank:                                             # beyond one LDS sort network: host numpy
            print('Applying person re-ranking ...')
            distmat_qq = pairwise_distance_tensor(qf, qf).cpu().numpy()
            distmat_gg = pairwise_distance_tensor(gf, gf).cpu().numpy()
            distmat = re_ranking(distmat, distmat_qq, distmat_gg)
        return evaluate_seq(distmat, q_pids, q_camids, g_pids, g_camids, path)
