"""Batch-hard triplet loss (reference: reid/loss/triplet.py:16-90, the `mode='id'`,
`dis_func='eu'`, `batch_hard=True` path the trainer uses with margin 'soft':
trainer.py:13,141) on MI355X: `grl_triplet_fwd` (distance rows, hardest positive / negative,
per-anchor loss) and `grl_triplet_bwd` (gather-form gradient, fixed order)."""
import ctypes as C

import torch
import torch.nn as nn
from torch import autograd

from grl_amd._lib import ptr, require_device
from grl_amd.reid.loss.oim import _call, _labels


class _BatchHard(autograd.Function):
    @staticmethod
    def forward(ctx, feat, ids, soft, margin):
        require_device(feat, 'TripletLoss feat')
        f = feat.contiguous()
        y = _labels(ids, f.device)
        n, D = f.shape
        loss = torch.empty(n, dtype=torch.float32, device=f.device)
        dist = torch.empty((n, n), dtype=torch.float32, device=f.device)
        z = torch.empty(n, dtype=torch.float32, device=f.device)
        sel = torch.empty((n, 2), dtype=torch.int32, device=f.device)
        _call('grl_triplet_fwd', ptr(f), ptr(y), n, D, soft, C.c_float(margin), ptr(loss), ptr(dist), ptr(sel),
              ptr(z))
        ctx.save_for_backward(f, dist, sel, z)
        ctx.soft, ctx.margin = soft, margin
        return loss

    @staticmethod
    def backward(ctx, dloss):
        f, dist, sel, z = ctx.saved_tensors
        n, D = f.shape
        df = torch.empty_like(f)
        _call('grl_triplet_bwd', ptr(f), ptr(dist), ptr(sel), ptr(z), ptr(dloss.contiguous().float()), ctx.soft,
              C.c_float(ctx.margin), ptr(df), n, D)
        return df, None, None, None


class TripletLoss(nn.Module):
    def __init__(self, margin=0, batch_hard=False, dim=2048):
        super(TripletLoss, self).__init__()
        self.batch_hard = batch_hard
        if isinstance(margin, float) or margin == 'soft':
            self.margin = margin
        else:
            raise NotImplementedError('The margin {} is not recognized in TripletLoss()'.format(margin))

    def forward(self, feat, id=None, pos_mask=None, neg_mask=None, mode='id', dis_func='eu', n_dis=0):
        if mode != 'id' or n_dis != 0 or not self.batch_hard or dis_func != 'eu':
            raise NotImplementedError('only the batch-hard id-mode Euclidean path of the reference trainer is provided')
        if id is None:
            raise RuntimeError('foward is in id mode, please input id!')
        soft = self.margin == 'soft'
        return _BatchHard.apply(feat, id, 1 if soft else 0, 0.0 if soft else float(self.margin))


class TripletLoss_OIM(TripletLoss):
    """Instantiated by the reference trainer (trainer.py:11) but never called."""
