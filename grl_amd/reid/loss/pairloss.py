"""Pair verification loss (reference: reid/loss/pairloss.py:8-45) on MI355X: BCE between the
match probability of every (probe_i, gallery_j) pair and [pid_i == pid_j], plus the top-1
precision of the (1-s, s) pseudo-logits -- one `grl_pair_bce` launch that also leaves
d loss / d score for the backward.  The label mask is built on the device (the reference
round-trips it through a Python list, a host sync per step).  `pair_prob` is the two-class
softmax the trainer applies to the verification scores first (trainer.py:146-148)."""
import torch
from torch import nn, autograd

from grl_amd import _lib
from grl_amd._lib import ptr, require_device
from grl_amd.reid.loss.oim import _call, _labels


class _PairBCE(autograd.Function):
    @staticmethod
    def forward(ctx, score, tar_probe, tar_gallery):
        require_device(score, 'PairLoss score')
        s = score.contiguous()
        n = s.size(0)
        if s.dim() != 2 or s.size(1) != n:
            raise _lib.GrlHipError('PairLoss expects a square [n, n] score matrix')
        tp, tg = _labels(tar_probe, s.device), _labels(tar_gallery, s.device)
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        prec = torch.empty((), dtype=torch.float32, device=s.device)
        dprob = torch.empty_like(s)
        _call('grl_pair_bce', ptr(s), ptr(tp), ptr(tg), n, ptr(loss), ptr(prec), ptr(dprob))
        ctx.save_for_backward(dprob)
        ctx.mark_non_differentiable(prec)
        return loss, prec

    @staticmethod
    def backward(ctx, gloss, _gprec):
        dprob, = ctx.saved_tensors
        g = gloss.contiguous().float()
        ds = torch.empty_like(dprob)
        import ctypes as C
        _call('grl_scale_dev', ptr(dprob), ptr(g), C.c_float(1.0), ptr(ds), dprob.numel())
        return ds, None, None


class _PairProb(autograd.Function):
    @staticmethod
    def forward(ctx, scores):
        require_device(scores, 'verification scores')
        s = scores.contiguous()
        if s.size(-1) != 2:
            raise _lib.GrlHipError('pair_prob expects [..., 2] scores')
        prob = torch.empty(s.shape[:-1], dtype=torch.float32, device=s.device)
        prob0 = torch.empty_like(prob)
        _call('grl_softmax2', ptr(s), ptr(prob), ptr(prob0), prob.numel())
        ctx.save_for_backward(prob, prob0)
        return prob

    @staticmethod
    def backward(ctx, dprob):
        prob, prob0 = ctx.saved_tensors
        ds = torch.empty(prob.shape + (2,), dtype=torch.float32, device=prob.device)
        _call('grl_softmax2_bwd', ptr(prob), ptr(prob0), ptr(dprob.contiguous()), ptr(ds), prob.numel())
        return ds


def pair_prob(scores):
    """softmax(scores, -1)[..., 1] (trainer.py:146-148)."""
    return _PairProb.apply(scores)


class PairLoss(nn.Module):
    def __init__(self):
        super(PairLoss, self).__init__()

    def forward(self, score, tar_probe, tar_gallery):
        return _PairBCE.apply(score, tar_probe, tar_gallery)
