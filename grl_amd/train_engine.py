"""Train-mode (batch-statistics BatchNorm + backward) side of the GRL path and
the pair-verification heads.

Reference call sites: reid/models/Siamese.py:108-142,
reid/models/Siamese_video.py:158-184, reid/train/trainer.py:107-170.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ptr
from . import engine
from .engine import _call, _new, _plan, EvalPlan


class VerifyEvalPlan(EvalPlan):
    def __init__(self, head):
        super().__init__(head)
        self.scale, self.shift = self.fold(head.classifierBN)
        self.w = head.classifierlinear.weight.detach().contiguous()
        self.b = head.classifierlinear.bias.detach().contiguous()


_vplans = {}


def _verify_eval(head, probe, gallery):
    key = id(head)
    p = _vplans.get(key)
    if p is None or p.key != engine._state_key(head):
        p = VerifyEvalPlan(head)
        _vplans[key] = p
    nb, k = probe.shape
    ncls = p.w.shape[0]
    out = _new((nb, gallery.shape[0], ncls), probe)
    _call('grl_pair_verify', ptr(probe), ptr(gallery), ptr(p.scale), ptr(p.shift), ptr(p.w),
          ptr(p.b), ptr(out), nb, gallery.shape[0], k, ncls)
    return out


def _train_not_ready(what):
    raise NotImplementedError(
        '%s: the train-mode (batch-stat BN + HIP backward) path is not built yet; '
        'call .eval() for the inference path' % what)


def grl_forward_train(model, inputs):
    _train_not_ready('ResNet50_GRL_Model.forward')


def siamese_self_attention_train(siam, x):
    _train_not_ready('Siamese.self_attention')


def siamese_forward(siam, x):
    """Siamese.forward: de-interleave (probe, gallery) pairs, pool each half with
    the temporal attention, verification head on all probe x gallery pairs."""
    if siam.training:
        _train_not_ready('Siamese.forward')
    with torch.no_grad():
        bsz, t, d = x.shape
        xv = x.contiguous().view(bsz // 2, 2, t, d)
        out = _new((bsz, d), x)
        half = bsz // 2
        engine._attn_into(siam, xv[:, 0].contiguous(), out[:half], d)
        engine._attn_into(siam, xv[:, 1].contiguous(), out[half:], d)
        cls = _verify_eval(siam, out[:half], out[half:])
        return cls, out


def siamese_video_forward(head, x):
    """Siamese_video.forward on pooled [B,D] features."""
    if head.training:
        _train_not_ready('Siamese_video.forward')
    with torch.no_grad():
        bsz = x.shape[0]
        xv = x.contiguous().view(bsz // 2, 2, -1)
        out = torch.cat((xv[:, 0], xv[:, 1])).contiguous()
        half = bsz // 2
        cls = _verify_eval(head, out[:half], out[half:])
        return cls, out
