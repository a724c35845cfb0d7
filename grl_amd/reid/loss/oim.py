"""Online Instance Matching loss (reference: reid/loss/oim.py:8-53) on MI355X.

forward : logits = scalar * x . LUT^T (the MFMA GEMM kernel, scale in the epilogue), then
          `grl_softmax_ce` = F.cross_entropy(mean) AND its gradient in one launch.
backward: grad_x = g * scalar * dlogits . LUT (`grl_oim_grad`) with the LUT AS IT IS WHEN THE
          BACKWARD RUNS (oim.py:23 reads self.lut at backward time), then -- inside backward, as
          upstream -- the LUT rows of the batch labels are momentum-updated one sample at a time
          and re-normalised (`grl_oim_update`).  SEQTrainer uses one criterion twice per step
          (trainer.py:126,138): autograd runs the later (clip-level) node first, so the
          frame-level backward reads a LUT the clip-level update has already changed and stacks
          its own update on top.  tests/golden/oim.npz pins exactly that against the reference.

The reference's legacy non-static autograd Function cannot be applied on torch >= 1.5; this is
a static Function with the same maths, pinned to the reference's forward/backward bodies
(tests/golden/make_golden.py:oim_golden).
No torch op computes here and nothing syncs with the host; there is no CPU path.

Multi-process data parallel: every rank applies the updates of ALL ranks in rank order
(all_gather of the small (features, labels) block), so the LUTs stay identical without
a parameter broadcast."""
import ctypes as C

import torch
from torch import nn, autograd

from grl_amd import _lib
from grl_amd._lib import ptr, require_device, MATH_F32


def _call(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args, _lib.stream()), name)


def _labels(t, dev):
    if not torch.is_tensor(t):
        raise _lib.GrlHipError('labels must be a tensor')
    return t.to(device=dev, dtype=torch.int64).contiguous()


class OIM(autograd.Function):
    """(loss, scaled logits) = CE(scalar * x . LUT^T, targets); the logits are returned for
    the precision read-out only (trainer.py:127) and carry no gradient."""

    @staticmethod
    def forward(ctx, inputs, targets, lut, momentum, scalar, weight):
        from grl_amd import engine
        require_device(inputs, 'OIM inputs')
        require_device(lut, 'OIM lut')
        x = inputs.contiguous()
        y = _labels(targets, x.device)
        n, D = x.shape
        c = lut.size(0)
        scale = torch.full((c,), float(scalar), dtype=torch.float32, device=x.device)
        logits = torch.empty((n, c), dtype=torch.float32, device=x.device)
        engine.gemm(x, lut, logits, n, c, D, scale=scale, math=MATH_F32, kblock=True)      # (K-blocked: split over K, include/grl_hip.h)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dlogits = torch.empty_like(logits)
        ws = torch.empty(2 * n, dtype=torch.float32, device=x.device)
        if weight is not None:
            require_device(weight, 'OIM class weight')
            weight = weight.contiguous()
        # correct[0] = rows whose arg-max is the label (first index on ties, as the reference's topk read-out,
        # eva_functions.py:118-131): the trainer's precision comes out of the same launch instead of topk / eq / sum
        correct = torch.empty((), dtype=torch.float32, device=x.device)
        _call('grl_softmax_ce', ptr(logits), c, ptr(y), ptr(weight), n, c, ptr(loss), ptr(correct), ptr(dlogits), c,
              ptr(ws))
        ctx.save_for_backward(x, y, dlogits)
        ctx.lut, ctx.momentum, ctx.scalar = lut, momentum, float(scalar)
        ctx.mark_non_differentiable(logits, correct)
        return loss, logits, correct

    @staticmethod
    def backward(ctx, gloss, _glogits, _gcorrect):
        x, y, dlogits = ctx.saved_tensors
        lut, m = ctx.lut, ctx.momentum
        n, D = x.shape
        grad_inputs = None
        if ctx.needs_input_grad[0]:
            g = gloss.contiguous().float()
            grad_inputs = torch.empty_like(x)
            _call('grl_oim_grad', ptr(dlogits), lut.size(0), ptr(lut), ptr(g), C.c_float(ctx.scalar),
                  ptr(grad_inputs), n, lut.size(0), D)
        from grl_amd import dist as grl_dist
        xs, ys = grl_dist.gather_rank_order(x, y)
        _call('grl_oim_update', ptr(lut), ptr(xs), ptr(ys), xs.size(0), D, lut.size(0), C.c_float(m))
        return grad_inputs, None, None, None, None, None


def oim(inputs, targets, lut, momentum=0.5, scalar=1.0, weight=None):
    """(loss, scaled logits) as the reference's OIMLoss.forward; the logits carry ``grl_top1 = (correct rows, rows)``
    (device scalar) for the trainer's precision read-out."""
    loss, logits, correct = OIM.apply(inputs, targets, lut, momentum, scalar, weight)
    logits.grl_top1 = (correct, logits.size(0))
    return loss, logits


class OIMLoss(nn.Module):
    def __init__(self, num_features, num_classes, scalar=1.0, momentum=0.5,
                 weight=None, size_average=True):
        super(OIMLoss, self).__init__()
        self.num_features = num_features
        self.num_classes = num_classes
        self.momentum = momentum
        self.scalar = scalar
        self.weight = weight
        self.register_buffer('lut', torch.zeros(num_classes, num_features))
        self.size_average = size_average

    def forward(self, inputs, targets):
        return oim(inputs, targets, self.lut, self.momentum, self.scalar, self.weight)
