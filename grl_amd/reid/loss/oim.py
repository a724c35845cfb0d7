"""Online Instance Matching loss (reference: reid/loss/oim.py:8-53), restated as a
static autograd.Function -- the reference's legacy non-static Function does not run
on torch >= 1.5 (PARITY UNPINNED for OIM).

forward: logits = x . LUT^T, scaled, cross-entropy.
backward: grad_x = g . LUT, then -- inside backward, as upstream -- the LUT rows of the
batch labels are momentum-updated one sample at a time and re-normalised.

Multi-process data parallel: every rank applies the updates of ALL ranks in rank order
(all_gather of the small (features, labels) block), so the LUTs stay identical without
a parameter broadcast.  These are torch ops on the device: the losses are host-side
glue in this round."""
import torch
import torch.nn.functional as F
from torch import nn, autograd


class OIM(autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, lut, momentum):
        ctx.save_for_backward(inputs, targets)
        ctx.lut, ctx.momentum = lut, momentum
        return inputs.mm(lut.t())

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, targets = ctx.saved_tensors
        lut, m = ctx.lut, ctx.momentum
        grad_inputs = grad_outputs.mm(lut) if ctx.needs_input_grad[0] else None
        xs, ys = inputs.detach(), targets
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            xl = [torch.empty_like(xs) for _ in range(dist.get_world_size())]
            yl = [torch.empty_like(ys) for _ in range(dist.get_world_size())]
            dist.all_gather(xl, xs.contiguous())
            dist.all_gather(yl, ys.contiguous())
            xs, ys = torch.cat(xl), torch.cat(yl)
        if lut.is_cuda:                               # one HIP launch, no host sync
            from grl_amd import _lib
            import ctypes as C
            xs, ys = xs.contiguous().float(), ys.contiguous().long()
            _lib.check(_lib.load().grl_oim_update(lut.data_ptr(), xs.data_ptr(), ys.data_ptr(), xs.size(0),
                                                  xs.size(1), C.c_float(m), _lib.stream()), 'grl_oim_update')
        else:
            for x, y in zip(xs, ys.tolist()):         # sequential per sample (oim.py:24-26)
                row = m * lut[y] + (1. - m) * x
                lut[y] = row / row.norm()
        return grad_inputs, None, None, None


def oim(inputs, targets, lut, momentum=0.5):
    return OIM.apply(inputs, targets, lut, momentum)


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
        inputs = oim(inputs, targets, self.lut, momentum=self.momentum)
        inputs = inputs * self.scalar
        loss = F.cross_entropy(inputs, targets, weight=self.weight)
        return loss, inputs
