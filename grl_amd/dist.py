"""Data-parallel plumbing for training: one process per GPU, torch.distributed with
the `nccl` backend (= RCCL over xGMI on ROCm) on the GPU node, `gloo` in the CPU tests.

The reference has no distributed code: it wraps the CNN in a single-process
``nn.DataParallel`` (mars_train.py:80).  Here the global batch is split at PAIR
granularity (Siamese.forward needs interleaved (anchor, positive) rows,
Siamese.py:116), every rank keeps its own BatchNorm statistics (= DataParallel's
per-replica BN) and the only exchange is ONE all-reduce per step of a flat fp32 bucket
holding every parameter gradient (54.76 M values = 219 MB): on a fully connected xGMI
node RCCL runs it as reduce-scatter + all-gather over all 7 links.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_pairs(batch_size, rank, world):
    """Index range [lo, hi) of this rank's rows of a global batch of interleaved pairs."""
    if batch_size % 2:
        raise RuntimeError("the batch size should be even number!")
    pairs = batch_size // 2
    if pairs % world:
        raise ValueError('global batch of %d pairs does not split over %d ranks' % (pairs, world))
    per = pairs // world
    return 2 * per * rank, 2 * per * (rank + 1)


def gather_rank_order(x, y, group=None):
    """(features, labels) of every rank concatenated in rank order -- the OIM look-up tables
    replay all ranks' updates in that order so they stay identical without a broadcast
    (oim.py:24-26 is order dependent for repeated labels).  Identity when not distributed."""
    if not is_distributed():
        return x, y
    world = dist.get_world_size(group)
    xl = [torch.empty_like(x) for _ in range(world)]
    yl = [torch.empty_like(y) for _ in range(world)]
    dist.all_gather(xl, x.contiguous(), group=group)
    dist.all_gather(yl, y.contiguous(), group=group)
    return torch.cat(xl), torch.cat(yl)


class GradBucket(object):
    """Flat gradient bucket.  Parameters that never receive a gradient on any rank
    (Siamese.featV*, the unused uncorr verification head) contribute zeros so the
    bucket layout is identical on every rank."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)

    def allreduce_mean(self, group=None):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            else:
                self.flat[off:off + n].zero_()
            off += n
        if is_distributed():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(self.flat[off:off + n].view_as(p.grad))
            off += n
        return self.flat
