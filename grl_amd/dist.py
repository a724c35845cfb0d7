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


def shard_rows(n, rank, world):
    """[lo, hi) of this rank's contiguous block of n rows (blocks differ by at most one row)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_distmat(qf, gf, fn, group=None):
    """The evaluator's query x gallery matrix with the gallery rows sharded over the ranks
    (SURVEY.md 8(e)): every rank runs ``fn(qf, gf[lo:hi])`` -- an independent GEMM, e.g.
    engine.cosin_dist -- and the column blocks are all-gathered (RCCL on the GPU node).  All ranks
    hold the full qf / gf; identity when not distributed."""
    if not is_distributed():
        return fn(qf, gf)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ng = gf.size(0)
    lo, hi = shard_rows(ng, rank, world)
    width = -(-ng // world)                           # equal-sized padded blocks for all_gather
    block = qf.new_zeros((qf.size(0), width))
    if hi > lo:
        block[:, :hi - lo] = fn(qf, gf[lo:hi].contiguous())
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block, group=group)
    cols = [blocks[r][:, :shard_rows(ng, r, world)[1] - shard_rows(ng, r, world)[0]] for r in range(world)]
    return torch.cat(cols, 1)


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
