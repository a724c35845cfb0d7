"""Data-parallel plumbing for training and evaluation: one process per GPU, torch.distributed
with the `nccl` backend (= RCCL over xGMI on ROCm) on the GPU node, `gloo` in the CPU tests.

The reference has no distributed code: it wraps the CNN in a single-process
``nn.DataParallel`` (mars_train.py:80).  Here

* the global batch is split at PAIR granularity (``shard_pairs`` / ``ShardedPairSampler`` /
  ``PairShardedBatches``: Siamese.forward needs interleaved (anchor, positive) rows,
  Siamese.py:116, sampler.py:104-123);
* every rank keeps its own BatchNorm statistics (= DataParallel's per-replica BN);
* the only exchange of a step is the gradient average, ``GradSync``: the parameter gradients
  of a step live in ONE flat fp32 buffer per tape (train_engine.Tape.reserve_param_grads; 54.76 M
  values = 219 MB in all), cut into four contiguous buckets that are all-reduced
  ASYNCHRONOUSLY as the backward finishes them -- TRL + tail (92 MB) first, then layer 4 + GCE,
  layer 3, and layers 2/1 + stem -- so that on an 8-GPU xGMI node all but the last few MB ride
  under the remaining backward kernels (RCCL runs a 219 MB all-reduce as reduce-scatter +
  all-gather over all 7 links in ~0.4 ms; a ring would take ~2.5 ms);
* the OIM look-up tables stay identical across ranks by replaying every rank's (feature, label)
  block in rank order (``gather_rank_order``);
* evaluation: clips are independent (no collective); the query x gallery matrix shards by
  gallery rows (``sharded_distmat``, used by ATTEvaluator.evaluate when distributed).

Documented difference from the reference's single-process DataParallel run: upstream only the
CNN is replicated -- Siamese, the pair-verification matrix (n^2 BCE terms), the batch-hard
triplet mining and classifierBN see the GLOBAL batch on device 0 (mars_train.py:80-82,
trainer.py:137-162).  Here every rank builds them from its LOCAL pairs and the gradients are
averaged: cross-rank negatives are not mined.  That is what north_star prescribes ("all-reduce
on the gradient step only"); it is not bit-equivalent to a 2-GPU DataParallel run.
"""
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # RCCL across processes needs dmabuf IPC on this driver

import torch
import torch.distributed as dist


def is_distributed():
    """More than one rank -- or GRL_SYNC_FORCE=1 with an initialised group of one (tests: the only way to run the
    RCCL call sequence on a single-GPU box; every collective is then an identity)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('GRL_SYNC_FORCE') == '1'


# ----------------------------------------------------------------------------
# batch sharding
# ----------------------------------------------------------------------------
def shard_pairs(batch_size, rank, world):
    """Index range [lo, hi) of this rank's rows of a global batch of interleaved pairs."""
    if batch_size % 2:
        raise RuntimeError("the batch size should be even number!")
    pairs = batch_size // 2
    if pairs % world:
        raise ValueError('global batch of %d pairs does not split over %d ranks' % (pairs, world))
    per = pairs // world
    return 2 * per * rank, 2 * per * (rank + 1)


def _rank_world(rank, world):
    if rank is None or world is None:
        if is_distributed():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1
    return rank, world


class ShardedPairSampler(torch.utils.data.Sampler):
    """Wraps a sampler that emits consecutive (index, cross-camera positive) pairs -- the
    reference's RandomPairSamplerForMars (sampler.py:83-125) -- and keeps, of every run of
    ``global_batch`` indices, this rank's pair shard.  All ranks must iterate the base sampler
    with the same seed; the DataLoader on top uses batch_size = global_batch // world, so a rank
    decodes only its own clips and every local batch is whole pairs."""

    def __init__(self, sampler, global_batch, rank=None, world=None):
        self.sampler, self.global_batch = sampler, global_batch
        self.rank, self.world = _rank_world(rank, world)
        self.lo, self.hi = shard_pairs(global_batch, self.rank, self.world)

    def __iter__(self):
        run = []
        for idx in self.sampler:
            run.append(idx)
            if len(run) == self.global_batch:
                for j in run[self.lo:self.hi]:
                    yield j
                run = []
        # a trailing partial global batch is dropped (the reference loader uses drop_last=True)

    def __len__(self):
        return len(self.sampler) // self.global_batch * (self.hi - self.lo)


class PairShardedBatches(object):
    """For a loader that already yields GLOBAL batches (imgs, pids, camids) on every rank: keep
    this rank's pair shard of each batch (simple, but every rank decodes the whole batch --
    prefer ShardedPairSampler)."""

    def __init__(self, loader, rank=None, world=None):
        self.loader = loader
        self.rank, self.world = _rank_world(rank, world)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for imgs, pids, cams, *extra in self.loader:
            lo, hi = shard_pairs(len(pids), self.rank, self.world)
            yield (imgs[lo:hi], pids[lo:hi], cams[lo:hi]) + tuple(e[lo:hi] for e in extra)


class _RankBatchSampler(object):
    """Every world-th batch of a batch sampler, starting at ``rank``: the rank's DataLoader workers then decode
    only this rank's clips (filtering a loader's OUTPUT would still decode every batch on every rank)."""

    def __init__(self, batch_sampler, rank, world):
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world

    def __iter__(self):
        for i, idx in enumerate(self.batch_sampler):
            if i % self.world == self.rank:
                yield idx

    def __len__(self):
        n = len(self.batch_sampler)
        return (n - self.rank + self.world - 1) // self.world if n > self.rank else 0


def shard_loader_batches(loader, rank=None, world=None):
    """Iterable over batches ``rank, rank + world, ...`` of ``loader`` (evaluation: batch i belongs to rank
    i % world).  A torch DataLoader is re-built around a rank-filtered batch sampler -- same dataset, workers,
    collate function, generator, multiprocessing context, pin-memory device -- so a rank's host side loads
    1/world of the data; any other iterable is filtered.

    "Batch i % world" partitions the data only if every rank enumerates the SAME batch list.  A sequential sampler
    does; a shuffling one (RandomSampler, a weighted sampler) draws its own permutation per process unless all
    ranks share a seeded generator -- such loaders are NOT re-built: their OUTPUT is filtered instead (every rank
    then decodes every batch, but no clip is duplicated or lost only if the caller seeds them identically, which
    is the caller's contract either way)."""
    rank, world = _rank_world(rank, world)
    if world == 1:
        return loader
    bs = getattr(loader, 'batch_sampler', None)
    sequential = isinstance(getattr(bs, 'sampler', None), torch.utils.data.SequentialSampler)
    if isinstance(loader, torch.utils.data.DataLoader) and bs is not None and sequential:
        kw = dict(num_workers=loader.num_workers, collate_fn=loader.collate_fn, pin_memory=loader.pin_memory,
                  timeout=loader.timeout, worker_init_fn=loader.worker_init_fn, generator=loader.generator,
                  multiprocessing_context=loader.multiprocessing_context if loader.num_workers > 0 else None,
                  pin_memory_device=getattr(loader, 'pin_memory_device', ''))
        if loader.num_workers > 0:
            kw.update(prefetch_factor=loader.prefetch_factor, persistent_workers=loader.persistent_workers)
        mine = _RankBatchSampler(bs, rank, world)
        out = torch.utils.data.DataLoader(loader.dataset, batch_sampler=mine, **kw)
        assert len(out) == len(mine) == len(range(rank, len(bs), world))
        return out
    return (b for i, b in enumerate(loader) if i % world == rank)


def gather_rank_order(x, y, group=None):
    """(features, labels) of every rank concatenated in rank order -- the OIM look-up tables
    replay all ranks' updates in that order so they stay identical without a broadcast
    (oim.py:24-26 is order dependent for repeated labels).  Identity when not distributed."""
    if not is_distributed() or global_heads():      # (global heads: the block every rank holds is the global one already)
        return x, y
    world = dist.get_world_size(group)
    xl = [torch.empty_like(x) for _ in range(world)]
    yl = [torch.empty_like(y) for _ in range(world)]
    _all_gather(xl, x.contiguous(), group)
    _all_gather(yl, y.contiguous(), group)
    return torch.cat(xl), torch.cat(yl)


def global_heads():
    """GRL_DP_GLOBAL_HEADS=1 (opt-in DP fidelity switch): the Siamese heads and the loss block see the GLOBAL batch, as
    in the reference's single-process nn.DataParallel run where only the CNN is replicated and Siamese, the n^2
    verification terms, batch-hard triplet mining and classifierBN run on the gathered batch on device 0
    (mars_train.py:80-82, trainer.py:137-162).  Default (north_star): rank-local heads, gradient all-reduce only."""
    return is_distributed() and os.environ.get('GRL_DP_GLOBAL_HEADS') == '1'


class _AllGatherGrad(torch.autograd.Function):
    """x [b_local, ...] -> the rank-ordered global batch [world * b_local, ...] on every rank.  Backward: every rank
    holds the (identical) gradient of the global loss w.r.t. the global tensor and keeps its own slice, multiplied by
    the world size -- GradSync AVERAGES parameter gradients afterwards, and sum_r J_r^T g_r is the gradient of the
    global loss (the head parameters see identical full gradients on every rank; their average is that gradient)."""

    @staticmethod
    def forward(ctx, x, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(world)]
        _all_gather(parts, x, group)
        ctx.rank, ctx.world, ctx.n = rank, world, x.size(0)
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        lo = ctx.rank * ctx.n
        return g[lo:lo + ctx.n] * float(ctx.world), None


def gather_global(x, group=None):
    """Differentiable rank-ordered all-gather along dim 0 (identity when not distributed)."""
    if not is_distributed():
        return x
    if x.requires_grad:
        return _AllGatherGrad.apply(x, group)
    parts = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
    _all_gather(parts, x.contiguous(), group)
    return torch.cat(parts, 0)


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
    _all_gather(blocks, block, group)
    cols = [blocks[r][:, :shard_rows(ng, r, world)[1] - shard_rows(ng, r, world)[0]] for r in range(world)]
    return torch.cat(cols, 1)


def gather_feature_batches(mine, n_batches, group=None):
    """Evaluation feature extraction sharded by batch (batch i belongs to rank i % world; clips are
    independent): ``mine`` = [(batch index, features [r_i, D], pids, camids)] of this
    rank -> the full, batch-ordered (features, pids, camids) on every rank.  One padded all_gather
    of the feature rows (RCCL) and one all_gather_object of the small id lists."""
    if not is_distributed():
        mine = sorted(mine, key=lambda e: e[0])
        return torch.cat([e[1] for e in mine], 0), [x for e in mine for x in e[2]], [x for e in mine for x in e[3]]
    world = dist.get_world_size(group)
    meta = [None] * world
    dist.all_gather_object(meta, [(i, f.size(0), list(p), list(c)) for i, f, p, c in mine], group=group)
    rows = [sum(m[1] for m in r) for r in meta]
    ref = mine[0][1] if mine else None
    dims = [None] * world
    dist.all_gather_object(dims, None if ref is None else (ref.size(1), str(ref.device)), group=group)
    D = next(d[0] for d in dims if d is not None)
    dev = ref.device if ref is not None else torch.device(next(d[1] for d in dims if d is not None))
    width = max(rows)
    block = torch.zeros((width, D), dtype=torch.float32, device=dev)
    if mine:
        block[:rows[dist.get_rank(group)]] = torch.cat([e[1] for e in mine], 0)
    blocks = [torch.empty_like(block) for _ in range(world)]
    _all_gather(blocks, block, group)
    pieces = {}
    for r in range(world):
        off = 0
        for i, nrow, pids, cams in meta[r]:
            pieces[i] = (blocks[r][off:off + nrow], pids, cams)
            off += nrow
    order = sorted(pieces)
    assert order == list(range(n_batches)), 'gather_feature_batches: missing batches'
    return (torch.cat([pieces[i][0] for i in order], 0), [x for i in order for x in pieces[i][1]],
            [x for i in order for x in pieces[i][2]])


# ----------------------------------------------------------------------------
# collectives (gloo cannot take device tensors on every build: stage through the host there)
# ----------------------------------------------------------------------------
def _host_staged(t, group):
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _all_gather(outs, t, group=None):
    if _host_staged(t, group):
        houts = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(houts, t.cpu(), group=group)
        for o, h in zip(outs, houts):
            o.copy_(h)
    else:
        dist.all_gather(outs, t, group=group)


class _HostWork(object):
    """all-reduce of a device tensor through the host (gloo test backend): completes at wait()."""

    def __init__(self, t, group):
        self.t, self.group = t, group

    def wait(self):
        h = self.t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        self.t.copy_(h)


# ----------------------------------------------------------------------------
# gradient averaging
# ----------------------------------------------------------------------------
class GradSync(object):
    """Bucketed, backward-overlapped gradient averaging.

    ``begin()`` (any time before ``loss.backward()`` -- before or after the forward) registers the object with
    train_engine; while the backward runs, every tape first declares the parameters whose gradients live in its
    flat buffer (``own``, from Tape.backward) and then hands over contiguous slices of that buffer as soon as all
    gradients inside are final (``reduce``): an asynchronous all-reduce is launched on each -- torch's NCCL/RCCL
    work stream first waits for the kernels already queued on the compute stream, i.e. the producers of that
    slice, and the backward kernels that follow overlap the transfer.  ``finish()`` (before
    ``optimizer.step()``) waits for the collectives and makes sure every ``p.grad`` holds the averaged values
    (autograd normally adopts the tape's views as ``p.grad``, in which case nothing is copied).  On RCCL the
    average is taken by the collective itself (ReduceOp.AVG: no scaling pass over the 219 MB afterwards); gloo
    (tests) sums and scales.  ``abort()`` drops the step's state without issuing a collective (error path).

    Parameters that never receive a gradient (Siamese.featV*, the unused uncorr verification
    head) sit in the flat buffers as zeros on every rank; their ``p.grad`` stays None everywhere --
    ``finish`` asserts once that the None-pattern is identical across ranks.

    Bookkeeping of the last step for tests / the bench line: ``launched`` [(label, numel)] per bucket,
    ``stray`` = number of per-parameter fall-back reductions (gradients no tape declared: none on this path),
    ``collectives`` = every collective issued (buckets + strays + the one-off mask check)."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if is_distributed() else 1
        # GRL_SYNC_FORCE=1 (tests): issue the collectives even in a world of one rank -- the only way to run the
        # RCCL call sequence (async all-reduce of flat-buffer slices under the backward) on a single-GPU box
        self.force = os.environ.get('GRL_SYNC_FORCE') == '1' and dist.is_available() and dist.is_initialized()
        self.avg_op = (self.world > 1 or self.force) and dist.get_backend(group) == 'nccl'
        self._works = []          # (work, flat slice, label)
        self.timing = False       # bench.py: HIP events around every bucket's wait in finish() -> exposed_ms()
        self._waits = []          # (label, event before the wait, event after it) of the last finish()
        self._owned = []          # (param, flat, offset) of every tape-owned gradient of this step
        self._checked = False
        self.launched = []        # (label, numel) per collective of the last step (tests / logging)
        self.stray = 0
        self.collectives = 0

    # -- protocol with train_engine.Tape -------------------------------------------------------
    def begin(self):
        from . import train_engine
        train_engine.set_grad_sync(self)
        self._works, self._owned, self.launched = [], [], []
        self.stray = self.collectives = 0

    def own(self, param, flat, offset):
        self._owned.append((param, flat, offset))

    def reduce(self, piece, label=''):
        """Average ``piece`` (a contiguous slice of a flat gradient buffer) over the ranks; returns
        at once, the result is valid after ``finish()``."""
        self.launched.append((label, piece.numel()))
        if (self.world == 1 and not self.force) or piece.numel() == 0:
            return
        self.collectives += 1
        if _host_staged(piece, self.group):
            self._works.append((_HostWork(piece, self.group), piece, label))
        else:
            op = dist.ReduceOp.AVG if self.avg_op else dist.ReduceOp.SUM
            self._works.append((dist.all_reduce(piece, op=op, group=self.group, async_op=True), piece, label))

    def abort(self):
        """Error path (the backward raised on this rank): forget the step WITHOUT issuing any collective -- the
        peers are at an unknown point of the sequence, a blocking all-reduce here would hang and mask the error."""
        from . import train_engine
        train_engine.set_grad_sync(None)
        self._works, self._owned = [], []

    def exposed_ms(self):
        """[(bucket label, ms the launch stream waited for that bucket in the last finish())] -- what the backward
        did not cover.  Needs ``timing = True``; call it after the stream has been synchronised (bench.py reads it
        after the timed region: nothing here blocks the host inside a step)."""
        return [(lab, e0.elapsed_time(e1)) for lab, e0, e1 in self._waits]

    def finish(self):
        from . import train_engine
        train_engine.set_grad_sync(None)
        inv = 1.0 / self.world
        self._waits = []
        for work, piece, label in self._works:
            timed = self.timing and piece.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            work.wait()
            if timed:
                e1.record()
                self._waits.append((label, e0, e1))
            if not self.avg_op:
                piece.mul_(inv)
        self._works = []
        stray = []
        for p, flat, off in self._owned:
            if p.grad is None:
                continue
            if p.grad.data_ptr() != flat.data_ptr() + 4 * off:      # autograd copied instead of adopting the view
                p.grad.copy_(flat[off:off + p.numel()].view_as(p))
        owned = set(id(p) for p, _, _ in self._owned)
        for p in self.params:                                       # gradients no tape owns (none on this path)
            if id(p) not in owned and p.grad is not None:
                stray.append(p)
        self.stray = len(stray)
        if stray and self.world > 1:
            for p in stray:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                p.grad.mul_(inv)
                self.collectives += 1
        if not self._checked and (self.world > 1 or self.force):
            mask = torch.tensor([0 if p.grad is None else 1 for p in self.params], dtype=torch.int32)
            if dist.get_backend(self.group) != 'gloo':              # RCCL reduces device tensors only
                mask = mask.to(self.params[0].device)
            lo, hi = mask.clone(), mask.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            self.collectives += 2
            if not torch.equal(lo, hi):
                raise RuntimeError('GradSync: the set of parameters that receive gradients differs across ranks')
            self._checked = True
        self._owned = []
