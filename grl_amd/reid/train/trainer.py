"""SEQTrainer with the reference's constructor / train() signature and loss
composition (/root/reference/reid/train/trainer.py:16-177).  The CNN / Siamese
forward and backward run on MI355X through grl_amd (autograd.Functions around the HIP
kernels); the only addition is the data-parallel gradient all-reduce between
``backward()`` and ``step()`` when torch.distributed is initialised (one process per
GPU; RCCL over xGMI).  The loss block (OIM cross entropy, pair BCE, batch-hard triplet) is
HIP too (grl_amd/csrc/loss.hip); torch only adds the five scalars."""
import os
import time

import torch

from grl_amd import dist as grl_dist
from grl_amd.reid.evaluator import accuracy
from grl_amd.reid.loss import TripletLoss
from grl_amd.reid.loss.pairloss import pair_prob
from grl_amd.utils.meters import AverageMeter

try:                                            # tensorboardX is optional here
    from tensorboardX import SummaryWriter
except Exception:                               # pragma: no cover
    class SummaryWriter(object):
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

criterion_triplet = TripletLoss('soft', True)

HEAD_STREAMS = __import__('os').environ.get('GRL_HEAD_STREAMS', '1') != '0'
_head_streams = {}


class _HeadFork(object):
    """``with _HeadFork(x) as fk:`` runs the block on a side HIP stream forked from the current one; ``fk.join(*results)``
    makes the launch stream wait for it (no-op on CPU tensors or with GRL_HEAD_STREAMS=0).

    Allocator hygiene: ``x`` was allocated on the launch stream and is read on the side stream, the block's results are
    allocated on the side stream and consumed on the launch stream -- both are recorded on the other stream
    (``record_stream``), so that an early free (a ``no_grad`` forward, an exception between fork and join) cannot
    hand a block back to one stream's pool while the other still has work on it in flight.  An exception inside the
    block still orders the streams: ``__exit__`` joins when it sees one."""

    def __init__(self, x):
        self.on = HEAD_STREAMS and x.is_cuda
        self.joined = False
        if self.on:
            key = x.device.index if x.device.index is not None else torch.cuda.current_device()
            if key not in _head_streams:
                _head_streams[key] = torch.cuda.Stream(x.device)
            self.side, self.main = _head_streams[key], torch.cuda.current_stream(x.device)
            self.ctx = torch.cuda.stream(self.side)
            self.x = x

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            self.x.record_stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            if exc[0] is not None:
                self.join()
        return False

    def join(self, *results):
        """The launch stream waits for the side stream (call before the first use of the block's results and pass
        them: tensors allocated on the side stream that the launch stream goes on to consume)."""
        if self.on and not self.joined:
            self.main.wait_stream(self.side)
            self.joined = True
        if self.on:
            for r in results:
                if torch.is_tensor(r) and r.is_cuda:
                    r.record_stream(self.main)


LAZY_METERS = os.environ.get('GRL_LAZY_METERS', '1') != '0'
_gc_frozen = [False]


def _freeze_collector_once():
    """After the first completed step: ``gc.freeze()``.  A train step is ~1500 HIP launches issued from Python and, on
    bf16 storage, as long as the host needs to issue them; every pause of Python's cyclic collector is step time.  A full
    collection walks every container alive in the process -- model, optimizer state, the dataset's tracklet lists -- and
    the step's tape of closures makes the collector run often: measured 40-110 ms pauses every ~40 steps, 19.4 against
    17.8 ms per step over 60 steps (tools/jpeg_feed_order.py --gc default / freeze --trace).  freeze() moves what is alive
    now (all long-lived) out of the collector's sight; the garbage of later steps is still collected.
    GRL_GC_FREEZE=0 leaves the collector alone."""
    if _gc_frozen[0]:
        return
    _gc_frozen[0] = True
    if os.environ.get('GRL_GC_FREEZE', '1') != '0':
        import gc
        gc.collect()
        gc.freeze()


class BaseTrainer(object):
    def __init__(self, model, criterion):
        self.model = model
        self.criterion_ver = criterion
        self.criterion_ver_uncorr = criterion
        self.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
        self._bucket = None
        self._meter_ring = ([torch.empty(4, dtype=torch.float32, pin_memory=True) for _ in range(4)]
                            if torch.cuda.is_available() else None)

    def _all_params(self):
        raise NotImplementedError

    def train(self, epoch, data_loader, optimizer1):
        self.model.train()
        batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter()
        precisions, precisions1, precisions2 = AverageMeter(), AverageMeter(), AverageMeter()
        end = time.time()
        from grl_amd import engine
        batches = data_loader
        if grl_dist.is_distributed() and not getattr(data_loader, 'grl_rank_sharded', False):
            # a loader that yields GLOBAL batches on every rank: keep this rank's pair shard
            # (loaders built on dist.ShardedPairSampler set .grl_rank_sharded = True)
            batches = grl_dist.PairShardedBatches(data_loader)
        if torch.device(self.device).type == 'cuda':     # next batch's H2D copy under this step's kernels
            batches = engine.DevicePrefetcher(batches, self.device)
        # The step's four scalars (loss + three precisions) feed the meters and the writer exactly as upstream
        # (trainer.py:70-77, :85-87) -- but they are READ one step late: `loss.item()` right after the forward stalls the
        # host until the forward has run, and the step is as long as its ~1500 launches take to issue (bf16 storage: 19.4
        # ms per iteration against 17.8 for the step itself).  The four values go to a pinned host buffer with one stacked
        # async copy behind the forward; the next iteration (or the print / the end of the epoch) picks them up, by which
        # time the copy has long completed.  GRL_LAZY_METERS=0: read them at once, as upstream.
        lazy = LAZY_METERS and torch.device(self.device).type == 'cuda'
        pending = []

        def settle():
            while pending:
                host4, ev, n, num_iter = pending.pop(0)
                if ev is not None:
                    ev.synchronize()
                v = host4.tolist()
                losses.update(v[0], n)
                precisions.update(v[1], n)
                precisions1.update(v[2], n)
                precisions2.update(v[3], n)
                self.writer.add_scalar('train/total_loss_step', losses.val, num_iter)
                self.writer.add_scalar('train/total_loss_avg', losses.avg, num_iter)

        for i, inputs in enumerate(batches):
            data_time.update(time.time() - end)
            inputs, targets = self._parse_data(inputs)
            loss, uncorr_prec_id_vid, corr_prec_id_vid, corr_prec_id_frame = \
                self._forward(inputs, targets, i, epoch)
            num_iter = len(data_loader) * epoch + i
            four = torch.stack([v.detach().to(torch.float32).reshape(()) if isinstance(v, torch.Tensor)
                                else torch.tensor(float(v), dtype=torch.float32, device=loss.device)
                                for v in (loss, uncorr_prec_id_vid, corr_prec_id_vid, corr_prec_id_frame)])
            settle()                                      # the previous step's values
            if lazy:
                slot = self._meter_ring[i % len(self._meter_ring)]
                slot.copy_(four, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((slot, ev, targets.size(0), num_iter))
            else:
                pending.append((four.cpu(), None, targets.size(0), num_iter))
                settle()

            optimizer1.zero_grad()
            # data parallel: the gradient buckets are all-reduced (RCCL) while the backward still
            # runs -- TRL + tail first, layers 2/1 + stem last -- and averaged before the step
            sync = None
            if grl_dist.is_distributed():
                if self._bucket is None:
                    self._bucket = grl_dist.GradSync(self._all_params())
                sync = self._bucket
                sync.begin()
            try:
                loss.backward()
            except BaseException:
                if sync is not None:
                    sync.abort()          # no collective on the error path: the peers are somewhere else in the sequence
                raise
            if sync is not None:
                sync.finish()
            optimizer1.step()

            _freeze_collector_once()
            batch_time.update(time.time() - end)
            end = time.time()
            if (i + 1) % 100 == 0:
                settle()
                print('Epoch: [{}][{}/{}]\t'
                      'Loss {:.3f} ({:.3f})\t'
                      'uncorr_vid {:.2%} ({:.2%})\t'
                      'corr_vid {:.2%} ({:.2%})\t'
                      'corr_frame {:.2%} ({:.2%})\t'
                      .format(epoch, i + 1, len(data_loader), losses.val, losses.avg,
                              precisions.val, precisions.avg, precisions1.val, precisions1.avg,
                              precisions2.val, precisions2.avg))
        settle()
        self.meters = dict(loss=losses, uncorr_vid=precisions, corr_vid=precisions1, corr_frame=precisions2)

    def _parse_data(self, inputs):
        raise NotImplementedError

    def _forward(self, inputs, targets, i, epoch):
        raise NotImplementedError


class SEQTrainer(BaseTrainer):
    def __init__(self, cnn_model, siamese_model, siamese_model_uncorr, criterion_veri,
                 criterion_corr, criterion_uncorr, logdir):
        super(SEQTrainer, self).__init__(cnn_model, criterion_veri)
        self.siamese_model = siamese_model
        self.siamese_model_uncorr = siamese_model_uncorr
        self.criterion_uncorr = criterion_uncorr
        self.criterion_corr = criterion_corr
        self.writer = SummaryWriter(log_dir=logdir)
        self.device = next(cnn_model.parameters()).device

    def _all_params(self):
        mods = (self.model, self.siamese_model, self.siamese_model_uncorr)
        return [p for m in mods for p in m.parameters()]

    def _parse_data(self, inputs):
        """(imgs, pids, camids) as upstream (trainer.py:99-105); a 4th element -- the augmentation
        parameter block of grl_amd.reid.data.augment for raw uint8 clips -- makes flip / erase /
        ToTensor / Normalize run on the device instead of in the loader's PIL transforms."""
        imgs, pids = inputs[0].to(self.device), inputs[1]
        if len(inputs) > 3:
            from grl_amd import engine
            imgs = engine.augment_normalize_u8(engine.rect_scale_u8(imgs), inputs[3])   # RectScale comes first
        return [imgs], pids.to(self.device)

    @staticmethod
    def _top1(output_id, target):
        """top-1 precision of the id logits (trainer.py:127,139,153: accuracy(output.data, target.data)[0]).  The OIM
        criterion's cross-entropy launch already counted the rows whose arg-max is the label (same tie rule as topk);
        any other criterion goes through the reference's accuracy()."""
        top1 = getattr(output_id, 'grl_top1', None)
        if top1 is not None:
            return top1[0] * (1.0 / top1[1])
        return accuracy(output_id.data, target.data)[0]

    @staticmethod
    def _pair_prob(encode_scores):
        """softmax over the two verification logits, class-1 column (trainer.py:146-148)."""
        return pair_prob(encode_scores)

    def _forward(self, inputs, targets, i, epoch):
        """trainer.py:107-170: id loss on frames + id loss on pooled clips (same LUT) +
        20 x pair verification + batch-hard triplet on the correlated branch, id loss on
        the uncorrelated branch."""
        batch_size, seq_len = inputs[0].size(0), inputs[0].size(1)
        x_uncorr, x_corr = self.model(inputs[0])
        if grl_dist.global_heads():
            # opt-in DP fidelity (GRL_DP_GLOBAL_HEADS=1): as under the reference's nn.DataParallel, only the CNN is
            # data parallel -- its outputs and the labels are gathered in rank order and the Siamese heads, the n^2
            # verification terms, triplet mining and the OIM tables see the GLOBAL batch (mars_train.py:80-82)
            x_uncorr, x_corr = grl_dist.gather_global(x_uncorr), grl_dist.gather_global(x_corr)
            targets = grl_dist.gather_global(targets)
            batch_size = x_corr.size(0)
        frame_corr = x_corr.reshape(batch_size * seq_len, -1)
        targetX = targets.unsqueeze(1).expand(batch_size, seq_len).reshape(-1)
        pairs = targets.data.view(batch_size // 2, -1)
        tar_probe, tar_gallery = pairs[:, 0], pairs[:, 1]
        target = torch.cat((tar_probe, tar_gallery))
        # The uncorrelated branch's head and id loss depend on nothing below: both heads are chains of small,
        # latency-bound launches, so this one is issued on a side HIP stream next to the correlated branch's (autograd
        # runs each node's backward on its forward stream, so the backward overlaps the same way); joined before the sum.
        fk = _HeadFork(x_uncorr)
        with fk:
            encode_scores_u, siamese_out_u = self.siamese_model_uncorr(x_uncorr)
            uncorr_id_loss_vid, output_id = self.criterion_uncorr(siamese_out_u, target)
            uncorr_prec_id_vid = self._top1(output_id, target)
        # (the reference also evaluates the uncorr verification loss but never adds it)
        corr_id_loss_frame, output_id = self.criterion_corr(frame_corr, targetX)
        corr_prec_id_frame = self._top1(output_id, targetX)

        encode_scores, siamese_out = self.siamese_model(x_corr)
        corr_id_loss_vid, output_id = self.criterion_corr(siamese_out, target)
        corr_prec_id_vid = self._top1(output_id, target)
        corr_loss_tri = criterion_triplet(siamese_out, target).mean()
        corr_loss_ver, _ = self.criterion_ver(self._pair_prob(encode_scores), tar_probe, tar_gallery)

        fk.join(uncorr_id_loss_vid, uncorr_prec_id_vid, siamese_out_u, encode_scores_u)
        corr_loss = corr_id_loss_frame + corr_id_loss_vid + corr_loss_ver * 20 + corr_loss_tri
        all_loss = uncorr_id_loss_vid + corr_loss
        return all_loss, uncorr_prec_id_vid, corr_prec_id_vid, corr_prec_id_frame

    def train(self, epoch, data_loader, optimizer1):
        self.siamese_model.train()
        self.siamese_model_uncorr.train()
        super(SEQTrainer, self).train(epoch, data_loader, optimizer1)
