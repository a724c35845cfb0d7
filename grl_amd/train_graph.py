"""One SEQTrainer step -- forward, the 5-term loss, the HIP backward and the optimizer update
(/root/reference/reid/train/trainer.py:53-55,107-170) -- captured ONCE into a HIP graph and replayed.

Why: the step is a static chain of ~1500 short launches.  In exact fp32 the GPU is the limit (60 ms of kernels behind
~25 ms of Python), but in the bf16-storage mode the kernels take ~21 ms and the host needs 23-28 ms to ISSUE them
(ctypes calls, tensor allocations, autograd bookkeeping; it varies with the box's CPU): the step is host-bound.  A
captured graph replays the same launches -- same kernels, same order, same side streams (the two TRL directions, the
weight-gradient stream: their event forks and joins become graph edges) -- with no Python in between, so the result
is bit-identical to the eager step and the step time is the GPU's.

Scope: one process, one GPU (the RCCL gradient exchange of the data-parallel step is not captured), a FIXED batch
shape, the model in train mode.  Inputs are copied into static buffers; the loss and the three precisions come back
as device tensors.  torch.cuda.CUDAGraph is the capture vehicle (our launches go to torch's capturing stream); no
tracing compiler is involved."""
import torch

from . import engine


class GraphedTrainStep(object):
    def __init__(self, trainer, optimizer, clips, pids, warmup=3, warmup_batches=None):
        """``trainer``: a grl_amd SEQTrainer; ``optimizer``: the torch optimizer over trainer._all_params().  The
        capture needs the optimizer's momentum buffers and the allocator's pools to exist, so real steps run first --
        they DO train: on ``warmup_batches`` [(clips, pids), ...] if given, else ``warmup`` times on (clips, pids).
        The capture itself executes nothing."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError('GraphedTrainStep captures a single-GPU step (the RCCL gradient exchange is not captured)')
        self.trainer, self.opt = trainer, optimizer
        self.mods = (trainer.model, trainer.siamese_model, trainer.siamese_model_uncorr)
        self.clips, self.pids = clips.clone(), pids.clone()
        side = torch.cuda.Stream(clips.device)
        side.wait_stream(torch.cuda.current_stream(clips.device))
        with torch.cuda.stream(side):                      # warm-up on a side stream, as torch's capture recipe asks
            for c, p in (warmup_batches if warmup_batches is not None else [(clips, pids)] * max(warmup, 1)):
                self.clips.copy_(c); self.pids.copy_(p)
                self._eager()
            self.clips.copy_(clips); self.pids.copy_(pids)
        torch.cuda.current_stream(clips.device).wait_stream(side)
        torch.cuda.synchronize()
        self.opt.zero_grad(set_to_none=True)               # the captured backward re-creates the gradients in the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.trainer._forward([self.clips], self.pids, 0, 0)
            self.out[0].backward()
            self.opt.step()
        self.steps = 0

    def _eager(self):
        out = self.trainer._forward([self.clips], self.pids, 0, 0)
        self.opt.zero_grad(set_to_none=True)
        out[0].backward()
        self.opt.step()
        return out

    def __call__(self, clips=None, pids=None):
        """One training step on (clips, pids) -- same shapes as at capture (None: the batch already in the static
        buffers).  Returns (loss, uncorr_prec_id_vid, corr_prec_id_vid, corr_prec_id_frame) as device tensors that the
        NEXT replay overwrites."""
        if clips is not None:
            if tuple(clips.shape) != tuple(self.clips.shape) or clips.dtype != self.clips.dtype:
                raise ValueError('GraphedTrainStep was captured for clips %s %s' % (tuple(self.clips.shape), self.clips.dtype))
            self.clips.copy_(clips, non_blocking=True)
        if pids is not None:
            self.pids.copy_(pids, non_blocking=True)
        self.graph.replay()
        for m in self.mods:                                # parameters / running statistics moved: eval plans must re-fold
            engine.touch_state(m)
        self.steps += 1
        return self.out
