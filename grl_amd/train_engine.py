"""Train-mode side of the GRL path on MI355X: batch-statistics BatchNorm forward and
the hand-written HIP backward, wired into torch.autograd through two
``autograd.Function``s so that ``loss.backward()`` in the trainer
(/root/reference/reid/train/trainer.py:53-55) runs our kernels.

A tiny tape (list of closures) records the launch order of the forward; backward
replays it in reverse.  Gradients live in plain device tensors keyed by the forward
tensor they belong to; parameter gradients are accumulated (BatchNorms of the TRL memo
block are applied T times per forward, grl_model.py:153,167).

Reference call sites: reid/models/grl_model.py:211-228 (+ basebranch.py:52-68,
resnets1.py:73-109), reid/models/Siamese.py:79-142, reid/models/Siamese_video.py:158-184.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib, engine
from ._lib import GrlWgrad, check, ptr
from ._lib import MATH_F32
from .engine import _call, _new, EvalPlan, PIX


# Multiplier datapath of the TRAINING forward and data-gradient GEMMs (operands stay fp32 in HBM,
# accumulation is fp32; the weight-gradient GEMMs, BatchNorm and the loss block are always fp32):
#   'f32'    exact fp32 MFMA with K-blocked accumulation -- the default and the mode every train
#            parity claim is made in (the eval-side engine.set_math / GRL_MATH never leaks in here);
#   'bf16x3' split-bf16 products (hi*hi + hi*lo + lo*hi): fp32-class gradients at ~3x the GEMM rate;
#   'bf16'   operands rounded to bf16 while staging (BASELINE configs[2]'s "bf16 MFMA" datapath for
#            training): gradients to ~1e-2, forward + data-gradient GEMMs ~5x faster;
#   'mixed'  FORWARD exact fp32 (K-blocked: activations, batch statistics and -- what decides the accuracy of
#            the parameter gradients -- the ReLU masks are those of 'f32', bit for bit), BACKWARD GEMMs (data and
#            weight gradients) split-bf16: every product carries ~2^-16 relative error instead of 2^-24, no
#            mask can flip, so the gradients stay within ~1e-5 of the 'f32' ones at ~2/3 of the step time.
#   'bf16s'  bf16 STORAGE (BASELINE configs[2], "bf16 MFMA", as the training batch it describes): every
#            [pixels][channels] activation, saved tensor and activation gradient of the CNN is bf16 in HBM, every
#            GEMM -- forward, data gradient, weight gradient -- runs bf16 products with fp32 accumulation;
#            BatchNorm statistics (taken from the fp32 accumulators), per-channel vectors, the per-clip heads,
#            parameter gradients, master weights and the optimizer stay fp32.  Kernels: train_bf16.hip, the
#            bf16-in weight gradient of train.hip, the bf16-storage GEMM with the statistics epilogue.
_TRAIN_MATH = {'f32': (MATH_F32, MATH_F32), 'bf16x3': (engine.MATH_BF16X3, engine.MATH_BF16X3),
               'bf16': (engine.MATH_BF16, engine.MATH_BF16), 'mixed': (MATH_F32, engine.MATH_BF16X3),
               'bf16s': (MATH_F32, MATH_F32)}       # (bf16s: the datapath follows the operand dtype, see gemm below)
BF16 = torch.bfloat16
_train_mode = [__import__('os').environ.get('GRL_TRAIN_MATH', 'f32')]
_in_backward = [False]   # a tape is replaying (Tape.backward): see gemm()
_BWD_KBLOCK = __import__('os').environ.get('GRL_BWD_KBLOCK') == '1'      # A/B only: K-blocked data gradients as in round 2
_train_math = [_TRAIN_MATH[_train_mode[0]][0]]       # datapath of the GEMMs issued NOW (Tape.backward switches it)


def set_math(name):
    """Select the training GEMM datapath ('f32' | 'mixed' | 'bf16x3' | 'bf16'); returns the previous name."""
    old = _train_mode[0]
    _train_math[0] = _TRAIN_MATH[name][0]
    _train_mode[0] = name
    return old


def get_math():
    return _train_mode[0]


def gemm(a, w, *args, **kw):
    if a.dtype == BF16:                  # bf16-storage operands: the bf16-storage datapath, whatever the mode says
        kw['math'] = engine.MATH_BF16S
        if w.dtype != BF16:
            w = cast16(w)
        return engine.gemm(a, w, *args, **kw)
    kw.setdefault('math', _train_math[0])
    if kw['math'] == MATH_F32:
        # K-blocked accumulation (include/grl_hip.h: GrlGemm.kblock) in the FORWARD -- it decides the ReLU masks and
        # with them the parameter gradients' agreement with the reference -- and for the skinny GEMMs of either pass
        # (M <= 256: K-blocked is what lets them split over K).  The data-gradient GEMMs of the backward keep the one
        # chain: no mask depends on them, and the K-blocked kernels cost ~4 % on K >= 1024 (round 3).
        kw.setdefault('kblock', ((not _in_backward[0]) or args[1] <= 256 or _BWD_KBLOCK) and kw.get('bn') is None)
    return engine.gemm(a, w, *args, **kw)


def _b16(t):
    return t is not None and t.dtype == BF16


def _k(name, t):
    """Entry point for tensor ``t``'s storage type: the bf16 twin (train_bf16.hip / pointwise_bf16.hip) or the fp32 one."""
    return name + '_bf16' if t.dtype == BF16 else name


def _newl(shape, like):
    return torch.empty(shape, dtype=like.dtype, device=like.device)


def cast16(t):
    """fp32 -> bf16 copy (weights: once per step through Tape.w16)."""
    src = t if t.is_contiguous() else t.contiguous()
    out = torch.empty(src.shape, dtype=BF16, device=src.device)
    _call('grl_cast_bf16', ptr(src), ptr(out), src.numel())
    return out


def add_rowbcast(dst, v, M, Cc, rpg, scale, acc):
    """dst[m][c] (+)= v[m / rpg][c] * scale (grl_add_rowbcast / its bf16-storage twin)."""
    if dst.dtype == BF16:
        _call('grl_add_rowbcast_bf16', ptr(dst), ptr(v), M, Cc, rpg, C.c_float(scale), acc, 1 if v.dtype == BF16 else 0)
    else:
        _call('grl_add_rowbcast', ptr(dst), ptr(v), M, Cc, rpg, C.c_float(scale), acc)

FRAME_C = 2048


# ----------------------------------------------------------------------------
# tape
# ----------------------------------------------------------------------------
_grad_sync = [None]      # grl_amd.dist.GradSync of the running step (data-parallel training), or None


def set_grad_sync(sync):
    _grad_sync[0] = sync


class _Ops(list):
    """The tape's closure list.  A closure recorded while ``tape.side`` is set (the second TRL direction,
    issued on a side HIP stream) is replayed on that stream too."""

    def __init__(self, tape):
        list.__init__(self)
        self._tape = weakref.ref(tape)       # weak: Tape <-> _Ops must not be a cycle, or every finished step's
                                             # flat gradient buffer waits for a full garbage collection

    def append(self, fn):
        side = self._tape().side
        if side is not None:
            inner = fn

            def fn():
                with torch.cuda.stream(side):
                    inner()
        list.append(self, fn)


class WeightPrep(object):
    """All weight re-layouts / bf16 casts of one training step of one model in ONE launch (grl_weight_prep).

    The first step of a model in a storage mode runs the per-layer path (grl_pack_conv_weight, grl_transpose,
    grl_pack_dgrad_weight, grl_cast_bf16, ...) and LOGS, for every derived weight whose source is a parameter, the
    tape's cache key and the gather that produces it.  From the second step on the table is launched once at the start
    of the forward and the tape's weight cache is pre-filled with the persistent destination buffers -- the same values
    bit for bit (pure data movement, the same bf16 rounding), ~90 (fp32) / ~230 (bf16 storage) launches fewer per step.
    Re-built if a parameter's storage moved (``.to()``, a new ``Parameter``)."""

    def __init__(self):
        self.log = []          # (key, param, base, dims, strides, tiled, out_bf16, shape) while recording
        self.seen = set()
        self.ready = False
        self.cache = {}        # key -> persistent tensor
        self.table = None
        self.count = 0
        self.src = []          # (param, data_ptr at build time)

    def record(self, key, src, dims, strides, tiled, out_bf16, shape, params_by_ptr):
        """``src``: the tensor the gather reads (a parameter or a view of one)."""
        if self.ready or key in self.seen:
            return
        owner = params_by_ptr.get(src.untyped_storage().data_ptr())
        if owner is None:
            return                                 # a per-step temporary (e.g. the zero-padded gate weight): stays per layer
        base = (src.data_ptr() - owner.data_ptr()) // 4
        self.seen.add(key)
        self.log.append((key, owner, base, tuple(int(d) for d in dims), tuple(int(v) for v in strides), int(tiled),
                         int(out_bf16), tuple(shape)))

    def build(self, dev):
        if not self.log:
            return
        from ._lib import GrlPrepEntry
        n = len(self.log)
        arr = (GrlPrepEntry * n)()
        for i, (key, owner, base, dims, strides, tiled, out_bf16, shape) in enumerate(self.log):
            dst = torch.empty(shape, dtype=BF16 if out_bf16 else torch.float32, device=dev)
            self.cache[key] = dst
            e = arr[i]
            e.src, e.dst, e.base = owner.data_ptr(), dst.data_ptr(), base
            for d in range(4):
                e.dims[d], e.strides[d] = dims[d], strides[d]
            e.tiled, e.out_bf16 = tiled, out_bf16
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = raw.to(dev)                   # (one host->device copy, once per model and mode)
        self.count = n
        self.src = [(owner, owner.data_ptr()) for _, owner, *_ in self.log]
        self.ready = True

    def valid(self):
        return all(p.data_ptr() == q for p, q in self.src)

    def run(self):
        _call('grl_weight_prep', ptr(self.table), self.count)


WEIGHT_PREP = __import__('os').environ.get('GRL_WEIGHT_PREP', '1') != '0'      # 0: the per-layer path every step (A/B, tests)


def _weight_prep(model, b16):
    plans = model.__dict__.setdefault('_grl_weight_prep', {})
    wp = plans.get(b16)
    if wp is not None and wp.ready and not wp.valid():
        wp = None                                  # a parameter's storage moved: log again
    if wp is None:
        wp = plans[b16] = WeightPrep()
    return wp


class Tape(object):
    def __init__(self, dev):
        self.dev = dev
        self.side = None     # side stream closures are being recorded for (None: the launch stream)
        self.held = []       # tensors handed from one stream to the other: kept alive until the streams join
        self.wheld = []      # operands of weight-gradient launches in flight on the weight-gradient stream
        self.ops = _Ops(self)
        self.flat = None     # flat buffer of every parameter gradient of this tape
        self.cuts = {}       # section name -> (lo, hi) of `flat`: gradients final once the section's backward is done
        self._sent = 0       # elements of `flat` from the END already handed to the gradient sync
        self.g = {}          # id(forward tensor) -> gradient tensor
        self.pg = {}         # id(param) -> (param, grad)
        self.wc = {}         # packed weights for this step
        self.no_grad = set() # ids of tensors that need no gradient (network input)
        self.taps = None     # optional dict of intermediates (tests)
        self._pview = {}     # id(param) -> view into the flat gradient buffer
        self._owns = []      # (param, offset into `flat`)
        self.b16 = False     # bf16-storage step: the packed / transposed weights below are handed out as bf16 copies
        self.prep = None     # WeightPrep being recorded (first step of a model in this storage mode), or None
        self.params_by_ptr = None
        self.bnrec = {}         # id(activation of a conv_bn) -> _BNRec: what its BatchNorm backward needs (fused reduce)
        self.fuse_ok = set()    # ids of activations whose ONLY consumers are conv_bn ops (as input or residual)
        self.prep_event = None  # WeightPrep launched on the side stream: the launch stream joins it before layer 1
        self.stacks = []        # _WgradStack objects / stacked attention-MLP records of this tape: each must be drained
                                # (left == 0) or untouched when the backward ends, or a weight gradient was silently dropped
        self.foreign = {}       # storage data_ptr -> tensor: gradient buffers the tape does NOT own exclusively (handed in by
                                # autograd, or registered for more than one forward tensor): never masked / overwritten
                                # in place.  The tensors are held so that the address cannot be recycled within the step.

    # gradients of activations -------------------------------------------------
    def add_grad(self, t, g):
        cur = self.g.get(id(t))
        if cur is None:
            self.g[id(t)] = g                    # (adopted: every caller passes a buffer it allocated for this call)
        else:
            _call(_k('grl_axpby', cur), ptr(cur), ptr(g), ptr(cur), C.c_float(1.0), C.c_float(1.0), cur.numel())

    def add_masked(self, t, dy, act):
        """grad(t) += dy * (act > 0)"""
        cur = self.g.get(id(t))
        if cur is None:
            cur = _newl(tuple(dy.shape), dy)
            self.g[id(t)] = cur
            _call(_k('grl_relu_bwd', dy), ptr(dy), ptr(act), ptr(cur), dy.numel(), 0)
        else:
            _call(_k('grl_relu_bwd', dy), ptr(dy), ptr(act), ptr(cur), dy.numel(), 1)

    def full_grad(self, t):
        cur = self.g.get(id(t))
        if cur is None:
            cur = torch.zeros_like(t)
            self.g[id(t)] = cur
        return cur

    def take(self, t):
        return self.g.pop(id(t), None)

    def owns(self, g):
        """True if ``g`` is a gradient buffer a tape op allocated for exactly one forward tensor (so the op that
        popped it may overwrite it in place)."""
        # (storage identity, not data_ptr: a view of a foreign buffer at a non-zero offset is foreign too)
        return g.untyped_storage().data_ptr() not in self.foreign

    # gradients of parameters --------------------------------------------------
    def reserve_param_grads(self, params, cuts=None):
        """One zero-filled flat buffer for every parameter gradient of the step (one memset
        instead of one fill per parameter); ``pgrad`` hands out views, which autograd adopts as
        ``p.grad``.  ``cuts``: {section: index of its first parameter} -- in data-parallel training
        the slice from that parameter to the previously sent one is all-reduced as soon as the
        section's backward has run (``mark``)."""
        total = sum((p.numel() + 3) // 4 * 4 for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=self.dev)
        off, offs = 0, []
        for p in params:
            n = p.numel()
            self._pview[id(p)] = flat[off:off + n].view_as(p)
            offs.append(off)
            off += (n + 3) // 4 * 4
        self.flat = flat
        self._owns = list(zip(params, offs))     # declared to the gradient sync when the backward starts
        self.cuts = {name: offs[i] for name, i in (cuts or {}).items()}

    def mark(self, name):
        """Forward: called where section ``name`` BEGINS.  Backward (reverse replay): runs once
        everything recorded after it has run, i.e. when the gradients of the section's parameters
        -- and of every later section's -- are final: hand that tail of the flat buffer to the
        gradient sync, so its all-reduce overlaps the rest of the backward."""
        def done():
            self.flush(self.cuts.get(name), name)
        self.ops.append(done)

    def wgrad_join(self):
        """The launch stream waits for the weight-gradient stream (parameter gradients are about to be read)."""
        if self.wheld:
            check(_lib.load().grl_stream_wait_stream(_lib.stream(), _wgrad_stream(self.dev).cuda_stream), 'grl_stream_wait_stream')
            self.wheld = []

    def flush(self, lo=0, label='rest'):
        sync = _grad_sync[0]
        if sync is None or self.flat is None or lo is None:
            return                  # (single GPU: nothing consumes parameter gradients before the end of the backward,
        self.wgrad_join()           #  where Tape.backward joins the weight-gradient stream; no mid-backward joins)
        hi = self.flat.numel() - self._sent
        if lo < hi:
            sync.reduce(self.flat[lo:hi], label)
            self._sent = self.flat.numel() - lo

    def pgrad(self, p):
        e = self.pg.get(id(p))
        if e is None:
            v = self._pview.get(id(p))
            e = (p, v if v is not None else torch.zeros_like(p))
            self.pg[id(p)] = e
        return e[1]

    # packed weights -------------------------------------------------------------
    def _rec(self, key, src, dims, strides, tiled, out_bf16, shape):
        if self.prep is not None:
            self.prep.record(key, src, dims, strides, tiled, out_bf16, shape, self.params_by_ptr)

    def w_fwd(self, conv):
        """Forward GEMM weight [N][K] (3x3: tap-major pack)."""
        w = conv.weight
        key = ('f', id(w))
        if key not in self.wc:
            wd = w.detach()
            if wd.dim() == 4 and wd.shape[2] > 1:
                n, c, k, _ = wd.shape
                out = torch.empty(n, k * k * c, dtype=torch.float32, device=self.dev)
                wc = wd.contiguous()
                _call('grl_pack_conv_weight', ptr(wc), ptr(out), n, c, k, k)
                self.wc[key] = out
                self._rec(key, wd, (1, n, k * k, c), (0, c * k * k, 1, k * k), 0, self.b16, (n, k * k * c))      # out[n][t][c] = w[n][c][t]
            else:
                self.wc[key] = wd.contiguous().view(wd.shape[0], -1)
                if self.b16:
                    kk = self.wc[key].shape[1]
                    self._rec(key, wd, (1, 1, wd.shape[0], kk), (0, 0, kk, 1), 0, True, (wd.shape[0], kk))
            if self.b16:
                self.wc[key] = cast16(self.wc[key])
        return self.wc[key]

    def w16(self, w, key_obj=None):
        """bf16 copy of a dense fp32 weight for this step (identity when the step is fp32)."""
        if not self.b16:
            return w
        key = ('16', id(key_obj) if key_obj is not None else w.data_ptr(), tuple(w.shape))
        if key not in self.wc:
            self.wc[key] = cast16(w)
            if w.dim() == 2 and w.stride(1) == 1:
                self._rec(key, w, (1, 1, w.shape[0], w.shape[1]), (0, 0, w.stride(0), 1), 0, True, tuple(w.shape))
        return self.wc[key]

    def w_t(self, w2d, key_obj, ld=None, like=None):
        """Transposed [K][N] of a dense weight [N][K] (row stride ld) for the data gradient; ``like``: the gradient
        tensor it will multiply -- a bf16 one gets a bf16 copy (the per-clip heads stay fp32 in every mode)."""
        b16 = like is not None and like.dtype == BF16
        key = ('t', id(key_obj), w2d.data_ptr(), b16)
        if key not in self.wc:
            n, k = w2d.shape
            out = torch.empty(k, n, dtype=torch.float32, device=self.dev)
            _call('grl_transpose', ptr(w2d), ptr(out), n, k, ld or k)
            self.wc[key] = cast16(out) if b16 else out
            self._rec(key, w2d, (1, 1, k, n), (0, 0, 1, ld or k), 1, b16, (k, n))                     # out[k][n] = w[n][k]
        return self.wc[key]

    def w_dgrad(self, conv):
        w = conv.weight
        key = ('d', id(w))
        if key not in self.wc:
            wd = w.detach().contiguous()
            n, c, k, _ = wd.shape
            out = torch.empty(c, k * k * n, dtype=torch.float32, device=self.dev)
            _call('grl_pack_dgrad_weight', ptr(wd), ptr(out), n, c, k, k)
            self.wc[key] = cast16(out) if self.b16 else out
            taps = k * k                                                                               # out[c][taps-1-t][n] = w[n][c][t]
            self._rec(key, wd, (1, c, taps, n), (0, taps, -1, c * taps), 0, self.b16, (c, taps * n))
            if self.prep is not None and self.prep.log and self.prep.log[-1][0] == key:
                e = self.prep.log[-1]
                self.prep.log[-1] = e[:2] + (e[2] + taps - 1,) + e[3:]
        return self.wc[key]

    def w_dgrad_s2(self, conv, py, px):
        """Data-gradient weight of a 3x3 stride-2 conv for the input-pixel parity class (py, px):
        [cin][dy][dx][N] with the taps that reach that class -- ky = 1 for py = 0; ky = 2 (output row
        a) then ky = 0 (output row a+1) for py = 1; same in x.  A re-layout of 1/9 .. 4/9 of the
        weight, done with torch indexing once per step."""
        w = conv.weight
        key = ('d2', id(w), py, px)
        if key not in self.wc:
            kys = [1] if py == 0 else [2, 0]
            kxs = [1] if px == 0 else [2, 0]
            wd = w.detach()
            # (plain selects + stack, not list indexing: an index LIST becomes a host tensor and a host->device copy,
            # which a HIP-graph capture of the step cannot contain)
            wd = torch.stack([torch.stack([wd[:, :, ky, kx] for kx in kxs], 2) for ky in kys], 2)      # [N][cin][kh][kw]
            self.wc[key] = wd.permute(1, 2, 3, 0).contiguous().view(w.shape[1], -1)
            if self.b16:
                self.wc[key] = cast16(self.wc[key])
            n, c = w.shape[0], w.shape[1]                  # out[c][dy][dx][n] = w[n][c][ky(dy)][kx(dx)], ky = 1 | 2 - 2 dy
            self._rec(key, w.detach(), (c, len(kys), len(kxs), n), (9, -6 if py else 0, -2 if px else 0, c * 9), 0, self.b16,
                      (c, len(kys) * len(kxs) * n))
            if self.prep is not None and self.prep.log and self.prep.log[-1][0] == key:
                e = self.prep.log[-1]
                self.prep.log[-1] = e[:2] + (e[2] + (2 if py else 1) * 3 + (2 if px else 1),) + e[3:]
        return self.wc[key]

    def backward(self):
        sync = _grad_sync[0]
        if sync is not None and self.flat is not None:
            # ownership is declared HERE, from the tape's own offset table: the sync object only has to be
            # registered (GradSync.begin) before loss.backward(), not before the forward that built the tape
            for p, off in self._owns:
                sync.own(p, self.flat, off)
        fwd_math = _train_math[0]
        _train_math[0] = _TRAIN_MATH[_train_mode[0]][1]          # 'mixed': the backward GEMMs' datapath
        was, _in_backward[0] = _in_backward[0], True
        try:
            for fn in reversed(self.ops):
                fn()
        finally:
            _train_math[0] = fwd_math
            _in_backward[0] = was
        self.wgrad_join()
        self.ops = _Ops(self)
        # a stacked weight gradient is issued by the closure that fills the LAST block: a backward that skipped one of the
        # closures (a partial re-run, a closure that returned early) would drop dW of that layer without an error
        for st in self.stacks:
            left, steps = (st['left'], st['steps']) if isinstance(st, dict) else (st.left, st.steps)
            if left != 0 and left != steps:
                raise _lib.GrlHipError('Tape.backward: a stacked weight gradient was left with %d of %d blocks unfilled -- '
                                       'its dW was never issued' % (left, steps))
        self.stacks = []


STEM_WGRAD_FUSED = os.environ.get('GRL_STEM_WGRAD_FUSED', '1') != '0'   # A/B and tests only
BN_REDUCE_FUSED = os.environ.get('GRL_BN_REDUCE_FUSED', '1') != '0'     # A/B and tests only
BN_REDUCE_FUSED_BF16 = os.environ.get('GRL_BN_REDUCE_FUSED_BF16', '1') != '0'   # ... on bf16 storage (round 5)
PREP_ASYNC = os.environ.get('GRL_PREP_ASYNC', '1') != '0'               # A/B and tests only
RELU_BITS = os.environ.get('GRL_RELU_BITS', '1') != '0'                 # A/B and tests only
STEM_TAIL_FUSED = os.environ.get('GRL_STEM_TAIL_FUSED', '1') != '0'     # A/B and tests only


_NO_CONV = (0,) * 9
_wgrad_ws = {}


def wgrad(dz, x, dw, M, N, K, ldz=None, ldx=None, conv=None, k_out=0, accumulate=1, math=None, stream=None):
    """dw[N][K] (+)= dz^T . X  through grl_conv_wgrad_f32 (datapath: the training math mode).  ``stream``: a raw
    hipStream_t to launch on instead of torch's current stream (wgrad_async); returns the workspace tensor (the caller of
    a side-stream launch keeps it alive until the streams join)."""
    mth = _train_math[0] if math is None else math
    b16 = dz.dtype == BF16                  # bf16-storage operands (both): plain bf16 products, fp32 dW
    if b16 and x.dtype != BF16:
        raise _lib.GrlHipError('wgrad: dz is bf16 but x is %s' % x.dtype)
    cv = conv if conv is not None else _NO_CONV
    # one positional constructor call (field order of GrlWgrad: include/grl_hip.h) instead of ~20 attribute stores, and the
    # workspace size of a shape asked once: ~65 weight gradients per step on a step the host barely keeps ahead of
    d = GrlWgrad(ptr(dz), ptr(x), ptr(dw), None, M, N, K, ldz or N, ldx or K, k_out, accumulate, 1 if conv is not None else 0,
                 cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], cv[8], mth, 1 if b16 else 0)
    lib = _lib.load()
    key = (M, N, K, d.ldz, d.ldx, k_out, conv, mth, b16)
    nws = _wgrad_ws.get(key)
    if nws is None:
        nws = _wgrad_ws[key] = int(lib.grl_wgrad_workspace_floats(C.byref(d)))
    ws = torch.empty(nws, dtype=torch.float32, device=dz.device)
    d.workspace = ptr(ws)
    check(lib.grl_conv_wgrad_f32(C.byref(d), _lib.stream() if stream is None else stream), 'grl_conv_wgrad_f32')
    if engine._DEBUG_SYNC:
        engine._debug_sync('wgrad %s bf16-in %d conv %s' % ((M, N, K), d.in_bf16, conv))
    return ws


# A layer's weight gradient depends on nothing downstream of it and nothing waits for it before the optimizer (or the
# section's all-reduce): it is issued on its own HIP stream, so the MFMA-bound wgrad launches run next to the HBM-bound
# BatchNorm passes and under-filled data-gradient GEMMs of the following layers instead of in between them.  The
# launch stream waits for that stream where parameter gradients are consumed (Tape.flush, end of Tape.backward); the
# operands stay referenced until then.  GRL_WGRAD_STREAM=0 / WGRAD_STREAM = False: everything on the launch stream.
WGRAD_STREAM = __import__('os').environ.get('GRL_WGRAD_STREAM', '1') != '0'
WGRAD_HANDOFF_TORCH = __import__('os').environ.get('GRL_WGRAD_HANDOFF', 'c') == 'torch'       # A/B only
_wgrad_streams = {}


def _wgrad_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _wgrad_streams:
        _wgrad_streams[key] = torch.cuda.Stream(dev)
    return _wgrad_streams[key]


def wgrad_async(tp, dz, x, dw, M, N, K, **kw):
    """``wgrad`` on the weight-gradient stream (ordered after everything issued so far on the current one)."""
    if not WGRAD_STREAM:
        wgrad(dz, x, dw, M, N, K, **kw)
        return
    # Round 6: the hand-off is ONE C call (grl_stream_wait_stream: pooled event, record + wait) and the launch takes the
    # side stream's raw handle -- no torch.cuda.Event, no `with torch.cuda.stream(..)` (15-30 us of Python per weight
    # gradient, ~65 per step, on a step that is host-bound in bf16 storage).  The workspace is allocated under the CURRENT
    # stream and, like the operands, held until the streams join (Tape.wgrad_join): its block cannot be handed out again
    # before the launch stream has waited for the weight-gradient stream.
    if WGRAD_HANDOFF_TORCH:               # (A/B switch: the rounds 3-5 hand-off through torch objects)
        ws = _wgrad_stream(tp.dev)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(tp.dev))
        ws.wait_event(ev)
        with torch.cuda.stream(ws):
            wgrad(dz, x, dw, M, N, K, **kw)
        tp.wheld.extend((dz, x, dw))
        return
    wsh = _wgrad_stream(tp.dev).cuda_stream
    check(_lib.load().grl_stream_wait_stream(wsh, _lib.stream()), 'grl_stream_wait_stream')
    buf = wgrad(dz, x, dw, M, N, K, stream=wsh, **kw)
    tp.wheld.extend((dz, x, dw, buf))


def colsum_into(g, M, Ccols, out, ld=None):
    """out[c] += sum_m g[m][c]."""
    rows = _lib.load().grl_col_stats_rows(M)
    slab = _new((rows, 2, Ccols), g)
    _call(_k('grl_col_stats', g), ptr(g), ptr(slab), M, Ccols, ld or Ccols, None)
    _call('grl_slab_sum', ptr(slab), rows, 2 * Ccols, Ccols, ptr(out), 1)


class _BNState(object):
    __slots__ = ('mean', 'invstd', 'scale', 'shift', 'beta', 'buf')


def bn_apply(z, st, res, y, M, Cc, relu, bits=None):
    """y = relu?((z - mean) * gamma*invstd + beta + res) -- centred first, as torch's train kernel.
    ``bits``: uint8 tensor that receives the (y > 0) mask, one bit per output (RELU_BITS)."""
    _call(_k('grl_bn_apply_centered', z), ptr(z), ptr(st.mean), ptr(st.scale), ptr(st.beta), ptr(res), ptr(y), M, Cc,
          1 if relu else 0, ptr(bits))


def bn_finalize(slab, rows, Cc, count, bn, dev, gamma=None, beta=None, rm=None, rv=None, pivot=None, out=None):
    """``out``: four length-Cc vectors (mean, invstd, scale, shift) to write instead of a fresh buffer -- e.g. one
    half of a concatenated pair, so that no copy follows."""
    st = _BNState()
    buf = torch.empty((4, Cc), dtype=torch.float32, device=dev) if out is None else out
    # the four vectors as raw row addresses of `buf` (ptr() passes ints through): four tensor slices per BatchNorm were
    # ~10 us of host time, 89 times per step
    base, row = buf.data_ptr(), buf.stride(0) * 4
    st.buf = buf
    st.mean, st.invstd, st.scale, st.shift = base, base + row, base + 2 * row, base + 3 * row
    gamma = bn.weight if gamma is None else gamma
    beta = bn.bias if beta is None else beta
    rm = bn.running_mean if rm is None else rm
    rv = bn.running_var if rv is None else rv
    st.beta = beta
    # num_batches_tracked (torch's int64 counter) is bumped by the same launch
    _call('grl_bn_stats_finalize', ptr(slab), rows, Cc, count, ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
          ptr(bn.num_batches_tracked), C.c_float(bn.momentum), C.c_float(bn.eps), ptr(st.mean), ptr(st.invstd),
          ptr(st.scale), ptr(st.shift), ptr(pivot))
    return st


BN_FINAPPLY = True       # the library's switch decides (GRL_BN_FINAPPLY=1 / set_bn_finapply: off by default -- measured slower)


_finapply_state = [None]


def _finapply_on():
    """the library's switch (GRL_BN_FINAPPLY / grl_bn_finalize_apply_mode), read once; set_bn_finapply() changes both sides"""
    if _finapply_state[0] is None:
        _finapply_state[0] = bool(_lib.load().grl_bn_finalize_apply_mode(-1))
    return _finapply_state[0]


def set_bn_finapply(on):
    """tests / A-B: switch the fused BatchNorm finalize + apply form (forward here, backward inside the library); returns
    the previous setting"""
    was = bool(_lib.load().grl_bn_finalize_apply_mode(1 if on else 0))
    _finapply_state[0] = bool(on)
    return was


def bn_finalize_apply(slab, rows, Cc, count, bn, dev, z, res, y, relu, bits=None):
    """bn_finalize + bn_apply as ONE launch (grl_bn_finalize_apply: every workgroup of the apply pass reduces the slab columns
    of its own 64 channels) -- for rows <= 64 slab rows and Cc % 64 == 0, bit-identical to the two calls."""
    st = _BNState()
    buf = torch.empty((4, Cc), dtype=torch.float32, device=dev)
    base, row = buf.data_ptr(), Cc * 4
    st.buf = buf
    st.mean, st.invstd, st.scale, st.shift = base, base + row, base + 2 * row, base + 3 * row
    st.beta = bn.bias
    _call(_k('grl_bn_finalize_apply', z), ptr(slab), rows, Cc, count, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
          ptr(bn.running_var), ptr(bn.num_batches_tracked), C.c_float(bn.momentum), C.c_float(bn.eps), st.mean, st.invstd,
          st.scale, st.shift, None, ptr(z), ptr(res), ptr(y), count, 1 if relu else 0, ptr(bits))
    return st


def bn_backward(dy, z, act, st, gamma, dgamma, dbeta, M, Cc, gres=None, gres_acc=0, mask_from_z=False, bits=None,
                out=None):
    """``gres``: gradient buffer of the residual input (gets / accumulates the masked dy in the
    same pass that writes dz).  ``mask_from_z``: y = relu(bn(z)) without a residual -- the ReLU mask is recomputed
    from z with the forward's own operations (st.scale, st.beta) instead of reading the activation.  ``out``: where dz
    goes (a row block of a _WgradStack) instead of a fresh buffer."""
    dz = _newl((M, Cc), dy) if out is None else out
    rows = _lib.load().grl_col_stats_rows(M)
    slab = _new((rows, 2, Cc), dy)
    coef = _new((2, Cc), dy)
    _call(_k('grl_bn_bwd', dy), ptr(dy), ptr(z), None if (mask_from_z or bits is not None) else ptr(act), ptr(st.mean),
          ptr(st.invstd), ptr(gamma), ptr(dz), ptr(dgamma), ptr(dbeta), ptr(slab), ptr(coef), M, Cc, ptr(gres), gres_acc,
          ptr(st.scale) if mask_from_z else None, ptr(st.beta) if mask_from_z else None, ptr(bits))
    return dz


# ----------------------------------------------------------------------------
# ops (forward now, backward closure on the tape)
# ----------------------------------------------------------------------------
class _WgradStack(object):
    """Weight-gradient operands of a layer that one step applies ``steps`` times with the SAME weights (the TRL
    recurrences, grl_model.py:186-205): step i's input and its output gradient are row block i of two buffers, and dW
    is ONE product over all steps * M rows -- issued by the backward closure that fills the last block -- instead of
    ``steps`` products over M = B * 128 rows each (at 32 clips a quarter of the pixel range the weight-gradient kernels
    reach their rate on: 4096 x 2048 x 512 ran at 100 TFLOP/s, 16384 x 2048 x 512 runs at 123) and ``steps`` slab
    reductions.  ``extra``: input blocks past the last step (the recurrence's final output lives in the same buffer)."""

    def __init__(self, steps, rows_in, K, like, extra=0):
        self.steps, self.rows_in = steps, rows_in
        self.x = _newl(((steps + extra) * rows_in, K), like)
        self.dz, self.left = None, steps

    def x_block(self, i):
        return self.x[i * self.rows_in:(i + 1) * self.rows_in]

    def dz_block(self, i, M, N, like):
        if self.dz is None:
            self.dz = _newl((self.steps * M, N), like)
        return self.dz[i * M:(i + 1) * M]

    def filled(self):
        """One more block of dz is written; True when that was the last one."""
        self.left -= 1
        return self.left == 0

    def inputs(self):
        return self.x[:self.steps * self.rows_in]


ADOPT_STATS = [0, 0]       # residual-branch gradients adopted in place / copied (Tape.owns said 'foreign'); tests read it
WGRAD_STACK = os.environ.get('GRL_WGRAD_STACK', '1') != '0'             # A/B and tests only


class _BNRec(object):
    """What the data-gradient GEMM that completes grad(a), a = relu?(bn(z) (+res)), needs to run that BatchNorm's
    backward reduce in its epilogue (GrlGemm.bn_z): ``uses`` counts the contributions still to come."""
    __slots__ = ('z', 'st', 'bits', 'from_z', 'uses', 'slab')

    def args(self):
        return (self.z, self.st.mean, self.st.invstd, self.st.scale if self.from_z else None,
                self.st.beta if self.from_z else None, self.bits)


def _bn_use(tp, t, n=1):
    r = tp.bnrec.get(id(t)) if t is not None else None
    if r is not None:
        r.uses += n
    return r


def conv_bn(tp, x, n_img, H, W, conv, bn, relu, res=None, gbias=None, rpg=0, kcols=None, ldw=None, out=None, wstack=None):
    """Conv (1x1 / 3x3, stride 1 / 2) + train-mode BN (+residual) (+ReLU).
    x: channels-last [n_img*H*W][cin].  ``kcols``/``ldw``: use only the first kcols input
    channels of a wider 1x1 weight (GCE split weight).  ``out``: the activation's buffer (a block of the NEXT layer's
    _WgradStack); ``wstack`` = (stack, i): x IS block i of the stack and this layer's dz goes to the stack's block i --
    the weight gradient is one product over the whole stack."""
    w = conv.weight
    N, k = w.shape[0], (w.shape[2] if w.dim() == 4 else 1)
    stride = conv.stride[0] if hasattr(conv, 'stride') else 1
    cin = kcols or w.shape[1]
    wf = tp.w_fwd(conv)
    K = k * k * cin
    if k == 1 and stride == 1:
        Ho, Wo, geom = H, W, None
    else:
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        geom = (H, W, cin, Ho, Wo, k, k, stride, pad)
    M = n_img * Ho * Wo
    z = _newl((M, N), x)
    _, slab = gemm(x, wf, z, M, N, K, ldw=ldw or wf.shape[1], gbias=gbias, rows_per_group=rpg,
                   stats=True, conv=geom)
    a = _newl((M, N), x) if out is None else out
    # y = relu(bn(z) + res): the backward needs the mask (y > 0); recorded as one bit per output here, it is read
    # back instead of the whole activation (1/16 of the bytes of the widest BatchNorms of the step)
    bits = None
    if relu and res is not None and RELU_BITS:
        bits = torch.empty(M * N // (8 if a.dtype == BF16 else 4), dtype=torch.uint8, device=tp.dev)
    if BN_FINAPPLY and slab.shape[0] <= 64 and N % 64 == 0 and _finapply_on():
        # small slab (the TRL memo bottleneck: M = B * 128 pixel rows): the finalize runs inside the apply launch
        st = bn_finalize_apply(slab, slab.shape[0], N, M, bn, tp.dev, z, res, a, relu, bits=bits)
    else:
        st = bn_finalize(slab, slab.shape[0], N, M, bn, tp.dev)
        bn_apply(z, st, res, a, M, N, relu, bits=bits)
    # (fused BatchNorm-backward reduce, GrlGemm.bn_z: this op is one consumer of x and of res, and `a` gets a record
    # that the data-gradient GEMM completing grad(a) can use)
    _bn_use(tp, x)
    _bn_use(tp, res)
    rec = None
    # (bf16 storage, round 5: the 128-row tile family carries the reduce in its interior epilogue -- every tile of the
    #  data-gradient GEMM that completes grad(a) must be one: M % 128 == 0, N % 128 == 0 or N == 64)
    bn16_ok = BN_REDUCE_FUSED_BF16 and M % 128 == 0 and (N % 128 == 0 or N == 64)
    if relu and (res is None or bits is not None) and (a.dtype != BF16 or bn16_ok):
        rec = _BNRec()
        rec.z, rec.st, rec.bits, rec.from_z, rec.uses, rec.slab = z, st, bits, res is None, 0, None
        tp.bnrec[id(a)] = rec

    def bwd():
        da = tp.take(a)
        if da is None:
            if wstack is not None:          # no gradient reaches this step: its block of the stacked product is zero
                wstack[0].dz_block(wstack[1], M, N, a).zero_()
                if wstack[0].filled():
                    wgrad_async(tp, wstack[0].dz, wstack[0].inputs(), tp.pgrad(conv.weight), wstack[0].steps * M, N, K,
                                conv=geom)
            return
        act = a if relu else None
        gres, gacc = None, 0
        if res is not None:                 # grad(res) (+)= da * (a > 0)
            gres = tp.g.get(id(res))
            gacc = 1 if gres is not None else 0
            if gres is None:
                # first contribution to grad(res): it IS the masked da -- the reduce pass of grl_bn_bwd masks da in
                # place (gres == dy) and this tape entry adopts the buffer (da was popped: nobody else reads it)
                adopt = relu and tp.owns(da)
                ADOPT_STATS[0 if adopt else 1] += 1
                gres = tp.g[id(res)] = da if adopt else _newl((M, N), da)
            _bn_use(tp, res, -1)
        if rec is not None and rec.slab is not None:
            # the GEMM that completed grad(a) masked it and left the two column sums (rec.slab): finalize + apply only
            dz = _newl((M, N), da) if wstack is None else wstack[0].dz_block(wstack[1], M, N, da)
            coef = _new((2, N), da)
            _call(_k('grl_bn_bwd_finish', da), ptr(da), ptr(z), ptr(st.mean), ptr(st.invstd), ptr(bn.weight), ptr(dz),
                  ptr(tp.pgrad(bn.weight)), ptr(tp.pgrad(bn.bias)), ptr(rec.slab), rec.slab.shape[0], ptr(coef), M, N,
                  ptr(gres), gacc)
            rec.slab = None
        else:
            dz = bn_backward(da, z, act, st, bn.weight, tp.pgrad(bn.weight), tp.pgrad(bn.bias), M, N,
                             gres=gres, gres_acc=gacc, mask_from_z=relu and res is None, bits=bits,
                             out=None if wstack is None else wstack[0].dz_block(wstack[1], M, N, da))
        conv_param_and_input_grads(tp, dz, x, conv, n_img, H, W, Ho, Wo, cin, N, k, stride, geom,
                                   kcols=kcols, ldw=ldw, wstack=wstack)
        if gbias is not None:
            tp.g[('gbias', id(z))] = dz          # the caller's closure reduces it per clip
    tp.ops.append(bwd)
    return a, Ho, Wo, z


def conv_param_and_input_grads(tp, dz, x, conv, n_img, H, W, Ho, Wo, cin, N, k, stride, geom,
                               kcols=None, ldw=None, wstack=None):
    """dW += dz^T . X (gathered), dx = dz . W (data gradient)."""
    w = conv.weight
    M = n_img * Ho * Wo
    K = k * k * cin
    if wstack is not None:
        stk = wstack[0]                     # (images are independent: the stacked steps are just more of them)
        if stk.filled():
            wgrad_async(tp, stk.dz, stk.inputs(), tp.pgrad(w), stk.steps * M, N, K, conv=geom)
    elif kcols is None:
        wgrad_async(tp, dz, x, tp.pgrad(w), M, N, K, conv=geom)
    else:                                   # first kcols columns of a wider weight
        tmp = torch.empty(N, kcols, dtype=torch.float32, device=tp.dev)
        wgrad(dz, x, tmp, M, N, K, accumulate=0)
        tp.pgrad(w).view(N, -1)[:, :kcols].add_(tmp)        # (`view[...] += t` would copy the slice onto itself afterwards)
    if id(x) in tp.no_grad:
        return
    Min = n_img * H * W
    # if x already has a gradient (residual joins), accumulate in the GEMM epilogue
    # (res = y = that tensor: every element is read and written by the same lane)
    cur = tp.g.get(id(x))
    dx = cur if cur is not None else _newl((Min, cin), dz)
    # this data gradient completes grad(x) and x = relu(bn(z') (+res')) of a conv_bn whose consumers are all known: run
    # that BatchNorm's backward reduce in the GEMM epilogue (GrlGemm.bn_z) instead of re-reading the gradient
    rec = tp.bnrec.get(id(x)) if (BN_REDUCE_FUSED and id(x) in tp.fuse_ok) else None
    fuse = (rec is not None and rec.uses == 1 and stride == 1 and kcols is None)
    if rec is not None:
        rec.uses -= 1
    if k == 1:
        w2d = w.detach().view(N, -1)
        wt = tp.w_t(w2d[:, :cin] if kcols else w2d, w, ld=w2d.shape[1], like=dz)
        if fuse:
            _, rec.slab = gemm(dz, wt, dx, Min, cin, N, res=cur, stats=True, bn=rec.args())
        elif stride != 1:
            # a 1x1 stride-2 conv only reads the even pixels: its data gradient is a GEMM at OUTPUT
            # resolution scattered to them (a quarter of the zero-stuffed GEMM's FLOPs)
            small = _newl((M, cin), dz)
            gemm(dz, wt, small, M, cin, N)
            _call(_k('grl_dilate2', small), ptr(small), ptr(dx), n_img, Ho, Wo, H, W, cin, 1 if cur is not None else 0, 0, 0)
        else:
            gemm(dz, wt, dx, Min, cin, N, res=cur)
    elif stride == 1:
        if fuse:
            _, rec.slab = gemm(dz, tp.w_dgrad(conv), dx, Min, cin, k * k * N, conv=(H, W, N, H, W, k, k, 1, k // 2), res=cur,
                               stats=True, bn=rec.args())
        else:
            gemm(dz, tp.w_dgrad(conv), dx, Min, cin, k * k * N, conv=(H, W, N, H, W, k, k, 1, k // 2), res=cur)
    elif k == 3 and stride == 2 and H == 2 * Ho and W == 2 * Wo:
        # 3x3 stride-2 pad-1: input pixel (2a+py, 2b+px) only meets the taps ky = 1 (py = 0) or
        # ky = 2, 0 (py = 1; output rows a, a+1), likewise in x -- four small stride-1 convolutions
        # over dz at OUTPUT resolution (1, 2, 2 and 4 taps), each scattered to its parity class:
        # the forward FLOPs instead of the 4x of a zero-stuffed full-resolution convolution.
        for py in (0, 1):
            for px in (0, 1):
                wc = tp.w_dgrad_s2(conv, py, px)                  # [cin][taps][N]
                kh, kw = 1 + py, 1 + px
                small = _newl((M, cin), dz)
                gemm(dz, wc, small, M, cin, kh * kw * N, conv=(Ho, Wo, N, Ho, Wo, kh, kw, 1, 0))
                _call(_k('grl_dilate2', small), ptr(small), ptr(dx), n_img, Ho, Wo, H, W, cin,
                      1 if cur is not None else 2, py, px)
    else:
        src = _newl((Min, N), dz)
        _call(_k('grl_dilate2', dz), ptr(dz), ptr(src), n_img, Ho, Wo, H, W, N, 0, 0, 0)
        gemm(src, tp.w_dgrad(conv), dx, Min, cin, k * k * N, conv=(H, W, N, H, W, k, k, 1, k // 2), res=cur)
    if cur is None:
        tp.g[id(x)] = dx


def biased_conv_relu(tp, x, M, conv, wstack=None):
    """1x1 conv with bias + ReLU (TRL f1/f2, grl_model.py:95-121).  ``wstack`` = (stack, i): as in conv_bn (the bias
    gradient is one column sum over the whole stack too)."""
    w = conv.weight
    N, K = w.shape[0], w.shape[1]
    w2d = w.detach().view(N, K)
    a = _newl((M, N), x)
    gemm(x, tp.w16(w2d, w), a, M, N, K, shift=conv.bias.detach(), relu=True)

    def stack_done():
        stk = wstack[0]
        if stk.filled():
            colsum_into(stk.dz, stk.steps * M, N, tp.pgrad(conv.bias))
            wgrad_async(tp, stk.dz, stk.inputs(), tp.pgrad(w), stk.steps * M, N, K)

    def bwd():
        da = tp.take(a)
        if da is None:
            if wstack is not None:          # no gradient reaches this step: its block of the stacked product is zero
                wstack[0].dz_block(wstack[1], M, N, a).zero_()
                stack_done()
            return
        g = _newl((M, N), da) if wstack is None else wstack[0].dz_block(wstack[1], M, N, da)
        _call(_k('grl_relu_bwd', da), ptr(da), ptr(a), ptr(g), da.numel(), 0)
        if wstack is None:
            colsum_into(g, M, N, tp.pgrad(conv.bias))
            wgrad_async(tp, g, x, tp.pgrad(w), M, N, K)
        else:
            stack_done()
        cur = tp.g.get(id(x))
        if cur is not None and tuple(cur.shape) != (M, K):
            cur = cur.view(M, K)
        dx = cur if cur is not None else _newl((M, K), g)
        gemm(g, tp.w_t(w2d, w, like=g), dx, M, K, N, res=cur)
        if cur is None:
            tp.g[id(x)] = dx
    tp.ops.append(bwd)
    return a


def linear_bn_relu(tp, x, M, lin, bn):
    """Linear(+bias) + BN1d(train) + ReLU on [M][K] rows (basebranch.py:38-40)."""
    w = lin.weight
    N, K = w.shape
    z = _new((M, N), x)
    gemm(x, w.detach(), z, M, N, K, shift=lin.bias.detach())
    rows = _lib.load().grl_col_stats_rows(M)
    slab = _new((rows, 2, N), x)
    _call('grl_col_stats', ptr(z), ptr(slab), M, N, N, ptr(z))           # pivot = row 0 of z
    st = bn_finalize(slab, rows, N, M, bn, tp.dev, pivot=z)
    a = _new((M, N), x)
    bn_apply(z, st, None, a, M, N, True)

    def bwd():
        da = tp.take(a)
        if da is None:
            return
        dz = bn_backward(da, z, a, st, bn.weight, tp.pgrad(bn.weight), tp.pgrad(bn.bias), M, N)
        colsum_into(dz, M, N, tp.pgrad(lin.bias))
        wgrad(dz, x, tp.pgrad(w), M, N, K)
        dx = _new((M, K), dz)
        gemm(dz, tp.w_t(w.detach(), w), dx, M, K, N)
        tp.add_grad(x, dx)
    tp.ops.append(bwd)
    return a


def group_mean_op(tp, x, groups, rows, Cc):
    y = _new((groups, Cc), x)
    _call(_k('grl_group_mean', x), ptr(x), ptr(y), groups, rows, Cc, Cc, C.c_float(1.0), 0)

    def bwd():
        dy = tp.take(y)
        if dy is None:
            return
        cur = tp.g.get(id(x))
        if cur is None:
            cur = _newl(tuple(x.shape), x)
            tp.g[id(x)] = cur
            acc = 0
        else:
            acc = 1
        add_rowbcast(cur, dy, groups * rows, Cc, rows, 1.0 / rows, acc)
    tp.ops.append(bwd)
    return y


def bn1d_l2norm(tp, f, rows, Cc, bn):
    """BatchNorm1d(train) + F.normalize on [rows][C] (grl_model.py:222-226)."""
    nr = _lib.load().grl_col_stats_rows(rows)
    slab = _new((nr, 2, Cc), f)
    _call('grl_col_stats', ptr(f), ptr(slab), rows, Cc, Cc, ptr(f))
    st = bn_finalize(slab, nr, Cc, rows, bn, tp.dev, pivot=f)
    y = _new((rows, Cc), f)
    bn_apply(f, st, None, y, rows, Cc, False)
    out = _new((rows, Cc), f)
    _call('grl_affine_l2norm', ptr(y), None, None, ptr(out), rows, Cc, Cc)

    def bwd():
        dout = tp.take(out)
        if dout is None:
            return
        dy = _new((rows, Cc), f)
        _call('grl_l2norm_bwd', ptr(dout), Cc, ptr(out), Cc, ptr(y), ptr(dy), rows, Cc)
        df = bn_backward(dy, f, None, st, bn.weight, tp.pgrad(bn.weight), tp.pgrad(bn.bias), rows, Cc)
        tp.add_grad(f, df)
    tp.ops.append(bwd)
    return out


# ----------------------------------------------------------------------------
# model sections
# ----------------------------------------------------------------------------
def trunk_train(tp, model, x):
    base = model.backbone.base
    n, _, H0, W0 = x.shape
    H, W = H0, W0
    tp.no_grad.add(id(x))
    tp.mark('stem')
    Hs, Ws = H // 2, W // 2
    M0 = n * Hs * Ws
    conv1, bn1 = base[0], base[1]
    ones, zeros = torch.ones(64, device=tp.dev), torch.zeros(64, device=tp.dev)
    adt = BF16 if tp.b16 else torch.float32              # storage type of the activations from here on
    z0 = torch.empty((M0, 64), dtype=adt, device=tp.dev)
    w0 = conv1.weight.detach().contiguous()
    # the stem kernel stages its weights in LDS per workgroup: from the packed image that is a 16-byte copy, from the
    # raw [64][147] tensor a scalar gather with an integer division per element (0.35 instead of 0.25 ms per 32 x 4 step)
    if tp.b16:
        wp0 = torch.empty(64 * 184, dtype=BF16, device=tp.dev)
        _call('grl_stem_pack_weight_bf16', ptr(w0), ptr(wp0))
    else:
        wp0 = torch.empty(64 * 164, dtype=torch.float32, device=tp.dev)
        _call('grl_stem_pack_weight', ptr(w0), ptr(wp0))
    _call(_k('grl_stem_conv7x7', z0), ptr(x), ptr(w0), ptr(ones), ptr(zeros), ptr(z0), n, H, W, 0, ptr(wp0))
    rows = _lib.load().grl_col_stats_rows(M0)
    slab = _new((rows, 2, 64), x)
    pivot = z0[0].float().contiguous() if tp.b16 else z0           # (the bf16 kernel takes the pivot as an fp32 vector)
    _call(_k('grl_col_stats', z0), ptr(z0), ptr(slab), M0, 64, 64, ptr(pivot))
    st = bn_finalize(slab, rows, 64, M0, bn1, tp.dev, pivot=pivot)
    Hp, Wp = (Hs + 1) // 2, (Ws + 1) // 2
    p0 = _newl((n * Hp * Wp, 64), z0)
    fused_tail = STEM_TAIL_FUSED and tp.taps is None
    if fused_tail:
        # BatchNorm apply + ReLU + max-pool in one pass with the windows' first-maximum positions: the post-ReLU map
        # (268 MB per 32 x 4 step) is never written; the backward routes dp through the positions and recomputes the
        # ReLU mask from z0 (bit-identical to the separate passes, tested)
        a0 = None
        pidx = torch.empty((n * Hp * Wp, 64), dtype=torch.uint8, device=tp.dev)
        _call(_k('grl_bn_relu_maxpool3x3s2', z0), ptr(z0), ptr(st.mean), ptr(st.scale), ptr(st.beta), ptr(p0), ptr(pidx),
              n, Hs, Ws, 64)
    else:
        a0 = _newl((M0, 64), z0)
        bn_apply(z0, st, None, a0, M0, 64, True)
        _call(_k('grl_maxpool3x3s2', a0), ptr(a0), ptr(p0), n, Hs, Ws, 64)

    def bwd_stem():
        dp = tp.take(p0)
        if dp is None:
            return
        da = _newl((M0, 64), dp)
        if fused_tail:
            _call(_k('grl_maxpool3x3s2_bwd_idx', dp), ptr(pidx), ptr(dp), ptr(da), n, Hs, Ws, 64)
            dz = bn_backward(da, z0, None, st, bn1.weight, tp.pgrad(bn1.weight), tp.pgrad(bn1.bias), M0, 64,
                             mask_from_z=True)
        else:
            _call(_k('grl_maxpool3x3s2_bwd', dp), ptr(a0), ptr(dp), ptr(da), n, Hs, Ws, 64)
            dz = bn_backward(da, z0, a0, st, bn1.weight, tp.pgrad(bn1.weight), tp.pgrad(bn1.bias), M0, 64)
        if STEM_WGRAD_FUSED:          # straight from the NCHW clip: no 671 MB im2col matrix (csrc/train.hip)
            ws = torch.empty(_lib.load().grl_stem_wgrad_workspace_floats(n, H0, W0), dtype=torch.float32, device=tp.dev)
            _call('grl_stem_wgrad', ptr(x), ptr(dz), 1 if dz.dtype == BF16 else 0, ptr(tp.pgrad(conv1.weight)), ptr(ws),
                  n, H0, W0, 1)                                               # H, W are rebound below
        else:
            col = _newl((M0, 160), dp)
            _call(_k('grl_stem_im2col', dp), ptr(x), ptr(col), n, H0, W0, 160)
            wgrad(dz, col, tp.pgrad(conv1.weight), M0, 64, 160, k_out=147)
    tp.ops.append(bwd_stem)

    if tp.taps is not None:
        tp.taps['stem'] = engine._to_nchw(a0, n, Hs, Ws)
        tp.taps['pool'] = engine._to_nchw(p0, n, Hp, Wp)
    cur, H, W = p0, Hp, Wp
    if tp.prep_event is not None:                        # the step's weight table, launched next to the stem
        torch.cuda.current_stream(tp.dev).wait_event(tp.prep_event)
        tp.prep_event = None
    for li in (4, 5, 6, 7):
        tp.mark('layer%d' % (li - 3))
        for blk in base[li]:
            o1, _, _, _ = conv_bn(tp, cur, n, H, W, blk.conv1, blk.bn1, True)
            o2, Ho, Wo, _ = conv_bn(tp, o1, n, H, W, blk.conv2, blk.bn2, True)
            if blk.downsample is not None:
                res, _, _, _ = conv_bn(tp, cur, n, H, W, blk.downsample[0], blk.downsample[1], False)
            else:
                res = cur
            cur, _, _, _ = conv_bn(tp, o2, n, Ho, Wo, blk.conv3, blk.bn3, True, res=res)
            # every consumer of o1, o2 and of a block's output is a conv_bn of this loop (the LAST block's output also
            # feeds GCE: not marked) -- the premise of the fused BatchNorm-backward reduce
            tp.fuse_ok.update((id(o1), id(o2)))
            if not (li == 7 and blk is base[li][-1]):
                tp.fuse_ok.add(id(cur))
            H, W = Ho, Wo
        if tp.taps is not None:
            tp.taps['layer%d' % (li - 3)] = engine._to_nchw(cur, n, H, W)
    return cur


def _pad32(vec, fill=0.0, n=32):
    out = torch.full((n,), fill, dtype=torch.float32, device=vec.device)
    out[:vec.numel()] = vec.detach().reshape(-1)
    return out


def gce_train(tp, model, x4, b, t):
    bb = model.backbone
    M = x4.shape[0]
    rpc = t * PIX
    x_glo = group_mean_op(tp, x4, b, rpc, 2048)
    glo = linear_bn_relu(tp, x_glo, b, bb.glo_fc[0], bb.glo_fc[1])
    conv0, bn0 = bb.corr_atte[0], bb.corr_atte[1]
    w0 = conv0.weight.detach().view(1024, 3072)
    gb = _new((b, 1024), x4)
    gemm(glo, w0[:, 2048:], gb, b, 1024, 1024, ldw=3072)
    h1, _, _, z1 = conv_bn(tp, x4, b * t, 16, 8, conv0, bn0, False, gbias=gb, rpg=rpc, kcols=2048, ldw=3072)

    def bwd_gbias():
        dz1 = tp.g.pop(('gbias', id(z1)), None)
        if dz1 is None:
            return
        dgb = _new((b, 1024), x4)
        _call(_k('grl_group_mean', dz1), ptr(dz1), ptr(dgb), b, rpc, 1024, 1024, C.c_float(float(rpc)), 0)
        tmp = torch.empty(1024, 1024, dtype=torch.float32, device=tp.dev)
        wgrad(dgb, glo, tmp, b, 1024, 1024, accumulate=0)
        tp.pgrad(conv0.weight).view(1024, 3072)[:, 2048:].add_(tmp)
        dglo = _new((b, 1024), x4)
        gemm(dgb, tp.w_t(w0[:, 2048:], ('wg', id(conv0.weight)), ld=3072), dglo, b, 1024, 1024)
        tp.add_grad(glo, dglo)
    # h1's closure (already on the tape) must run BEFORE this one in backward, and the
    # closures of glo's producers after it: insert right below the h1 closure.
    tp.ops.insert(len(tp.ops) - 1, bwd_gbias)

    h2, _, _, _ = conv_bn(tp, h1, b * t, 16, 8, bb.corr_atte[2], bb.corr_atte[3], True)
    # 256 -> 1 conv + BN(1): run as a PW-wide zero-padded channel block (K of the data-gradient GEMM must be a
    # multiple of the K stage: 32 fp32, 64 in the bf16-storage datapath); only column 0 is real.
    conv5, bn6 = bb.corr_atte[5], bb.corr_atte[6]
    PW = 64 if tp.b16 else 32
    w5 = torch.zeros(PW, 256, dtype=torch.float32, device=tp.dev)
    w5[0] = conv5.weight.detach().view(256)
    z3 = _newl((M, PW), x4)
    _, slab = gemm(h2, tp.w16(w5), z3, M, PW, 256, stats=True)
    # gamma | beta | running mean | running var of the ONE real channel in column 0 of a zero (4, PW) block (three
    # launches instead of the eight of four padded vectors; the padding channels' statistics are never read)
    pad4 = torch.zeros(4, PW, dtype=torch.float32, device=tp.dev)
    pad4[:, 0] = torch.cat((bn6.weight.detach(), bn6.bias.detach(), bn6.running_mean, bn6.running_var))
    g32, b32, rm32, rv32 = pad4[0], pad4[1], pad4[2], pad4[3]
    st = bn_finalize(slab, slab.shape[0], PW, M, bn6, tp.dev, gamma=g32, beta=b32, rm=rm32, rv=rv32)
    torch._foreach_copy_([bn6.running_mean, bn6.running_var], [rm32[:1], rv32[:1]])
    y3 = _newl((M, PW), x4)
    bn_apply(z3, st, None, y3, M, PW, False)
    cmap = _new((M,), x4)
    xc, xu = _newl((M, 2048), x4), _newl((M, 2048), x4)
    _call(_k('grl_gate_apply', x4), ptr(y3), PW, ptr(x4), ptr(cmap), ptr(xc), ptr(xu), M, 2048)

    def bwd_gate():
        dxc, dxu = tp.take(xc), tp.take(xu)
        if dxc is None and dxu is None:
            return
        dxc = dxc if dxc is not None else torch.zeros_like(xc)
        dxu = dxu if dxu is not None else torch.zeros_like(xu)
        dy3 = torch.zeros((M, PW), dtype=x4.dtype, device=tp.dev)
        cur = tp.g.get(id(x4))
        if cur is None:
            cur = _newl((M, 2048), x4)
            tp.g[id(x4)] = cur
            acc = 0
        else:
            acc = 1
        _call(_k('grl_gate_bwd', x4), ptr(dxc), ptr(dxu), ptr(x4), ptr(cmap), ptr(cur), acc, ptr(dy3), PW, M, 2048)
        dgb = torch.zeros(2, PW, device=tp.dev)
        dg32, db32 = dgb[0], dgb[1]
        dz3 = bn_backward(dy3, z3, None, st, g32, dg32, db32, M, PW)
        dw5 = torch.empty(PW, 256, dtype=torch.float32, device=tp.dev)
        wgrad(dz3, h2, dw5, M, PW, 256, accumulate=0)
        torch._foreach_add_([tp.pgrad(bn6.weight), tp.pgrad(bn6.bias), tp.pgrad(conv5.weight).view(1, 256)],
                            [dg32[:1], db32[:1], dw5[:1]])
        dh2 = _newl((M, 256), x4)
        gemm(dz3, tp.w_t(w5, ('w5', id(conv5.weight)), like=dz3), dh2, M, 256, PW)
        tp.add_grad(h2, dh2)
    tp.ops.append(bwd_gate)
    if tp.taps is not None:
        tp.taps.update(x_glo=x_glo, glo=glo, corr_map=cmap.view(b * t, 1, 16, 8))
    return xu, xc, cmap


def _axpy_frame(dst_full, ti, src, b, t, frame, alpha=1.0):
    """dst_full[b][ti] += alpha * src[b]  (frame-sized rows)."""
    _call(_k('grl_axpy_strided', dst_full), ptr(dst_full.view(-1)[ti * frame:]), t * frame, ptr(src), frame, b, frame,
          C.c_float(alpha), 1)


def trl_train(tp, model, xu, xc, b, t):
    """TRL forward (grl_model.py:170-208) recorded for the backward.  The two directions are independent
    recurrences over their own parameters; like the eval path (engine._TrlFork) they are issued on two HIP
    streams -- forward and, through the tape's stream tags, backward -- so one direction's small M = B*128
    GEMMs and latency-bound BatchNorm / attention kernels run next to the other's.  Everything the two
    directions ACCUMULATE into (f_corr, and in the backward the gradients of x_uncorr, GAP(x_corr) and the
    initial memo) is kept per direction and summed on the launch stream at the join."""
    trl = model.temporal_learning_block
    Cc, frame, Mb = FRAME_C, PIX * FRAME_C, b * PIX
    dirs = ((trl.forward_f1, trl.forward_f2, trl.channel_atte_foreward_corr, trl.uncorr_memo_forward),
            (trl.backward_f1, trl.backward_f2, trl.channel_atte_backward_corr, trl.uncorr_memo_backward))
    tp.mark('trl')
    fk = engine._TrlFork(tp.dev, tp.taps is None)
    # per-direction aliases: same storage, their own gradient slots on the tape
    xu_d = (xu, xu.view_as(xu)) if fk.two else (xu, xu)
    memo_d = []             # the initial memo, one copy per direction (filled below: block 0 of the direction's f1 stack)

    def bwd_memo0():
        if fk.two:
            g1 = tp.take(xu_d[1])
            if g1 is not None:
                tp.add_grad(xu, g1)
        dm = None
        for m0 in memo_d:
            g1 = tp.take(m0)
            if g1 is not None and dm is not None:
                _call(_k('grl_axpby', dm), ptr(dm), ptr(g1), ptr(dm), C.c_float(1.0), C.c_float(1.0), dm.numel())
            elif g1 is not None:
                dm = g1
        if dm is None:
            return
        g = tp.full_grad(xu)
        add_rowbcast(g, dm, b * t, frame, t, 1.0 / t, 1)
    tp.ops.append(bwd_memo0)

    gapc = group_mean_op(tp, xc, b * t, PIX, Cc)          # full_grad(xc) accumulates inside
    gapc_d = (gapc, gapc.view_as(gapc)) if fk.two else (gapc, gapc)

    def bwd_gapc1():
        g1 = tp.take(gapc_d[1]) if fk.two else None
        if g1 is not None:
            tp.add_grad(gapc, g1)
    tp.ops.append(bwd_gapc1)
    f2 = [biased_conv_relu(tp, xc, b * t * PIX, d[1][0]) for d in dirs]

    def bwd_join():                 # backward: both directions are done (runs before f2's backward on the launch stream)
        fk.join()
        tp.held = []
    tp.ops.append(bwd_join)
    fk.fork()
    fc = []
    # Each direction applies f1 and its bottleneck t times with the same weights: the operands of their weight gradients
    # are stacked step by step (_WgradStack) and every layer gets ONE product over t * Mb rows.  stk[di] = stacks of
    # (f1, conv1, conv2, conv3); f1's input stack is the memo sequence memo_0 .. memo_t (t + 1 blocks).
    stk, att = [], []
    for di, (_, _, _, blk) in enumerate(dirs):
        with fk.on(di):
            fc.append(torch.zeros((b, t, Cc), dtype=torch.float32, device=tp.dev) if (fk.two or di == 0) else fc[0])
            if WGRAD_STACK:
                stk.append((_WgradStack(t, Mb, Cc, xu, extra=1), _WgradStack(t, Mb, Cc, xu),
                            _WgradStack(t, Mb, blk.conv2.weight.shape[1], xu), _WgradStack(t, Mb, blk.conv3.weight.shape[1], xu)))
                memo_d.append(stk[di][0].x_block(0))
            else:
                stk.append(None)
                memo_d.append(_newl((Mb, Cc), xu))
            _call(_k('grl_temporal_mean', xu), ptr(xu), ptr(memo_d[di]), b, t, frame)
            # (the channel-attention MLP is applied t times per direction with the same two weights: its operands are
            #  stacked like the recurrence's -- 2 x 2 weight-gradient products per direction instead of 2 x 2 x t tiny ones)
            att.append(dict(dvec=_new((t * b, Cc), xu), hid=_new((t * b, 128), xu), ds=None, dhp=None, left=t, steps=t) if WGRAD_STACK else None)
            if WGRAD_STACK:
                tp.stacks.extend(stk[di] + (att[di],))
    memo = list(memo_d)
    for i in range(t):
        for di, (f1m, _, mlp, blk) in enumerate(dirs):
            with fk.on(di):
                tp.side = fk.side if (fk.two and di == 1) else None
                ti = i if di == 0 else t - 1 - i
                sk = stk[di]
                f1 = biased_conv_relu(tp, memo[di], Mb, f1m[0], wstack=(sk[0], i) if sk else None)
                at = att[di]
                dvec = at['dvec'][i * b:(i + 1) * b] if at else _new((b, Cc), xu)
                f2t = f2[di]
                _call(_k('grl_sqdiff_mean', f1), ptr(f1), ptr(f2t[ti * PIX:]), ptr(dvec), b, PIX, Cc, t * frame)
                hid = at['hid'][i * b:(i + 1) * b] if at else _new((b, 128), xu)
                catte = _new((b, Cc), xu)
                w1, w2 = mlp[0].weight, mlp[2].weight
                w2t = tp.w_t(w2.detach(), w2)                   # [128][2048]
                fcd, gap_al, xu_al = fc[di], gapc_d[di], xu_d[di]
                _call('grl_channel_atte', ptr(dvec), ptr(w1.detach()), ptr(w2t), ptr(gapc[ti:]), t * Cc,
                      ptr(catte), ptr(fcd.view(b * t, Cc)[ti:]), t * Cc, 1, b, Cc, 128, ptr(hid))

                def bwd_atte(f1=f1, f2t=f2t, ti=ti, dvec=dvec, hid=hid, catte=catte, w1=w1, w2=w2, w2t=w2t, fcd=fcd,
                             gap_al=gap_al, at=at, i=i):
                    dfc = tp.g.get(id(fcd))
                    if dfc is None:
                        return                  # (the same for every step of the direction: nothing is left half-stacked)
                    if at and at['ds'] is None:
                        at['ds'], at['dhp'] = _new((t * b, Cc), xu), _new((t * b, 128), xu)
                    ds = at['ds'][i * b:(i + 1) * b] if at else _new((b, Cc), xu)
                    dgap = tp.full_grad(gap_al)
                    _call('grl_catte_bwd', ptr(dfc.view(b * t, Cc)[ti:]), t * Cc, ptr(gapc[ti:]), t * Cc,
                          ptr(catte), ptr(ds), ptr(dgap[ti:]), t * Cc, 1, b, Cc)
                    dhid = _new((b, 128), xu)
                    gemm(ds, w2t, dhid, b, 128, Cc)
                    dhp = at['dhp'][i * b:(i + 1) * b] if at else _new((b, 128), xu)
                    _call('grl_relu_bwd', ptr(dhid), ptr(hid), ptr(dhp), dhid.numel(), 0)
                    if not at:
                        wgrad(ds, hid, tp.pgrad(w2), b, Cc, 128)
                        wgrad(dhp, dvec, tp.pgrad(w1), b, 128, Cc)
                    else:
                        at['left'] -= 1
                        if at['left'] == 0:
                            wgrad(at['ds'], at['hid'], tp.pgrad(w2), t * b, Cc, 128)
                            wgrad(at['dhp'], at['dvec'], tp.pgrad(w1), t * b, 128, Cc)
                    dd = _new((b, Cc), xu)
                    gemm(dhp, tp.w_t(w1.detach(), w1), dd, b, Cc, 128)
                    # through d = mean (f1 - f2)^2
                    df1 = _newl((Mb, Cc), xu)
                    df2 = tp.full_grad(f2t)
                    _call(_k('grl_sqdiff_bwd', f1), ptr(f1), ptr(f2t[ti * PIX:]), ptr(dd), ptr(df1),
                          ptr(df2[ti * PIX:]), b, PIX, Cc, t * frame, 1)
                    tp.add_grad(f1, df1)
                tp.ops.append(bwd_atte)
                if tp.taps is not None:
                    tp.taps.setdefault(('fwd', 'bwd')[di] + '_catte', []).append(catte)

                s = sk[1].x_block(i) if sk else _newl((Mb, Cc), xu)
                _call(_k('grl_add_strided', xu), ptr(memo[di]), ptr(xu.view(-1)[ti * frame:]), ptr(s), b, frame, t * frame)
                prev = memo[di]

                def bwd_add(s=s, prev=prev, ti=ti, xu_al=xu_al):
                    dsum = tp.take(s)
                    if dsum is None:
                        return
                    _axpy_frame(tp.full_grad(xu_al), ti, dsum, b, t, frame)
                    tp.add_grad(prev, dsum)
                tp.ops.append(bwd_add)
                if sk:
                    o, _, _, _ = conv_bn(tp, s, b, 16, 8, blk.conv1, blk.bn1, True, out=sk[2].x_block(i), wstack=(sk[1], i))
                    o2, _, _, _ = conv_bn(tp, o, b, 16, 8, blk.conv2, blk.bn2, True, out=sk[3].x_block(i), wstack=(sk[2], i))
                    memo[di], _, _, _ = conv_bn(tp, o2, b, 16, 8, blk.conv3, blk.bn3, True, res=s, out=sk[0].x_block(i + 1),
                                                wstack=(sk[3], i))
                else:
                    o, _, _, _ = conv_bn(tp, s, b, 16, 8, blk.conv1, blk.bn1, True)
                    o2, _, _, _ = conv_bn(tp, o, b, 16, 8, blk.conv2, blk.bn2, True)
                    memo[di], _, _, _ = conv_bn(tp, o2, b, 16, 8, blk.conv3, blk.bn3, True, res=s)
                tp.fuse_ok.update((id(o), id(o2)))       # single conv_bn consumers: fused BatchNorm-backward reduce
                tp.side = None
    mf, mb_ = memo
    fk.join(mb_, fc[1])
    fcorr = fc[0]
    if fk.two:
        fcorr = _new((b, t, Cc), xu)
        _call('grl_add_strided', ptr(fc[0]), ptr(fc[1]), ptr(fcorr), 1, b * t * Cc, 0)
    f_uncorr = _new((b, Cc), xu)
    _call(_k('grl_group_mean', mf), ptr(mf), ptr(f_uncorr), b, PIX, Cc, Cc, C.c_float(1.0), 0)
    _call(_k('grl_group_mean', mb_), ptr(mb_), ptr(f_uncorr), b, PIX, Cc, Cc, C.c_float(1.0), 1)

    def bwd_funcorr():              # backward: runs FIRST of the TRL closures, on the launch stream; then the fork
        d = tp.take(f_uncorr)
        dfc = tp.g.get(id(fcorr))
        if dfc is not None:         # f_corr = fc[0] + fc[1]: both directions read the same upstream gradient
            tp.g[id(fc[0])] = tp.g[id(fc[1])] = dfc
            tp.foreign[dfc.untyped_storage().data_ptr()] = dfc
        if d is not None:
            tp.held.append(d)
            for di, m in enumerate((mf, mb_)):
                g = _newl((Mb, Cc), xu)
                add_rowbcast(g, d, Mb, Cc, PIX, 1.0 / PIX, 0)
                tp.add_grad(m, g)
                tp.held.append(g)
        fk.fork()
    tp.ops.append(bwd_funcorr)
    if tp.taps is not None:
        tp.taps.update(f_uncorr=f_uncorr, f_corr=fcorr)
    return f_uncorr, fcorr


class _GrlTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_box, inputs, *params):
        model = model_box[0]
        engine.touch_state(model)            # running statistics change below, unseen by torch
        tp = Tape(inputs.device)
        tp.b16 = _train_mode[0] == 'bf16s'
        if WEIGHT_PREP:
            wp = _weight_prep(model, tp.b16)
            if wp.ready:
                # every packed / transposed / bf16 weight of the step: one launch.  Nothing before layer 1 reads them
                # (the stem takes the raw conv weight), so the launch runs on the (idle) weight-gradient stream next to
                # the stem and the launch stream joins it in front of layer 1
                if WGRAD_STREAM and PREP_ASYNC:
                    ws = _wgrad_stream(tp.dev)
                    ws.wait_stream(torch.cuda.current_stream(tp.dev))      # after the optimizer's parameter update
                    with torch.cuda.stream(ws):
                        wp.run()
                        tp.prep_event = torch.cuda.Event()
                        tp.prep_event.record(ws)
                else:
                    wp.run()
                tp.wc.update(wp.cache)
            else:                                        # first step in this mode: per-layer path, logged
                tp.prep = wp
                tp.params_by_ptr = {p.untyped_storage().data_ptr(): p for p in params}
        tp.reserve_param_grads(params, cuts=_grl_cuts(model))
        tp.taps = getattr(model, '_grl_taps', None)
        b, t = inputs.shape[:2]
        x = inputs.contiguous().view(b * t, 3, 256, 128)
        x4 = trunk_train(tp, model, x)
        xu, xc, _ = gce_train(tp, model, x4, b, t)
        f_uncorr, f_corr = trl_train(tp, model, xu, xc, b, t)
        fc2d = f_corr.view(b * t, 2048)

        def bwd_alias():        # gradient of the 2-D view belongs to the 3-D f_corr
            g = tp.g.pop(id(fc2d), None)
            if g is not None:
                tp.g[id(f_corr)] = g.view(b, t, 2048)
        tp.ops.append(bwd_alias)
        x_corr = bn1d_l2norm(tp, fc2d, b * t, 2048, model.corr_bn)
        x_uncorr = bn1d_l2norm(tp, f_uncorr, b, 2048, model.uncorr_bn)
        ctx.tape, ctx.outs, ctx.params, ctx.keep = tp, (x_uncorr, x_corr), params, (x, fc2d)
        return x_uncorr.clone(), x_corr.view(b, t, 2048).clone()

    @staticmethod
    def backward(ctx, d_uncorr, d_corr):
        tp = ctx.tape
        xu_out, xc_out = ctx.outs
        if d_uncorr is not None:
            tp.g[id(xu_out)] = d_uncorr.contiguous()
            tp.foreign[tp.g[id(xu_out)].untyped_storage().data_ptr()] = tp.g[id(xu_out)]
        if d_corr is not None:
            tp.g[id(xc_out)] = d_corr.contiguous().view(xc_out.shape)
            tp.foreign[tp.g[id(xc_out)].untyped_storage().data_ptr()] = tp.g[id(xc_out)]
        tp.backward()
        tp.flush()                                   # (whatever no section mark has sent)
        if tp.prep is not None and not tp.prep.ready:
            tp.prep.build(tp.dev)                    # the step's weight re-layouts as one table, for the next steps
        grads = []
        for p in ctx.params:
            e = tp.pg.get(id(p))
            grads.append(e[1] if e is not None else None)
        ctx.tape = None
        tp.pg, tp._pview = {}, {}                    # the returned views must be the only references:
        return (None, None) + tuple(grads)           # autograd then adopts them as p.grad (no copy)


def _grl_cuts(model):
    """Index (in model.parameters() order) of the first parameter of the sections whose completion
    releases a gradient bucket: [0, layer3) -> 'stem', [layer3, layer4) -> 'layer3',
    [layer4, TRL) (layer 4 + GCE) -> 'layer4', [TRL, end) (TRL + the two tail BatchNorms) -> 'trl'."""
    cuts = getattr(model, '_grl_cut_cache', None)
    if cuts is None:
        names = [n for n, _ in model.named_parameters()]
        first = lambda prefix: next(i for i, n in enumerate(names) if n.startswith(prefix))
        cuts = {'stem': 0, 'layer3': first('backbone.base.6.'), 'layer4': first('backbone.base.7.'),
                'trl': first('temporal_learning_block.')}
        model._grl_cut_cache = cuts
    return cuts


def grl_forward_train(model, inputs):
    params = tuple(model.parameters())
    return _GrlTrainFn.apply([model], inputs, *params)


# ----------------------------------------------------------------------------
# Siamese heads
# ----------------------------------------------------------------------------
class VerifyEvalPlan(EvalPlan):
    def __init__(self, head):
        super().__init__(head)
        self.scale, self.shift = self.fold(head.classifierBN)
        self.w = head.classifierlinear.weight.detach().contiguous()
        self.b = head.classifierlinear.bias.detach().contiguous()


def _verify_eval(head, probe, gallery):
    p = engine._plan(head, VerifyEvalPlan)
    nb, k = probe.shape
    ncls = p.w.shape[0]
    out = _new((nb, gallery.shape[0], ncls), probe)
    _call('grl_pair_verify', ptr(probe), ptr(gallery), ptr(p.scale), ptr(p.shift), ptr(p.w),
          ptr(p.b), ptr(out), nb, gallery.shape[0], k, ncls)
    return out


def attn_train(tp, siam, x, b, t):
    """Siamese.self_attention in train mode on x [b][t][C] (contiguous)."""
    Cc, D = x.shape[2], siam.featQ.out_features
    M = b * t
    wqk = torch.cat((siam.featQ.weight.detach(), siam.featK.weight.detach()), 0).contiguous()
    bqk = torch.cat((siam.featQ.bias.detach(), siam.featK.bias.detach())).contiguous()
    z = _new((M, 2 * D), x)
    gemm(x, wqk, z, M, 2 * D, Cc, shift=bqk)
    rows = _lib.load().grl_col_stats_rows(M)
    # the two BatchNorms (featQ_bn | featK_bn) finalize straight into the halves of one (4, 2D) block: the apply and
    # the backward below run over all 2D channels at once
    cat4 = _new((4, 2 * D), x)
    both = _BNState()
    both.mean, both.invstd, both.scale = cat4[0], cat4[1], cat4[2]
    gb = torch.cat((siam.featQ_bn.weight.detach(), siam.featK_bn.weight.detach(),
                    siam.featQ_bn.bias.detach(), siam.featK_bn.bias.detach())).view(2, 2 * D)    # gamma | beta
    both.beta = gb[1]
    for h, bn in enumerate((siam.featQ_bn, siam.featK_bn)):
        slab = _new((rows, 2, D), x)
        zh = z[:, h * D:]
        _call('grl_col_stats', ptr(zh), ptr(slab), M, D, 2 * D, ptr(zh))
        bn_finalize(slab, rows, D, M, bn, tp.dev, pivot=zh, out=cat4[:, h * D:(h + 1) * D])
    qk = _new((M, 2 * D), x)
    bn_apply(z, both, None, qk, M, 2 * D, False)
    out = _new((b, Cc), x)
    _call('grl_siamese_attn', ptr(qk), ptr(x), ptr(out), b, t, D, Cc, Cc)

    def bwd():
        dout = tp.take(out)
        if dout is None:
            return
        dqk = _new((M, 2 * D), x)
        cur = tp.g.get(id(x))
        acc = 1
        if cur is None:
            cur = _new(tuple(x.shape), x)
            tp.g[id(x)] = cur
            acc = 0
        _call('grl_siamese_attn_bwd', ptr(qk), ptr(x), ptr(out), Cc, ptr(dout), Cc, ptr(dqk), ptr(cur), acc,
              b, t, D, Cc)
        gamma = gb[0]                                           # (the optimizer steps after the backward)
        acc3 = torch.zeros(3, 2 * D, device=tp.dev)          # dgamma | dbeta | dbias of the Q|K halves: one fill
        dg, db, dbias = acc3[0], acc3[1], acc3[2]
        dz = bn_backward(dqk, z, None, both, gamma, dg, db, M, 2 * D)
        colsum_into(dz, M, 2 * D, dbias)
        dw = torch.empty(2 * D, Cc, dtype=torch.float32, device=tp.dev)
        wgrad(dz, x, dw, M, 2 * D, Cc, accumulate=0)
        # the eight parameter-gradient accumulations as ONE multi-tensor launch (the same fp32 adds)
        torch._foreach_add_(
            [tp.pgrad(siam.featQ_bn.weight), tp.pgrad(siam.featK_bn.weight), tp.pgrad(siam.featQ_bn.bias),
             tp.pgrad(siam.featK_bn.bias), tp.pgrad(siam.featQ.bias), tp.pgrad(siam.featK.bias),
             tp.pgrad(siam.featQ.weight), tp.pgrad(siam.featK.weight)],
            [dg[:D], dg[D:], db[:D], db[D:], dbias[:D], dbias[D:], dw[:D], dw[D:]])
        dx = _new((M, Cc), x)
        gemm(dz, tp.w_t(wqk, ('wqk', id(siam.featQ.weight))), dx, M, Cc, 2 * D)
        tp.add_grad(x, dx.view(tuple(x.shape)))
    tp.ops.append(bwd)
    return out


def verify_train(tp, head, probe, gallery):
    """(p_i - g_j)^2 -> BN1d(train) -> Linear(K, ncls) on all probe x gallery pairs."""
    nb, K = probe.shape
    ng = gallery.shape[0]
    P = nb * ng
    bn, lin = head.classifierBN, head.classifierlinear
    ncls = lin.out_features
    diff = _new((P, K), probe)
    _call('grl_pair_sqdiff', ptr(probe), ptr(gallery), ptr(diff), nb, ng, K)
    rows = _lib.load().grl_col_stats_rows(P)
    slab = _new((rows, 2, K), probe)
    _call('grl_col_stats', ptr(diff), ptr(slab), P, K, K, ptr(diff))
    st = bn_finalize(slab, rows, K, P, bn, tp.dev, pivot=diff)
    dn = _new((P, K), probe)
    bn_apply(diff, st, None, dn, P, K, False)
    wpad = torch.zeros(32, K, dtype=torch.float32, device=tp.dev)
    wpad[:ncls] = lin.weight.detach()
    bpad = _pad32(lin.bias)
    cls32 = _new((P, 32), probe)
    gemm(dn, wpad, cls32, P, 32, K, shift=bpad)
    cls = cls32[:, :ncls].contiguous().view(nb, ng, ncls)

    def bwd():
        dcls = tp.take(cls)
        if dcls is None:
            return
        d32 = torch.zeros((P, 32), dtype=torch.float32, device=tp.dev)
        d32[:, :ncls] = dcls.reshape(P, ncls)
        db = torch.zeros(32, device=tp.dev)
        colsum_into(d32, P, 32, db)
        dw = torch.empty(32, K, dtype=torch.float32, device=tp.dev)
        wgrad(d32, dn, dw, P, 32, K, accumulate=0)
        torch._foreach_add_([tp.pgrad(lin.bias), tp.pgrad(lin.weight)], [db[:ncls], dw[:ncls]])
        ddn = _new((P, K), probe)
        gemm(d32, tp.w_t(wpad, ('wcls', id(lin.weight))), ddn, P, K, 32)
        dd = bn_backward(ddn, diff, None, st, bn.weight, tp.pgrad(bn.weight), tp.pgrad(bn.bias), P, K)
        dp, dg = _new((nb, K), probe), _new((ng, K), probe)
        _call('grl_pair_sqdiff_bwd', ptr(probe), ptr(gallery), ptr(dd), ptr(dp), ptr(dg), nb, ng, K)
        tp.add_grad(probe, dp)
        tp.add_grad(gallery, dg)
    tp.ops.append(bwd)
    return cls


class _SiameseTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, box, x, *params):
        siam = box[0]
        engine.touch_state(siam)
        tp = Tape(x.device)
        tp.reserve_param_grads(params)
        bsz, t, d = x.shape
        half = bsz // 2
        xv = x.contiguous().view(half, 2, t, d)
        xp, xg = xv[:, 0].contiguous(), xv[:, 1].contiguous()
        # the reference pools the probe half first, then the gallery half (two BN calls each)
        pp = attn_train(tp, siam, xp, half, t)
        pg = attn_train(tp, siam, xg, half, t)
        cls = verify_train(tp, siam, pp, pg)
        ctx.tape, ctx.parts, ctx.params, ctx.shape = tp, (xp, xg, pp, pg, cls), params, (bsz, t, d)
        return cls.clone(), torch.cat((pp, pg)).contiguous()

    @staticmethod
    def backward(ctx, dcls, dout):
        tp = ctx.tape
        xp, xg, pp, pg, cls = ctx.parts
        bsz, t, d = ctx.shape
        half = bsz // 2
        if dcls is not None:
            tp.g[id(cls)] = dcls.contiguous()
        if dout is not None:
            dout = dout.contiguous()
            tp.add_grad(pp, dout[:half].clone())
            tp.add_grad(pg, dout[half:].clone())
        tp.backward()
        dx = torch.zeros((half, 2, t, d), dtype=torch.float32, device=xp.device)
        gp, gg = tp.take(xp), tp.take(xg)
        if gp is not None:
            dx[:, 0] = gp.view(half, t, d)
        if gg is not None:
            dx[:, 1] = gg.view(half, t, d)
        tp.flush()
        grads = [tp.pg[id(p)][1] if id(p) in tp.pg else None for p in ctx.params]
        ctx.tape = None
        tp.pg, tp._pview = {}, {}
        return (None, dx.view(bsz, t, d)) + tuple(grads)


class _VerifyTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, box, x, *params):
        head = box[0]
        engine.touch_state(head)
        tp = Tape(x.device)
        tp.reserve_param_grads(params)
        bsz = x.shape[0]
        half = bsz // 2
        xv = x.contiguous().view(half, 2, -1)
        xp, xg = xv[:, 0].contiguous(), xv[:, 1].contiguous()
        cls = verify_train(tp, head, xp, xg)
        ctx.tape, ctx.parts, ctx.params, ctx.bsz = tp, (xp, xg, cls), params, bsz
        return cls.clone(), torch.cat((xp, xg)).contiguous()

    @staticmethod
    def backward(ctx, dcls, dout):
        tp = ctx.tape
        xp, xg, cls = ctx.parts
        half = ctx.bsz // 2
        if dcls is not None:
            tp.g[id(cls)] = dcls.contiguous()
        tp.backward()
        dx = torch.zeros((half, 2, xp.shape[1]), dtype=torch.float32, device=xp.device)
        gp, gg = tp.take(xp), tp.take(xg)
        if gp is not None:
            dx[:, 0] = gp
        if gg is not None:
            dx[:, 1] = gg
        if dout is not None:
            dx[:, 0].add_(dout[:half])
            dx[:, 1].add_(dout[half:])
        tp.flush()
        grads = [tp.pg[id(p)][1] if id(p) in tp.pg else None for p in ctx.params]
        ctx.tape = None
        tp.pg, tp._pview = {}, {}
        return (None, dx.view(ctx.bsz, -1)) + tuple(grads)


def siamese_self_attention_train(siam, x):
    raise NotImplementedError(
        'Siamese.self_attention is only called in eval mode by the reference '
        '(attevaluator.py:107); call siamese.eval() first')


def siamese_forward(siam, x):
    """Siamese.forward: de-interleave (probe, gallery) pairs, pool each half with the
    temporal attention, verification head on all probe x gallery pairs."""
    if siam.training:
        return _SiameseTrainFn.apply([siam], x, *tuple(siam.parameters()))
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
        return _VerifyTrainFn.apply([head], x, *tuple(head.parameters()))
    with torch.no_grad():
        bsz = x.shape[0]
        xv = x.contiguous().view(bsz // 2, 2, -1)
        out = torch.cat((xv[:, 0], xv[:, 1])).contiguous()
        half = bsz // 2
        cls = _verify_eval(head, out[:half], out[half:])
        return cls, out
