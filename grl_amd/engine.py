"""Host-side orchestration of the GRL hot path on MI355X.

Python here only owns device memory (torch tensors), weight packing caches and
the launch order; every FLOP is issued through the C ABI in include/grl_hip.h
(libgrl_hip.so, hand-written HIP for gfx950).  Activations are channels-last
row-major matrices [N*H*W][C] from the stem to the TRL head.

Reference call sites (relative to /root/reference):
  grl_forward             reid/models/grl_model.py:211-228
    trunk                 reid/models/resnets1.py:73-93,101-109
    GCE                   reid/models/basebranch.py:52-68
    TRL                   reid/models/grl_model.py:131-180
  siamese_self_attention  reid/models/Siamese.py:79-106
  extract_features        reid/evaluator/attevaluator.py:100-112
  cosin_dist/pairwise     reid/evaluator/attevaluator.py:33-46
"""
import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import GrlGemm, GrlBneckTail, GrlBneckTailF32, EPI_AFFINE, EPI_NEGDOT, EPI_EUCLID, EPI_SQDIFF, check, ptr, require_device

import contextlib
import os

from ._lib import MATH_F32, MATH_BF16, MATH_BF16X3, MATH_BF16S

_MATH_NAMES = {'f32': MATH_F32, 'bf16': MATH_BF16, 'bf16x3': MATH_BF16X3, 'bf16s': MATH_BF16S}
# Multiplier datapath of the conv / linear GEMMs (accumulation is always fp32):
#   'f32'    exact fp32 MFMA, the default and the mode every parity claim is made in;
#   'bf16x3' split-bf16 (hi*hi + hi*lo + lo*hi), ~2^-16 relative per product;
#   'bf16'   operands rounded to bf16 while staging, activations still fp32 in HBM;
#   'bf16s'  bf16 STORAGE: activations and weights are bf16 in HBM from the stem to the TRL
#            memo (BASELINE configs[2] pipeline); per-clip vectors and the tail stay fp32.
# The evaluator distance matrices always use 'f32' (bit-exact ranking contract).
_math = [_MATH_NAMES[os.environ.get('GRL_MATH', 'f32')]]


def set_math(name):
    _math[0] = _MATH_NAMES[name]


def get_math():
    return {v: k for k, v in _MATH_NAMES.items()}[_math[0]]


@contextlib.contextmanager
def math_mode(name):
    old = _math[0]
    set_math(name)
    try:
        yield
    finally:
        _math[0] = old


FUSE_TRL_SQDIFF = True     # TRL step: (ReLU(f1) - f2)^2 GAP inside the f1 GEMM epilogue (A/B switch for tools/)
PIX = 128            # 16 x 8 feature map of layer4 (basebranch.py:59 hard-codes it)


# ----------------------------------------------------------------------------
# thin launch wrappers
# ----------------------------------------------------------------------------
def _new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# Raw uint8 clips are normalised on the device with the constants of the reference's loaders
# (reid/data/dataloader.py:20,51: ToTensor + Normalize(mean, std)) -- inside the stem in eval
# mode, by grl_normalize_u8 in train mode.  SURVEY.md 8(f) rank 4.
INPUT_MEAN = (0.485, 0.456, 0.406)
INPUT_STD = (0.229, 0.224, 0.225)
_mean_std = {}


def input_mean_std(dev):
    t = _mean_std.get(dev)
    if t is None:
        t = _mean_std[dev] = torch.tensor(INPUT_MEAN + INPUT_STD, dtype=torch.float32, device=dev)
    return t


def normalize_u8(x):
    """uint8 [..., 3, H, W] -> float32, (x/255 - mean)/std per channel (bit-identical to the host
    ToTensor + Normalize of seqtransforms.py:190,212-213)."""
    require_device(x, 'clips', allow_u8=True)
    if x.dtype != torch.uint8:
        return x
    x = x.contiguous()
    y = _new(x.shape, x)
    plane = x.shape[-1] * x.shape[-2]
    _call('grl_normalize_u8', ptr(x), ptr(input_mean_std(x.device)), ptr(y), x.numel() // (3 * plane), plane)
    return y


_resize_tables = {}


def rect_scale_u8(x, height=256, width=128):
    """RectScale(height, width) of the reference's loaders (seqtransforms.py:30-47, dataloader.py:53,68)
    on the device: uint8 frames [..., 3, Hin, Win] -> [..., 3, height, width], bit-identical to PIL's
    BILINEAR resize (frames that already have the size are returned as they are, as upstream)."""
    require_device(x, 'frames', allow_u8=True)
    if x.dtype != torch.uint8:
        raise ValueError('rect_scale_u8 expects raw uint8 frames')
    hin, win = x.shape[-2:]
    if (hin, win) == (height, width):
        return x
    key = (hin, win, height, width, x.device)
    tabs = _resize_tables.get(key)
    if tabs is None:
        from .reid.data.augment import pil_bilinear_coeffs
        bh, ch = pil_bilinear_coeffs(win, width)
        bv, cv = pil_bilinear_coeffs(hin, height)
        tabs = _resize_tables[key] = tuple(torch.from_numpy(t).contiguous().to(x.device) for t in (bh, ch, bv, cv))
    bh, ch, bv, cv = tabs
    x = x.contiguous()
    y = torch.empty(x.shape[:-2] + (height, width), dtype=torch.uint8, device=x.device)
    _call('grl_resize_bilinear_u8', ptr(x), ptr(y), ptr(bh), ptr(ch), ch.shape[1], ptr(bv), ptr(cv), cv.shape[1],
          x.numel() // (hin * win), hin, win, height, width)
    return y


def augment_normalize_u8(clips, params):
    """Training augmentation on the device: uint8 clips [B,T,3,H,W] + the host-drawn decisions
    (int32 [B, 1 + 8T], grl_amd.reid.data.augment) -> float32 clips, flipped / erased / normalised
    exactly as the reference's PIL transforms would (seqtransforms.py:92-190, dataloader.py:51-57)."""
    require_device(clips, 'clips', allow_u8=True)
    if clips.dtype != torch.uint8 or clips.dim() != 5 or clips.shape[2] != 3:
        raise ValueError('augment_normalize_u8 expects uint8 clips [B,T,3,H,W]')
    b, t, _, h, w = clips.shape
    params = params.to(device=clips.device, dtype=torch.int32).contiguous()
    if tuple(params.shape) != (b, 1 + 8 * t):
        raise ValueError('augmentation parameters must be [B, 1 + 8*T] (got %s)' % (tuple(params.shape),))
    clips = clips.contiguous()
    y = _new(clips.shape, clips)
    _call('grl_augment_normalize_u8', ptr(clips), ptr(params), ptr(input_mean_std(clips.device)), ptr(y), b, t, h, w)
    return y


FUSE_BNECK = os.environ.get('GRL_FUSE_BNECK', '1') != '0'       # A/B and tests: 0 = one launch per convolution
FUSE_STEM_POOL = os.environ.get('GRL_FUSE_STEM_POOL', '1') != '0'   # A/B and tests: 0 = stem and max-pool as two launches (bf16 storage)
FUSE_STEM_POOL_F32 = os.environ.get('GRL_FUSE_STEM_POOL_F32', '1') != '0'   # ... the exact-fp32 path (round 5)
FUSE_DOWN = os.environ.get('GRL_FUSE_DOWN', '1') != '0'         # A/B and tests: 0 = the downsample conv as its own launch
SLAB_CHECK = False  # tests only: poison every statistics slab and verify that the GEMM wrote all of it
SPLITK = True       # tests / A-B only: False = never hand the library split-K scratch (one workgroup per tile walks K)


def gemm(a, w, y, M, N, K, lda=0, ldw=None, ldy=None, scale=None, shift=None, res=None,
         ldres=0, gbias=None, rows_per_group=0, rowscale=None, relu=False,
         epilogue=EPI_AFFINE, rnorm=None, cnorm=None, stats=None, conv=None, math=None, out_f32=False,
         kblock=False, res_rows=0, res_gstride=0, bn=None):
    """Y[M][N] = epilogue(A . W^T) through grl_conv_gemm_f32.  ``conv`` is
    (H, W, C, Ho, Wo, kh, kw, stride, pad) for an implicit-GEMM convolution.
    ``stats=True`` allocates and returns the per-channel partial-sum slab
    (train-mode BatchNorm) as ``(y, slab)``."""
    want_stats = stats is True
    if want_stats:
        stats = None
    # one positional constructor call in the struct's field order (include/grl_hip.h GrlGemm / _lib.GrlGemm) instead of ~30
    # attribute stores: this wrapper runs ~220 times per training step, which is host-bound in bf16 storage (round 6)
    cv = (1,) + tuple(conv) if conv is not None else (0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    bnp = tuple(ptr(t) for t in bn) if bn is not None else ()        # BatchNorm-backward reduce in the epilogue (GrlGemm.bn_z; with stats=True)
    d = GrlGemm(ptr(a), ptr(w), ptr(y), ptr(scale), ptr(shift), ptr(res), ptr(gbias), ptr(rowscale), ptr(rnorm), ptr(cnorm),
                ptr(stats), M, N, K, lda or K, ldw or K, ldy or N, ldres or N, rows_per_group, 1 if relu else 0, epilogue,
                cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], cv[8], cv[9],
                (MATH_F32 if _math[0] == MATH_BF16S else _math[0]) if math is None else math, 1 if out_f32 else 0,
                res_rows, res_gstride, 1 if kblock else 0, None, 0, *bnp)
    lib = _lib.load()
    if SPLITK and kblock and conv is None and M <= 256 and K > 512:      # skinny K-blocked GEMM: split-K scratch (include/grl_hip.h)
        need = lib.grl_conv_gemm_f32_workspace_floats(C.byref(d))
        if need > 0:
            ws = torch.empty(need, dtype=torch.float32, device=y.device)
            d.splitk_ws, d.splitk_ws_floats = ptr(ws), need
    if want_stats:
        rows = lib.grl_conv_gemm_f32_stat_rows(C.byref(d))
        # (`rows` is the row count of the kernel that takes THIS launch -- two per 256-row tile of the bf16 256 x 256 kernel,
        #  one per tile row of the 128-row family -- and every one of them is written, ragged last tiles included: no fill.
        #  Rounds 2-4 zero-filled the bf16-storage slabs defensively: 79 fill launches per training step.  SLAB_CHECK
        #  (tests): poison the slab and verify after the launch that nothing of the poison is left.)
        slab = torch.empty((rows, 2, N), dtype=torch.float32, device=y.device)
        if SLAB_CHECK:
            slab.fill_(float('nan'))
        d.stats = ptr(slab)
        check(lib.grl_conv_gemm_f32(C.byref(d), _lib.stream()), 'grl_conv_gemm_f32')
        if SLAB_CHECK and bool(torch.isnan(slab).any()):
            raise AssertionError('statistics slab of gemm %s math %d conv %s: %d of %d rows not written' % (
                (M, N, K), d.math, conv, int(torch.isnan(slab).any(dim=2).any(dim=1).sum()), rows))
        if _DEBUG_SYNC:
            _debug_sync('gemm+stats %s math %d conv %s' % ((M, N, K), d.math, conv))
        return y, slab
    check(lib.grl_conv_gemm_f32(C.byref(d), _lib.stream()), 'grl_conv_gemm_f32')
    if _DEBUG_SYNC:
        _debug_sync('gemm %s math %d conv %s' % ((M, N, K), d.math, conv))
    return y


def gemm_group(calls):
    """Several dense GEMMs of ONE shape in one launch where the library can group them (grl_conv_gemm_f32_group;
    bit-identical to ``gemm(**c)`` for every c in order, which is also what it falls back to).  ``calls``: list of
    dicts with the plain-affine keyword subset of :func:`gemm` (a, w, y, M, N, K, scale, shift, res, relu, math, ld*)."""
    n = len(calls)
    arr = (GrlGemm * n)()
    for d, c in zip(arr, calls):
        d.a, d.w, d.y = ptr(c['a']), ptr(c['w']), ptr(c['y'])
        d.scale, d.shift, d.res = ptr(c.get('scale')), ptr(c.get('shift')), ptr(c.get('res'))
        d.M, d.N, d.K = c['M'], c['N'], c['K']
        d.lda = c.get('lda') or d.K
        d.ldw = c.get('ldw') or d.K
        d.ldy = c.get('ldy') or d.N
        d.ldres = c.get('ldres') or d.N
        d.relu = 1 if c.get('relu') else 0
        d.epilogue = EPI_AFFINE
        m = c.get('math')
        d.math = (MATH_F32 if _math[0] == MATH_BF16S else _math[0]) if m is None else m
    check(_lib.load().grl_conv_gemm_f32_group(arr, n, _lib.stream()), 'grl_conv_gemm_f32_group')
    if _DEBUG_SYNC:
        _debug_sync('gemm_group x%d %s' % (n, (calls[0]['M'], calls[0]['N'], calls[0]['K'])))
    return [c['y'] for c in calls]


_DEBUG_SYNC = bool(os.environ.get('GRL_DEBUG_SYNC'))      # debugging only: name every launch on stderr and wait for it


def _debug_sync(name):
    import sys
    sys.stderr.write('[grl] %s\n' % name)
    sys.stderr.flush()
    torch.cuda.synchronize()


_fns = {}


def _kb():
    """The per-clip linears of the eval path (global descriptor, its bias term, the Siamese Q|K projection: M = clips or
    frames, K = 1024..2048) accumulate K-BLOCKED in the exact-fp32 datapath: their 512-k segments then run as separate
    workgroups (GrlGemm.splitk_ws) instead of one workgroup per tile walking all of K -- 55 -> ~15 us each -- and the
    result does not depend on whether the library splits (M <= 256) or not: a clip's row stays batch-independent."""
    return _math[0] in (MATH_F32, MATH_BF16S)


def _call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(_lib.load(), name)
    rc = fn(*args, _lib.stream())
    if rc:
        check(rc, name)
    if _DEBUG_SYNC:
        _debug_sync(name)


# ----------------------------------------------------------------------------
# packed parameters
# ----------------------------------------------------------------------------
class _Conv(object):
    """A conv (or linear) with its eval-folded affine."""
    __slots__ = ('w', 'N', 'K', 'ldw', 'scale', 'shift', 'k', 'stride', 'cin', '_wb', '_wperm')

    def wb(self):
        """bf16 copy of the packed weight (bf16-storage pipeline), made on first use."""
        if getattr(self, '_wb', None) is None:
            w = self.w.contiguous()
            self._wb = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
            _call('grl_cast_bf16', ptr(w), ptr(self._wb), w.numel())
        return self._wb

    def wperm(self):
        """bf16 copy of a 1x1 weight [N][K] in the k order the chained MFMA of grl_bottleneck_tail_bf16 consumes
        (grl_bneck_perm32), made on first use."""
        if getattr(self, '_wperm', None) is None:
            w = self.w.contiguous()
            self._wperm = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
            _call('grl_bneck_perm32', ptr(w), 0, ptr(self._wperm), w.shape[0], w.shape[1])
        return self._wperm


def _state_key(module):
    """Identity of a module's state for the packed-plan caches: (pointer, torch version counter)
    of every parameter / buffer plus the module's train-forward generation.  The train-mode
    kernels update BatchNorm running statistics through raw device pointers, which torch's
    version counters never see -- ``touch_state`` (called by every train-mode forward) makes those
    updates visible here, so eval -> train forward under no_grad -> eval re-folds the BatchNorms."""
    return (getattr(module, '_grl_generation', 0),) + tuple(
        (t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def touch_state(module):
    """Mark ``module``'s buffers as changed behind torch's back (see ``_state_key``)."""
    module._grl_generation = getattr(module, '_grl_generation', 0) + 1


class EvalPlan(object):
    """Device-side packing of a model's parameters for the eval forward:
    3x3 weights re-laid tap-major, BatchNorm folded to scale/shift, Q|K weights
    concatenated.  Rebuilt whenever a parameter/buffer changes (version
    counters), so load_state_dict / optimizer steps are picked up."""

    def __init__(self, module):
        self.key = _state_key(module)
        self.dev = next(module.parameters()).device

    # -- helpers ---------------------------------------------------------------
    def fold(self, bn=None, bias=None, n=None):
        """(scale, shift) of eval BN (optionally after a biased layer)."""
        n = n if n is not None else (bn.num_features if bn is not None else bias.numel())
        scale = torch.empty(n, dtype=torch.float32, device=self.dev)
        shift = torch.empty(n, dtype=torch.float32, device=self.dev)
        if bn is not None:
            _call('grl_bn_fold', ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                  ptr(bn.running_var), ptr(bias), C.c_float(bn.eps), ptr(scale), ptr(shift), n)
        else:
            _call('grl_bn_fold', None, None, None, None, ptr(bias), C.c_float(0.0), ptr(scale),
                  ptr(shift), n)
        return scale, shift

    def conv(self, conv, bn=None):
        c = _Conv()
        c._wb = None
        c._wperm = None
        w = conv.weight.detach()
        c.N, c.cin = w.shape[0], w.shape[1]
        c.k = w.shape[2] if w.dim() == 4 else 1
        c.stride = conv.stride[0] if hasattr(conv, 'stride') else 1
        if c.k == 1:
            c.w = w.contiguous().view(c.N, c.cin)
        else:
            c.w = torch.empty(c.N, c.k * c.k * c.cin, dtype=torch.float32, device=self.dev)
            wc = w.contiguous()          # must outlive the launch below
            _call('grl_pack_conv_weight', ptr(wc), ptr(c.w), c.N, c.cin, c.k, c.k)
        c.K = c.w.shape[1]
        c.ldw = c.K
        bias = getattr(conv, 'bias', None)
        if bn is not None:
            c.scale, c.shift = self.fold(bn, bias.detach() if bias is not None else None)
        elif bias is not None:
            c.scale, c.shift = None, bias.detach()
        else:
            c.scale, c.shift = None, None
        return c


class GrlEvalPlan(EvalPlan):
    def __init__(self, model):
        super().__init__(model)
        bb = model.backbone
        base = bb.base
        self.stem_w = base[0].weight.detach().contiguous()
        self.stem_wp = torch.empty(64 * 164, dtype=torch.float32, device=self.dev)
        _call('grl_stem_pack_weight', ptr(self.stem_w), ptr(self.stem_wp))
        self.stem_wpb = torch.empty(64 * 184, dtype=torch.bfloat16, device=self.dev)
        _call('grl_stem_pack_weight_bf16', ptr(self.stem_w), ptr(self.stem_wpb))
        self.stem_wq = torch.empty(64 * 168, dtype=torch.float32, device=self.dev)
        _call('grl_stem_pack_weight_pool', ptr(self.stem_w), ptr(self.stem_wq))
        self.stem_scale, self.stem_shift = self.fold(base[1])
        self.blocks = []
        for li in (4, 5, 6, 7):
            for blk in base[li]:
                e = dict(c1=self.conv(blk.conv1, blk.bn1), c2=self.conv(blk.conv2, blk.bn2),
                         c3=self.conv(blk.conv3, blk.bn3), down=None, stride=blk.stride)
                if blk.downsample is not None:
                    e['down'] = self.conv(blk.downsample[0], blk.downsample[1])
                self.blocks.append(e)
        # GCE (basebranch.py:38-50)
        self.glo_fc = self.conv(bb.glo_fc[0], bb.glo_fc[1])
        self.corr0 = self.conv(bb.corr_atte[0], bb.corr_atte[1])        # [1024][3072]
        self.corr2 = self.conv(bb.corr_atte[2], bb.corr_atte[3])
        self.corr5_w = bb.corr_atte[5].weight.detach().contiguous().view(-1)
        self.corr6_scale, self.corr6_shift = self.fold(bb.corr_atte[6])
        # TRL (grl_model.py:93-128)
        trl = model.temporal_learning_block
        self.dirs = []
        for f1, f2, mlp, memo in (
                (trl.forward_f1, trl.forward_f2, trl.channel_atte_foreward_corr, trl.uncorr_memo_forward),
                (trl.backward_f1, trl.backward_f2, trl.channel_atte_backward_corr, trl.uncorr_memo_backward)):
            self.dirs.append(dict(
                f1=self.conv(f1[0]), f2=self.conv(f2[0]),
                w1=mlp[0].weight.detach().contiguous(),
                w2t=mlp[2].weight.detach().t().contiguous(),
                c1=self.conv(memo.conv1, memo.bn1), c2=self.conv(memo.conv2, memo.bn2),
                c3=self.conv(memo.conv3, memo.bn3)))
        self.corr_bn = self.fold(model.corr_bn)
        self.uncorr_bn = self.fold(model.uncorr_bn)


class SiameseEvalPlan(EvalPlan):
    def __init__(self, siam):
        super().__init__(siam)
        self.D = siam.featQ.out_features
        self.wqk = torch.cat((siam.featQ.weight.detach(), siam.featK.weight.detach()), 0).contiguous()
        sq, hq = self.fold(siam.featQ_bn, siam.featQ.bias.detach())
        sk, hk = self.fold(siam.featK_bn, siam.featK.bias.detach())
        self.scale = torch.cat((sq, sk)).contiguous()
        self.shift = torch.cat((hq, hk)).contiguous()


_plans = weakref.WeakKeyDictionary()


def _plan(module, cls):
    """Cached plan of class ``cls`` for ``module`` (a module may own several kinds: a Siamese
    has an attention plan and a verification-head plan); rebuilt when its state changes."""
    per = _plans.get(module)
    if per is None:
        per = {}
        _plans[module] = per
    p = per.get(cls)
    if p is None or p.key != _state_key(module):
        p = cls(module)
        per[cls] = p
    return p


# ----------------------------------------------------------------------------
# eval forward
# ----------------------------------------------------------------------------
def _conv_layer(x, c, n_img, H, W, stride=1, relu=True, res=None, **kw):
    """x: [n_img*H*W][cin] channels-last.  Returns (y, Ho, Wo)."""
    if c.k == 1 and stride == 1:
        M = n_img * H * W
        y = _new((M, c.N), x)
        gemm(x, c.w, y, M, c.N, c.K, ldw=c.ldw, scale=c.scale, shift=c.shift, res=res, relu=relu, **kw)
        return y, H, W
    pad = c.k // 2
    Ho = (H + 2 * pad - c.k) // stride + 1
    Wo = (W + 2 * pad - c.k) // stride + 1
    M = n_img * Ho * Wo
    y = _new((M, c.N), x)
    gemm(x, c.w, y, M, c.N, c.K, ldw=c.ldw, scale=c.scale, shift=c.shift, res=res, relu=relu,
         conv=(H, W, c.cin, Ho, Wo, c.k, c.k, stride, pad), **kw)
    return y, Ho, Wo


def _to_nchw(y, n, H, W):
    return y.view(n, H, W, -1).permute(0, 3, 1, 2)


STAGE_HOOK = None      # bench.py's per-stage timing pass: callable(stage name) at every stage boundary of the eval forward


def _stage(name):
    if STAGE_HOOK is not None:
        STAGE_HOOK(name)


def _stem_pool_ok(x, n):
    """what grl_stem_pool_{f32,bf16} require beyond the frame geometry (one grid row per frame; 2-byte loads of u8
    rows, 8-byte loads of fp32 rows): otherwise the two-launch stem + max-pool path takes the batch"""
    return n <= 65535 and x.data_ptr() % (2 if x.dtype == torch.uint8 else 8) == 0


def trunk_eval(plan, x, taps=None):
    """x [n,3,H,W] NCHW -> channels-last [n*16*8][2048] (for 256x128 input)."""
    n, _, H, W = x.shape
    Hs, Ws = H // 2, W // 2
    Hp, Wp = (Hs + 1) // 2, (Ws + 1) // 2
    cur = _new((n * Hp * Wp, 64), x)
    _stage('stem')
    if FUSE_STEM_POOL_F32 and taps is None and W == 128 and H % 4 == 0 and _stem_pool_ok(x, n):
        # stem + max-pool in one launch: the stem map never reaches HBM (grl_stem_pool_f32)
        u8 = x.dtype == torch.uint8
        _call('grl_stem_pool_f32', ptr(x), 1 if u8 else 0, ptr(input_mean_std(x.device)) if u8 else None,
              ptr(plan.stem_scale), ptr(plan.stem_shift), ptr(cur), n, H, W, ptr(plan.stem_wq))
    else:
        stem = _new((n * Hs * Ws, 64), x)
        if x.dtype == torch.uint8:           # raw pixels: normalised while the stem stages its patch
            _call('grl_stem_conv7x7_u8', ptr(x), ptr(input_mean_std(x.device)), ptr(plan.stem_w),
                  ptr(plan.stem_scale), ptr(plan.stem_shift), ptr(stem), n, H, W, 1, ptr(plan.stem_wp))
        else:
            _call('grl_stem_conv7x7', ptr(x), ptr(plan.stem_w), ptr(plan.stem_scale), ptr(plan.stem_shift),
                  ptr(stem), n, H, W, 1, ptr(plan.stem_wp))
        _call('grl_maxpool3x3s2', ptr(stem), ptr(cur), n, Hs, Ws, 64)
        if taps is not None:
            taps['stem'] = _to_nchw(stem, n, Hs, Ws)
            taps['pool'] = _to_nchw(cur, n, Hp, Wp)
        del stem
    H, W = Hp, Wp
    counts = (3, 4, 6, 3)
    bi = 0
    o1 = None
    for li, nb in enumerate(counts):
        _stage('layer%d' % (li + 1))       # (a fused tail computes the NEXT block's conv1: the first conv1 of layers 2 / 3 is booked here)
        for _ in range(nb):
            e = plan.blocks[bi]
            bi += 1
            s = e['stride']
            if o1 is None:
                o1, _, _ = _conv_layer(cur, e['c1'], n, H, W)
            o2, Ho, Wo = _conv_layer(o1, e['c2'], n, H, W, stride=s)
            if e['down'] is not None:
                res, _, _ = _conv_layer(cur, e['down'], n, H, W, stride=s, relu=False)
            else:
                res = cur
            nxt = plan.blocks[bi]['c1'] if bi < len(plan.blocks) else None
            o1 = None
            if FUSE_BNECK and nxt is not None and _bneck_tail_f32_ok(e['c3'], nxt, n * Ho * Wo):
                # layers 1-2: conv3 + residual + ReLU AND the next block's conv1 in one launch, bit-identical to the two
                # GEMM launches (fuse_f32.hip) -- the 4P-wide output is written once and never re-read
                cur, o1 = bneck_tail_f32(o2, e['c3'], res, nxt, n * Ho * Wo)
            else:
                cur, _, _ = _conv_layer(o2, e['c3'], n, Ho, Wo, res=res)
            H, W = Ho, Wo
        if taps is not None:
            taps['layer%d' % (li + 1)] = _to_nchw(cur, n, H, W)
    return cur, H, W


def gce_eval(plan, x4, b, t, taps=None):
    """x4 [b*t*128][2048] -> (x_uncorr, x_corr) same shape, corr_map [b*t*128]."""
    M = x4.shape[0]
    _stage('gce')
    x_glo = _new((b, 2048), x4)
    _call('grl_group_mean', ptr(x4), ptr(x_glo), b, t * PIX, 2048, 2048, C.c_float(1.0), 0)
    g = plan.glo_fc
    glo = _new((b, 1024), x4)
    gemm(x_glo, g.w, glo, b, 1024, 2048, scale=g.scale, shift=g.shift, relu=True, kblock=_kb())
    # W.[x; g] = Wx.x + Wg.g : the broadcast-concat of basebranch.py:59-61 becomes a
    # per-clip bias added inside the accumulator epilogue.
    c0 = plan.corr0
    gb = _new((b, 1024), x4)
    gemm(glo, c0.w[:, 2048:], gb, b, 1024, 1024, ldw=3072, kblock=_kb())
    h1 = _new((M, 1024), x4)
    gemm(x4, c0.w, h1, M, 1024, 2048, ldw=3072, gbias=gb, rows_per_group=t * PIX,
         scale=c0.scale, shift=c0.shift, relu=False)
    c2 = plan.corr2
    h2 = _new((M, 256), x4)
    gemm(h1, c2.w, h2, M, 256, 1024, scale=c2.scale, shift=c2.shift, relu=True)
    cmap = _new((M,), x4)
    xc = _new((M, 2048), x4)
    xu = _new((M, 2048), x4)
    _call('grl_gce_gate', ptr(h2), ptr(plan.corr5_w), ptr(plan.corr6_scale), ptr(plan.corr6_shift),
          ptr(x4), ptr(cmap), ptr(xc), ptr(xu), M, 256, 2048)
    if taps is not None:
        taps['x_glo'], taps['glo'] = x_glo, glo
        taps['corr_map'] = cmap.view(b * t, 1, 16, 8)
    return xu, xc, cmap


# The two TRL directions (forward / backward in time, grl_model.py:170-208) are independent recurrences over
# their own weights: each step's GEMMs have M = B*128 rows -- 128..256 tiles, half a chip -- and six small
# latency-bound kernels.  They are issued on two HIP streams (fork after the shared inputs, join before the
# pooled outputs) so the chip runs one direction's GEMM next to the other's small kernels.  Same kernels,
# same per-direction order, per-direction scratch: bit-identical to the single-stream order
# (GRL_TRL_STREAMS=0, or taps requested).
TRL_STREAMS = os.environ.get('GRL_TRL_STREAMS', '1') != '0'
TRL_GROUP = os.environ.get('GRL_TRL_GROUP', '0') != '0'       # bf16 storage: conv1 / conv2 of both directions as grouped launches (measured slower: see below)
# Round 5 (eval): each direction's ATTENTION branch of a step -- the f1 GEMM with its squared-difference epilogue and the
# three latency-bound kernels behind it (partial-sum GAP, the two channel-attention layers) -- only reads the step's memo
# and feeds f_corr, never the recurrence; in stream order it still sat in front of the step's add / conv1 / conv2 / conv3.
# It goes to a stream of its own (one per direction), forked from the direction's stream where the memo is ready, so the
# recurrence's GEMMs run next to it (knock-out bound: the small kernels cost 0.27 ms of configs[2] although nothing waits
# for their results before the join; measured configs[2] 10.03 -> 9.93-9.97 ms).  bf16 storage only: the exact-fp32 step
# LOSES 1 % to it (14.47 -> 14.63 ms; with only the small kernels moved 14.60) -- its f1 GEMMs are four times longer and
# sharing CUs costs them more than the bubbles they fill.  Same kernels on the same operands: bit-identical.
# GRL_TRL_ATT_STREAMS=0: the attention branch stays on its direction's stream.
TRL_ATT_STREAMS = os.environ.get('GRL_TRL_ATT_STREAMS', '1') != '0'
_side_streams = {}
_att_streams = {}


class _TrlFork(object):
    """streams[di] for the two TRL directions; ``with fork.on(di):`` routes launches and allocations;
    ``with fork.att_on(di):`` routes to the direction's attention stream (ordered after everything issued so far on
    the direction's own stream)."""

    def __init__(self, dev, enable, att=False):
        self.main = torch.cuda.current_stream(dev)
        self.two = bool(enable and TRL_STREAMS)
        self.att = None
        self.held = []          # tensors an attention stream reads: kept alive until the join
        if self.two:
            key = (dev.index if dev.index is not None else torch.cuda.current_device())
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(dev)
            self.side = _side_streams[key]
            if att and TRL_ATT_STREAMS:
                if key not in _att_streams:
                    _att_streams[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
                self.att = _att_streams[key]
        else:
            self.side = self.main

    def att_on(self, di, *reads):
        """``reads``: tensors of the direction's stream the branch reads (held until the join)."""
        if self.att is None:
            return self.on(di)
        ev = torch.cuda.Event()
        ev.record(self.side if di == 1 else self.main)
        self.att[di].wait_event(ev)
        self.held.extend(r for r in reads if r is not None)
        return torch.cuda.stream(self.att[di])

    def fork(self):
        if self.two:
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.side.wait_event(ev)

    def on(self, di):
        return torch.cuda.stream(self.side if di == 1 else self.main)

    def side_to_main(self):
        """main waits for everything issued on the side stream so far"""
        if self.two:
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)

    def main_to_side(self):
        """the side stream waits for everything issued on the main stream so far"""
        if self.two:
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.side.wait_event(ev)

    def join(self, *side_tensors):
        if self.two:
            for st in (self.side,) + (tuple(self.att) if self.att is not None else ()):
                ev = torch.cuda.Event()
                ev.record(st)
                self.main.wait_event(ev)
                if st is not self.side:
                    # blocks the attention streams read / write were allocated on the DIRECTION streams: once `held`
                    # is dropped the allocator may hand a side-stream block to the side stream's next launch, so the
                    # side stream is ordered behind the attention streams as well (main already is)
                    self.side.wait_event(ev)
            for x in side_tensors:
                x.record_stream(self.main)
            self.held = []


def trl_eval(plan, xu, xc, b, t, taps=None):
    """xu, xc [b][t][128][2048] (flat) -> f_uncorr [b][2048], f_corr [b][t][2048]."""
    Cc = 2048
    frame = PIX * Cc
    Mb = b * PIX
    _stage('trl')
    memo0 = _new((Mb, Cc), xu)
    _call('grl_temporal_mean', ptr(xu), ptr(memo0), b, t, frame)
    gapc = _new((b * t, Cc), xu)
    _call('grl_group_mean', ptr(xc), ptr(gapc), b * t, PIX, Cc, Cc, C.c_float(1.0), 0)
    # conv_f2(x_corr_i) does not depend on the recurrence: one GEMM over all T per direction
    fk = _TrlFork(xu.device, taps is None and len(plan.dirs) == 2)
    fk.fork()
    f2, fc, scr = [], [], []
    for di, d in enumerate(plan.dirs):
        with fk.on(di):
            y = _new((b * t * PIX, Cc), xu)
            gemm(xc, d['f2'].w, y, b * t * PIX, Cc, Cc, shift=d['f2'].shift, relu=True)
            f2.append(y)
            # per-direction accumulators / scratch (a + b == b + a: summing the two directions' f_corr
            # contributions at the join gives the bits the shared accumulator got in either arrival order)
            # (two streams: each direction has its own accumulator and writes every (clip, frame) row exactly once -- no
            #  zero fill, the attention kernel stores instead of accumulating)
            fc.append(_new((b, t, Cc), xu) if fk.two else (torch.zeros((b, t, Cc), dtype=torch.float32, device=xu.device) if di == 0 else fc[0]))
            scr.append((_new((b, Cc), xu), _new((Mb // 32, Cc), xu), _new((b, 128), xu)))
    memo = [memo0, memo0]
    catte = _new((b, Cc), xu) if taps is not None else None
    for i in range(t):
        for di, d in enumerate(plan.dirs):
            ti = i if di == 0 else t - 1 - i
            with fk.on(di):
                dvec, dpart, hid = scr[di]
                fcorr = fc[di]
                # d = GAP((ReLU(conv_f1(memo)) - f2_t)^2): the squared difference is reduced in the GEMM
                # epilogue (32-row partial sums), conv_f1's output never reaches HBM (grl_model.py:146-149)
                if FUSE_TRL_SQDIFF:
                    gemm(memo[di], d['f1'].w, dpart, Mb, Cc, Cc, shift=d['f1'].shift, epilogue=EPI_SQDIFF,
                         res=f2[di][ti * PIX:], res_rows=PIX, res_gstride=t * PIX)
                    _call('grl_group_mean', ptr(dpart), ptr(dvec), b, PIX // 32, Cc, Cc, C.c_float(1.0 / 32.0), 0)
                else:
                    f1 = _new((Mb, Cc), xu)
                    gemm(memo[di], d['f1'].w, f1, Mb, Cc, Cc, shift=d['f1'].shift, relu=True)
                    _call('grl_sqdiff_mean', ptr(f1), ptr(f2[di][ti * PIX:]), ptr(dvec), b, PIX, Cc, t * frame)
                _call('grl_channel_atte', ptr(dvec), ptr(d['w1']), ptr(d['w2t']), ptr(gapc[ti:]), t * Cc,
                      ptr(catte), ptr(fcorr.view(b * t, Cc)[ti:]), t * Cc, 0 if fk.two else 1, b, Cc, d['w1'].shape[0], ptr(hid))
                if taps is not None:
                    taps.setdefault(('fwd', 'bwd')[di] + '_catte', []).append(catte.clone())
            with fk.on(di):                                     # the recurrence
                s = _new((Mb, Cc), xu)
                _call('grl_add_strided', ptr(memo[di]), ptr(xu.view(-1)[ti * frame:]), ptr(s), b, frame, t * frame)
                o = _new((Mb, 512), xu)
                c1, c2, c3 = d['c1'], d['c2'], d['c3']
                gemm(s, c1.w, o, Mb, 512, Cc, scale=c1.scale, shift=c1.shift, relu=True)
                o2 = _new((Mb, 512), xu)
                gemm(o, c2.w, o2, Mb, 512, 512, scale=c2.scale, shift=c2.shift, relu=True)
                nm = _new((Mb, Cc), xu)
                gemm(o2, c3.w, nm, Mb, Cc, 512, scale=c3.scale, shift=c3.shift, res=s, relu=True)
                memo[di] = nm
    fk.join(memo[1], fc[1])
    fcorr = fc[0]
    if fk.two:
        fcorr = _new((b, t, Cc), xu)
        _call('grl_add_strided', ptr(fc[0]), ptr(fc[1]), ptr(fcorr), 1, b * t * Cc, 0)
    f_uncorr = _new((b, Cc), xu)
    _call('grl_group_mean', ptr(memo[0]), ptr(f_uncorr), b, PIX, Cc, Cc, C.c_float(1.0), 0)
    _call('grl_group_mean', ptr(memo[1]), ptr(f_uncorr), b, PIX, Cc, Cc, C.c_float(1.0), 1)
    if taps is not None:
        taps['f_uncorr'], taps['f_corr'] = f_uncorr, fcorr
    return f_uncorr, fcorr


def _grl_eval(model, inputs, taps=None, out_uncorr=None, ld_uncorr=2048):
    if _math[0] == MATH_BF16S:
        return _grl_eval_bf16s(model, inputs, taps, out_uncorr, ld_uncorr)
    plan = _plan(model, GrlEvalPlan)
    b, t, c, h, w = inputs.shape
    if (c, h, w) != (3, 256, 128):
        raise ValueError('GRL expects clips of [B,T,3,256,128] (got %s)' % (tuple(inputs.shape),))
    x = inputs.contiguous().view(b * t, c, h, w)
    x4, _, _ = trunk_eval(plan, x, taps)
    xu, xc, _ = gce_eval(plan, x4, b, t, taps)
    del x4
    f_uncorr, f_corr = trl_eval(plan, xu, xc, b, t, taps)
    _stage('tail')
    x_corr = _new((b, t, 2048), inputs)
    _call('grl_affine_l2norm', ptr(f_corr), ptr(plan.corr_bn[0]), ptr(plan.corr_bn[1]), ptr(x_corr),
          b * t, 2048, 2048)
    x_uncorr = out_uncorr if out_uncorr is not None else _new((b, 2048), inputs)
    _call('grl_affine_l2norm', ptr(f_uncorr), ptr(plan.uncorr_bn[0]), ptr(plan.uncorr_bn[1]),
          ptr(x_uncorr), b, 2048, ld_uncorr)
    return x_uncorr, x_corr


# ----------------------------------------------------------------------------
# bf16-storage eval forward (BASELINE configs[2])
# ----------------------------------------------------------------------------
def _newb(shape, like):
    return torch.empty(shape, dtype=torch.bfloat16, device=like.device)


FUSE_C64 = os.environ.get('GRL_CONV3X3_C64', '1') != '0'        # A/B and tests: 0 = layer 1's 3x3 convs on the generic kernel


def conv3x3_c64_bf16(x, c, n_img, H, W, relu=True):
    """Layer 1's 3x3 / stride 1, 64 -> 64 channels, W == 32 (resnets1.py:79-81): weights LDS-resident, each input pixel
    staged once per tile (grl_conv3x3_c64_bf16)."""
    y = _newb((n_img * H * W, 64), x)
    _call('grl_conv3x3_c64_bf16', ptr(x), ptr(c.wb()), ptr(c.scale), ptr(c.shift), ptr(y), n_img, H, W, 1 if relu else 0)
    return y


def _conv_b16(x, c, n_img, H, W, stride=1, relu=True, res=None, **kw):
    if (FUSE_C64 and c.k == 3 and stride == 1 and c.cin == 64 and c.N == 64 and W == 32 and H % 8 == 0 and res is None
            and not kw and n_img * H * W * 128 < (1 << 32)):          # (32-bit byte offsets inside that kernel)
        return conv3x3_c64_bf16(x, c, n_img, H, W, relu), H, W
    if c.k == 1 and stride == 1:
        M = n_img * H * W
        y = _newb((M, c.N), x)
        gemm(x, c.wb(), y, M, c.N, c.K, ldw=c.ldw, scale=c.scale, shift=c.shift, res=res, relu=relu,
             math=MATH_BF16S, **kw)
        return y, H, W
    pad = c.k // 2
    Ho, Wo = (H + 2 * pad - c.k) // stride + 1, (W + 2 * pad - c.k) // stride + 1
    M = n_img * Ho * Wo
    y = _newb((M, c.N), x)
    gemm(x, c.wb(), y, M, c.N, c.K, ldw=c.ldw, scale=c.scale, shift=c.shift, res=res, relu=relu,
         conv=(H, W, c.cin, Ho, Wo, c.k, c.k, stride, pad), math=MATH_BF16S, **kw)
    return y, Ho, Wo


FUSE_TAIL_L23 = os.environ.get('GRL_FUSE_TAIL_L23', '0') != '0'   # the layer 2 -> 3 tail (P 128, 4P 512, P' 256) fused too: measured slower


def _bneck_tail_ok(c3, c1n, M):
    # (the fused kernels address with 32-bit byte offsets: M * C4 * 2 bytes must stay below 4 GiB, else the per-conv
    #  launches -- 64-bit row addressing -- take the block)
    # (round 5: the P' = 256 variant -- layer 2's last block + layer 3's first conv1 -- runs 263 us against 240 us for the
    #  two launches it replaces (tools/bneck_tail_ab.py): a 256-wide second product leaves the kernel neither the
    #  registers (spills in its chunk loop at 16 waves) nor the occupancy; it stays available behind GRL_FUSE_TAIL_L23=1)
    if (c3.K, c3.N, c1n.N) == (128, 512, 256) and not FUSE_TAIL_L23:
        return False
    return (c3.k == 1 and c1n.k == 1 and c1n.K == c3.N and M * c3.N * 2 < (1 << 32) and
            bool(_lib.load().grl_bottleneck_tail_bf16_supported(c3.K, c3.N, c1n.N)))


def _bneck_down_ok(c3, c1n, down, stride):
    """The block's downsample branch (1x1, stride 1: layer 1's first block) can ride in the same launch."""
    return (down is not None and stride == 1 and down.k == 1 and c1n is not None and
            (c3.K, c3.N, c1n.N, down.K) == (64, 256, 64, 64))


def bneck_tail_bf16(t2, c3, res, c1n, M, down=None, x0=None):
    """y = relu(bn3(conv3(t2)) + res) [M][4P] and u = relu(bn1'(conv1'(y))) [M][P'] in one launch
    (grl_bottleneck_tail_bf16; resnets1.py:86-91 + :76-78 of the next block).  c1n None: y only.
    ``down`` / ``x0``: the residual is the block's downsample branch bnd(convd(x0)) (resnets1.py:83-84), computed in
    the same launch instead of being written by one launch and re-read by this one (``res`` is ignored)."""
    d = GrlBneckTail()
    y = _newb((M, c3.N), t2)
    d.t2, d.w3, d.scale3, d.shift3, d.res, d.y = ptr(t2), ptr(c3.wb()), ptr(c3.scale), ptr(c3.shift), ptr(res), ptr(y)
    d.M, d.P, d.C4, d.Pn = M, c3.K, c3.N, 0
    if down is not None:
        d.x0, d.wd, d.scaled, d.shiftd, d.Kd = ptr(x0), ptr(down.wb()), ptr(down.scale), ptr(down.shift), down.K
    u = None
    if c1n is not None:
        u = _newb((M, c1n.N), t2)
        d.w1n, d.scale1n, d.shift1n, d.u, d.Pn = ptr(c1n.wperm()), ptr(c1n.scale), ptr(c1n.shift), ptr(u), c1n.N
    check(_lib.load().grl_bottleneck_tail_bf16(C.byref(d), _lib.stream()), 'grl_bottleneck_tail_bf16')
    if _DEBUG_SYNC:
        _debug_sync('bneck_tail %s' % ((M, c3.K, c3.N, d.Pn),))
    return y, u


def _bneck_tail_f32_ok(c3, c1n, M):
    # (32-bit byte offsets inside the fused kernel: M * C4 * 4 bytes below 4 GiB, else one launch per convolution)
    return (_math[0] == MATH_F32 and c3.k == 1 and c1n.k == 1 and c1n.K == c3.N and M * c3.N * 4 < (1 << 32) and
            bool(_lib.load().grl_bottleneck_tail_f32_supported(c3.K, c3.N, c1n.N)))


def bneck_tail_f32(t2, c3, res, c1n, M):
    """The exact-fp32 twin (grl_bottleneck_tail_f32): bit-identical to conv3 (+res, ReLU) followed by conv1' on
    grl_conv_gemm_f32's one-chain fp32 datapath."""
    d = GrlBneckTailF32()
    y = _new((M, c3.N), t2)
    d.t2, d.w3, d.scale3, d.shift3, d.res, d.y = ptr(t2), ptr(c3.w), ptr(c3.scale), ptr(c3.shift), ptr(res), ptr(y)
    d.M, d.P, d.C4, d.Pn = M, c3.K, c3.N, 0
    u = None
    if c1n is not None:
        u = _new((M, c1n.N), t2)
        d.w1n, d.scale1n, d.shift1n, d.u, d.Pn = ptr(c1n.w), ptr(c1n.scale), ptr(c1n.shift), ptr(u), c1n.N
    check(_lib.load().grl_bottleneck_tail_f32(C.byref(d), _lib.stream()), 'grl_bottleneck_tail_f32')
    if _DEBUG_SYNC:
        _debug_sync('bneck_tail_f32 %s' % ((M, c3.K, c3.N, d.Pn),))
    return y, u


def _grl_eval_bf16s(model, inputs, taps=None, out_uncorr=None, ld_uncorr=2048):
    """Same launch order as _grl_eval with bf16 activations in HBM: stem -> trunk -> GCE ->
    TRL memo are bf16 tensors, every GEMM is the bf16-storage datapath, reductions land in
    fp32 vectors, the BN1d + L2 tail is the fp32 one."""
    plan = _plan(model, GrlEvalPlan)
    b, t, c, h, w = inputs.shape
    if (c, h, w) != (3, 256, 128):
        raise ValueError('GRL expects clips of [B,T,3,256,128] (got %s)' % (tuple(inputs.shape),))
    x = inputs.contiguous().view(b * t, c, h, w)
    n = b * t
    Hs, Ws = h // 2, w // 2
    H, W = (Hs + 1) // 2, (Ws + 1) // 2
    cur = _newb((n * H * W, 64), x)
    _stage('stem')
    if FUSE_STEM_POOL and taps is None and w == 128 and h % 4 == 0 and _stem_pool_ok(x, n):
        # stem + max-pool in one launch: the stem map never reaches HBM (grl_stem_pool_bf16)
        u8 = x.dtype == torch.uint8
        _call('grl_stem_pool_bf16', ptr(x), 1 if u8 else 0, ptr(input_mean_std(x.device)) if u8 else None,
              ptr(plan.stem_scale), ptr(plan.stem_shift), ptr(cur), n, h, w, ptr(plan.stem_wpb))
    else:
        stem = _newb((n * Hs * Ws, 64), x)
        if x.dtype == torch.uint8:
            _call('grl_stem_conv7x7_u8_bf16', ptr(x), ptr(input_mean_std(x.device)), ptr(plan.stem_w),
                  ptr(plan.stem_scale), ptr(plan.stem_shift), ptr(stem), n, h, w, 1, ptr(plan.stem_wpb))
        else:
            _call('grl_stem_conv7x7_bf16', ptr(x), ptr(plan.stem_w), ptr(plan.stem_scale), ptr(plan.stem_shift),
                  ptr(stem), n, h, w, 1, ptr(plan.stem_wpb))
        _call('grl_maxpool3x3s2_bf16', ptr(stem), ptr(cur), n, Hs, Ws, 64)
        del stem
    o1 = None
    for bi, e in enumerate(plan.blocks):
        if bi in (0, 3, 7, 13):
            _stage('layer%d' % ((0, 3, 7, 13).index(bi) + 1))
        s = e['stride']
        if o1 is None:
            o1, _, _ = _conv_b16(cur, e['c1'], n, H, W)
        o2, Ho, Wo = _conv_b16(o1, e['c2'], n, H, W, stride=s)
        nxt = plan.blocks[bi + 1]['c1'] if bi + 1 < len(plan.blocks) else None
        fuse = FUSE_BNECK and nxt is not None and _bneck_tail_ok(e['c3'], nxt, n * Ho * Wo)
        o1 = None
        if fuse and FUSE_DOWN and _bneck_down_ok(e['c3'], nxt, e['down'], s):
            # layer 1's first block: the downsample branch too -- its 4P-wide output is neither written nor re-read
            cur, o1 = bneck_tail_bf16(o2, e['c3'], None, nxt, n * Ho * Wo, down=e['down'], x0=cur)
        else:
            res = _conv_b16(cur, e['down'], n, H, W, stride=s, relu=False)[0] if e['down'] is not None else cur
            if fuse:
                # layers 1-2: conv3 + residual + ReLU AND the next block's conv1 in one launch -- the 4P-wide output is
                # written once (the next block's residual) and never re-read (fuse_bf16.hip)
                cur, o1 = bneck_tail_bf16(o2, e['c3'], res, nxt, n * Ho * Wo)
            else:
                cur, _, _ = _conv_b16(o2, e['c3'], n, Ho, Wo, res=res)
        H, W = Ho, Wo
    x4 = cur
    M = x4.shape[0]
    # GCE
    _stage('gce')
    x_glo = _new((b, 2048), x)
    _call('grl_group_mean_bf16', ptr(x4), ptr(x_glo), b, t * PIX, 2048, 2048, C.c_float(1.0), 0)
    g = plan.glo_fc
    glo = _new((b, 1024), x)
    gemm(x_glo, g.w, glo, b, 1024, 2048, scale=g.scale, shift=g.shift, relu=True, math=MATH_F32, kblock=True)
    c0 = plan.corr0
    gb = _new((b, 1024), x)
    gemm(glo, c0.w[:, 2048:], gb, b, 1024, 1024, ldw=3072, math=MATH_F32, kblock=True)
    h1 = _newb((M, 1024), x)
    gemm(x4, c0.wb(), h1, M, 1024, 2048, ldw=3072, gbias=gb, rows_per_group=t * PIX,
         scale=c0.scale, shift=c0.shift, relu=False, math=MATH_BF16S)
    c2 = plan.corr2
    h2 = _newb((M, 256), x)
    gemm(h1, c2.wb(), h2, M, 256, 1024, scale=c2.scale, shift=c2.shift, relu=True, math=MATH_BF16S)
    cmap = _new((M,), x)
    xc, xu = _newb((M, 2048), x), _newb((M, 2048), x)
    _call('grl_gce_gate_bf16', ptr(h2), ptr(plan.corr5_w), ptr(plan.corr6_scale), ptr(plan.corr6_shift),
          ptr(x4), ptr(cmap), ptr(xc), ptr(xu), M, 256, 2048)
    del x4, h1, h2
    if taps is not None:
        taps['corr_map'] = cmap.view(b * t, 1, 16, 8)
    # TRL
    _stage('trl')
    Cc, frame, Mb = 2048, PIX * 2048, b * PIX
    memo0 = _newb((Mb, Cc), x)
    _call('grl_temporal_mean_bf16', ptr(xu), ptr(memo0), b, t, frame)
    gapc = _new((b * t, Cc), x)
    _call('grl_group_mean_bf16', ptr(xc), ptr(gapc), b * t, PIX, Cc, Cc, C.c_float(1.0), 0)
    # bf16 storage: the large bf16 tiles own a CU's LDS, so the two directions' M = b * 128 GEMMs cannot share a CU and
    # each leaves half the chip idle.  Round 5: the two directions' conv1 / conv2 of a step (same shape, different
    # operands) go out as ONE grouped launch (grl_conv_gemm_f32_group: 256 x 128 tiles over both problems, bit-identical
    # to the separate launches) on the main stream, between two event hand-offs; everything else of a direction stays on
    # its own stream.  Every buffer that crosses streams is allocated on the main stream before the fork and lives until
    # the join.  MEASURED SLOWER in the pipeline (configs[2] same box 10.48 -> 10.68 ms; on ONE stream 10.78): alone the
    # grouped launch beats two launches (44 vs 2 x 35 us), but the 128 x 64 ring kernel the separate launches run on keeps
    # two workgroups per CU, so the two streams' launches already share every CU -- and the hand-offs cost their bubbles.
    # Off by default (GRL_TRL_GROUP=1 switches it on; kept tested: test_trl_grouped_launches_equal_two_stream_form).
    grouped = TRL_GROUP and len(plan.dirs) == 2
    fk = _TrlFork(x.device, taps is None and len(plan.dirs) == 2, att=not grouped)
    bufs = None
    if grouped:
        bufs = [dict(s=_newb((Mb, Cc), x), o=_newb((Mb, 512), x), o2=_newb((Mb, 512), x),
                     m=(_newb((Mb, Cc), x), _newb((Mb, Cc), x))) for _ in plan.dirs]
    fk.fork()
    f2, fc, scr = [], [], []
    for di, d in enumerate(plan.dirs):
        with fk.on(di):
            y = _newb((b * t * PIX, Cc), x)
            gemm(xc, d['f2'].wb(), y, b * t * PIX, Cc, Cc, shift=d['f2'].shift, relu=True, math=MATH_BF16S)
            f2.append(y)
            fc.append(_new((b, t, Cc), x) if fk.two else (torch.zeros((b, t, Cc), dtype=torch.float32, device=x.device) if di == 0 else fc[0]))
            scr.append((_new((b, Cc), x), _new((b, 128), x)))
    memo = [memo0, memo0]

    def f1_gemm(di, d, ti):
        if FUSE_TRL_SQDIFF and Mb % 256 == 0:
            # the squared difference reduced in the f1 GEMM's epilogue (32-row partial sums): conv_f1's output
            # never reaches HBM -- round 3: the bf16 256 x 256 kernel has the epilogue too
            dpart = _new((Mb // 32, Cc), x)
            gemm(memo[di], d['f1'].wb(), dpart, Mb, Cc, Cc, shift=d['f1'].shift, epilogue=EPI_SQDIFF,
                 res=f2[di][ti * PIX:], res_rows=PIX, res_gstride=t * PIX, math=MATH_BF16S)
            return dpart, None
        f1 = _newb((Mb, Cc), x)
        gemm(memo[di], d['f1'].wb(), f1, Mb, Cc, Cc, shift=d['f1'].shift, relu=True, math=MATH_BF16S)
        return None, f1

    def f1_tail(di, d, ti, dpart, f1):
        dvec, hid = scr[di]
        if dpart is not None:
            _call('grl_group_mean', ptr(dpart), ptr(dvec), b, PIX // 32, Cc, Cc, C.c_float(1.0 / 32.0), 0)
        else:
            _call('grl_sqdiff_mean_bf16', ptr(f1), ptr(f2[di][ti * PIX:]), ptr(dvec), b, PIX, Cc, t * frame)
        _call('grl_channel_atte', ptr(dvec), ptr(d['w1']), ptr(d['w2t']), ptr(gapc[ti:]), t * Cc,
              None, ptr(fc[di].view(b * t, Cc)[ti:]), t * Cc, 0 if fk.two else 1, b, Cc, d['w1'].shape[0], ptr(hid))

    def f1_branch(di, d, ti):
        f1_tail(di, d, ti, *f1_gemm(di, d, ti))

    for i in range(t):
        tis = [i, t - 1 - i]
        if grouped:
            for di, d in enumerate(plan.dirs):
                with fk.on(di):
                    f1_branch(di, d, tis[di])
                    _call('grl_add_strided_bf16', ptr(memo[di]), ptr(xu.view(-1)[tis[di] * frame:]), ptr(bufs[di]['s']),
                          b, frame, t * frame)
            fk.side_to_main()
            gemm_group([dict(a=bufs[di]['s'], w=d['c1'].wb(), y=bufs[di]['o'], M=Mb, N=512, K=Cc, scale=d['c1'].scale,
                             shift=d['c1'].shift, relu=True, math=MATH_BF16S) for di, d in enumerate(plan.dirs)])
            gemm_group([dict(a=bufs[di]['o'], w=d['c2'].wb(), y=bufs[di]['o2'], M=Mb, N=512, K=512, scale=d['c2'].scale,
                             shift=d['c2'].shift, relu=True, math=MATH_BF16S) for di, d in enumerate(plan.dirs)])
            fk.main_to_side()
            for di, d in enumerate(plan.dirs):
                with fk.on(di):
                    c3 = d['c3']
                    nm = bufs[di]['m'][i & 1]
                    gemm(bufs[di]['o2'], c3.wb(), nm, Mb, Cc, 512, scale=c3.scale, shift=c3.shift, res=bufs[di]['s'], relu=True,
                         math=MATH_BF16S)
                    memo[di] = nm
            continue
        for di, d in enumerate(plan.dirs):
            ti = tis[di]
            with fk.att_on(di, memo[di]):                       # the attention branch (its own stream: see _TrlFork)
                f1_branch(di, d, ti)
            with fk.on(di):
                s_ = _newb((Mb, Cc), x)
                _call('grl_add_strided_bf16', ptr(memo[di]), ptr(xu.view(-1)[ti * frame:]), ptr(s_), b, frame, t * frame)
                c1, c2_, c3 = d['c1'], d['c2'], d['c3']
                o = _newb((Mb, 512), x)
                gemm(s_, c1.wb(), o, Mb, 512, Cc, scale=c1.scale, shift=c1.shift, relu=True, math=MATH_BF16S)
                o2 = _newb((Mb, 512), x)
                gemm(o, c2_.wb(), o2, Mb, 512, 512, scale=c2_.scale, shift=c2_.shift, relu=True, math=MATH_BF16S)
                nm = _newb((Mb, Cc), x)
                gemm(o2, c3.wb(), nm, Mb, Cc, 512, scale=c3.scale, shift=c3.shift, res=s_, relu=True, math=MATH_BF16S)
                memo[di] = nm
    fk.join(memo[1], fc[1])
    fcorr = fc[0]
    if fk.two:
        fcorr = _new((b, t, Cc), x)
        _call('grl_add_strided', ptr(fc[0]), ptr(fc[1]), ptr(fcorr), 1, b * t * Cc, 0)
    f_uncorr = _new((b, Cc), x)
    _call('grl_group_mean_bf16', ptr(memo[0]), ptr(f_uncorr), b, PIX, Cc, Cc, C.c_float(1.0), 0)
    _call('grl_group_mean_bf16', ptr(memo[1]), ptr(f_uncorr), b, PIX, Cc, Cc, C.c_float(1.0), 1)
    if taps is not None:
        taps['f_uncorr'], taps['f_corr'] = f_uncorr, fcorr
    _stage('tail')
    x_corr = _new((b, t, 2048), inputs)
    _call('grl_affine_l2norm', ptr(fcorr), ptr(plan.corr_bn[0]), ptr(plan.corr_bn[1]), ptr(x_corr),
          b * t, 2048, 2048)
    x_uncorr = out_uncorr if out_uncorr is not None else _new((b, 2048), inputs)
    _call('grl_affine_l2norm', ptr(f_uncorr), ptr(plan.uncorr_bn[0]), ptr(plan.uncorr_bn[1]),
          ptr(x_uncorr), b, 2048, ld_uncorr)
    return x_uncorr, x_corr


def grl_forward(model, inputs, taps=None):
    """ResNet50_GRL_Model.forward.  eval(): folded-BN inference path.
    train(): batch-statistics forward recorded for the HIP backward."""
    require_device(inputs, 'inputs', allow_u8=True)
    if inputs.dim() != 5:
        raise ValueError('inputs must be [B,T,3,256,128]')
    if inputs.dtype == torch.uint8:
        inputs = rect_scale_u8(inputs)           # raw frames of another size: RectScale(256, 128) first
    if model.training:
        from . import train_engine
        return train_engine.grl_forward_train(model, normalize_u8(inputs))
    with torch.no_grad():
        return _grl_eval(model, inputs, taps)


# ----------------------------------------------------------------------------
# Siamese heads
# ----------------------------------------------------------------------------
def _attn_into(siam, x, out, ldy):
    plan = _plan(siam, SiameseEvalPlan)
    b, t, c = x.shape
    x = x.contiguous()
    qk = _new((b * t, 2 * plan.D), x)
    gemm(x, plan.wqk, qk, b * t, 2 * plan.D, c, scale=plan.scale, shift=plan.shift, kblock=_kb())
    _call('grl_siamese_attn', ptr(qk), ptr(x), ptr(out), b, t, plan.D, c, ldy)
    return out


def siamese_self_attention(siam, x):
    require_device(x, 'input')
    if siam.training:
        from . import train_engine
        return train_engine.siamese_self_attention_train(siam, x)
    with torch.no_grad():
        return _attn_into(siam, x, _new((x.shape[0], x.shape[2]), x), x.shape[2])


def siamese_forward(siam, x):
    require_device(x, 'input')
    from . import train_engine
    return train_engine.siamese_forward(siam, x)


def siamese_video_forward(siamv, x):
    require_device(x, 'input')
    from . import train_engine
    return train_engine.siamese_video_forward(siamv, x)


def extract_features(cnn, siam, clips):
    """attevaluator.py:100-112 in one pass: [b,T,3,256,128] -> [b,6144] =
    cat(x_uncorr, self_attention(x_corr), mean_T(x_corr)), each written straight
    into its slice of the feature row."""
    require_device(clips, 'clips', allow_u8=True)
    cnn = getattr(cnn, 'module', cnn)            # nn.DataParallel wrapper (mars_train.py:80)
    if cnn.training or siam.training:
        raise RuntimeError('extract_features needs cnn.eval() and siamese.eval()')
    with torch.no_grad():
        if clips.dtype == torch.uint8:
            clips = rect_scale_u8(clips)
        b, t = clips.shape[:2]
        feat = _new((b, 6144), clips)
        _, x_corr = _grl_eval(cnn, clips, out_uncorr=feat, ld_uncorr=6144)
        _attn_into(siam, x_corr, feat[:, 2048:], 6144)
        _call('grl_mean_T', ptr(x_corr), ptr(feat[:, 4096:]), b, t, 2048, 6144)
        return feat


def rows_mean(x):
    """[n, C] -> [1, C]: mean over rows (dense-mode clip average, attevaluator.py:96) through
    grl_group_mean."""
    require_device(x, 'features')
    x = x.contiguous()
    n, c = x.shape
    y = _new((1, c), x)
    _call('grl_group_mean', ptr(x), ptr(y), 1, n, c, c, C.c_float(1.0), 0)
    return y


class DevicePrefetcher(object):
    """Iterates a loader of (imgs, pids, camids) with the NEXT batch's host->device copy in flight
    on a side HIP stream while the current batch computes (pinned staging, non-blocking copy, an
    event hands the buffer to the compute stream).  uint8 batches stay uint8 (the stem normalises
    them, a quarter of the PCIe bytes); anything else is made float32 on the host, as
    `imgs.to(device)` upstream (attevaluator.py:70,76)."""

    def __init__(self, loader, device, depth=None, jpeg_size=(256, 128)):
        """``depth``: batches prepared ahead, each on its own side stream (default 1; 2 for loaders that hand over
        compressed frames).  ``jpeg_size``: the RectScale target (dataloader.py:53,68) that a compressed batch of MIXED
        frame sizes is brought to while it is decoded (jpeg.decode_jpeg_batch ``size``)."""
        import collections
        self.jpeg_size = jpeg_size
        self._ring = None
        self.it = iter(loader)
        self.dev = torch.device(device)
        self.depth = depth
        self.streams = []
        self._compressed = False
        self.queue = collections.deque()
        self._k = 0
        self._fill()

    def _stream(self):
        want = self.depth or 1
        while len(self.streams) < want:
            # NORMAL priority (GRL_PREFETCH_PRIORITY=-1: high, for A/B).  History: streams of one priority level share
            # GPU_MAX_HW_QUEUES (4) in-order hardware queues, and the first decoder's 10 ms entropy kernel on a normal-priority
            # prefetch stream sat in front of compute kernels of the SAME hardware queue (24.8 ms per eval step = 14.5 + 10.3);
            # high-priority streams have hardware queues of their own, which fixed that.  With the 0.8 ms decoder the
            # interference is gone -- and high priority has a cliff: a stream gets its hardware queue at FIRST USE, and when the
            # high-priority prefetch streams are used before the engine's side streams (a loader built before the first step:
            # the normal order), the normal-priority streams used afterwards serialise: bf16-storage train step 17.8 -> 32 ms,
            # fp32 53 -> 68, configs[2] eval 9.8 -> 12.2 (tools/jpeg_feed_order.py, profiles/r06_jpeg_feed_order.txt).
            pr = int(os.environ.get('GRL_PREFETCH_PRIORITY', '0')) if self._compressed else 0
            self.streams.append(torch.cuda.Stream(self.dev, priority=pr))
        self._k += 1
        return self.streams[self._k % want]

    def _fill(self):
        while len(self.queue) < (self.depth or 1):
            if not self._load():
                break

    def _load(self):
        try:
            imgs, pids, cams, *extra = next(self.it)   # extra: e.g. the augmentation parameter block
        except StopIteration:
            return False
        from grl_amd.reid.data.jpeg import JpegBatch, decode_jpeg_batch
        if isinstance(imgs, JpegBatch):
            # compressed frames (a loader with decode='device'): the bytes cross PCIe, grl_jpeg_decode_batch turns them
            # into the uint8 clip tensor on a prefetch stream, next to the current batch's compute (video_loader.py:124-141)
            if self.depth is None:
                self.depth = 2
            if not self._compressed:
                self._compressed, self.streams = True, []
            with torch.cuda.stream(self._stream()) as _:
                d = decode_jpeg_batch(imgs, self.dev, size=self.jpeg_size)
                ev = torch.cuda.Event()
                ev.record()
            self.queue.append((d, pids, cams, ev, None, extra))
            return True
        if imgs.dtype not in (torch.uint8, torch.float32):
            imgs = imgs.float()
        if imgs.is_cuda:
            self.queue.append((imgs, pids, cams, None, None, extra))
            return True
        host = imgs.contiguous()
        st = self._stream()
        with torch.cuda.stream(st):
            if host.is_pinned():                         # a loader with pin_memory=True (the reference's: dataloader.py:37-79)
                d = host.to(self.dev, non_blocking=True)
            else:
                # pageable batch: staged through a ring of reusable pinned buffers (no pinned allocation per batch) by ONE
                # host thread (below).  With `host.pin_memory()` here the bf16-storage training loop ran at 34.7 ms per
                # iteration on unpinned uint8 batches against 19.4 on pinned ones (tools/loop_rate.py)
                from grl_amd.reid.data.jpeg import _PinnedRing
                if self._ring is None:
                    self._ring = _PinnedRing()
                nbytes = host.numel() * host.element_size()
                slot, buf = self._ring.get(nbytes)
                stage = buf[:nbytes].view(host.dtype).view(host.shape)
                # ONE thread copies: `stage.copy_(host)` fans out over every OpenMP thread torch has (128 on the test boxes),
                # whose spin-waiting afterwards takes the cores from the thread that issues the step's ~1500 launches
                # (34.8 ms per bf16-storage iteration with copy_, the copy itself being 0.03 ms)
                C.memmove(stage.data_ptr(), host.data_ptr(), nbytes)
                d = stage.to(self.dev, non_blocking=True)
                self._ring.mark(slot)
                host = None
        ev = torch.cuda.Event()
        ev.record(st)
        self.queue.append((d, pids, cams, ev, host, extra))  # `host` kept alive until the copy is consumed
        return True

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        d, pids, cams, ev, _host, extra = self.queue.popleft()
        if ev is not None:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(ev)
            d.record_stream(cur)
        self._fill()
        return (d, pids, cams) + tuple(extra)


class GraphedExtractor(object):
    """`extract_features` captured once per input shape into a HIP graph (through
    torch.cuda.CUDAGraph: our launches go to torch's capturing stream) and replayed.
    One step is ~150 short launches; at small batches (the dense test_all.py mode feeds
    chunks of <= 8 clips, attevaluator.py:72-76) the host launch cost, not the GPU, bounds
    the eager path.  Output is bit-identical to the eager call (same kernels, same order).
    The packed-weight plans are built eagerly before capture and re-checked on every call:
    a parameter change drops the captured graphs."""

    def __init__(self, cnn, siam):
        self.cnn = getattr(cnn, 'module', cnn)
        self.siam = siam
        self._graphs = {}
        self._key = None

    def __call__(self, clips):
        require_device(clips, 'clips', allow_u8=True)
        key = (_state_key(self.cnn), _state_key(self.siam))
        if key != self._key:
            self._graphs.clear()
            self._key = key
        shape = tuple(clips.shape)
        g = self._graphs.get(shape)
        if g is None:
            static_in = clips.clone()
            for _ in range(2):                              # builds plans, warms the allocator
                extract_features(self.cnn, self.siam, static_in)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = extract_features(self.cnn, self.siam, static_in)
            g = (graph, static_in, static_out)
            self._graphs[shape] = g
        graph, static_in, static_out = g
        static_in.copy_(clips)
        graph.replay()
        return static_out.clone()


# ----------------------------------------------------------------------------
# evaluator distance matrices
# ----------------------------------------------------------------------------
def cosin_dist(qf, gf):
    """-qf . gf^T  (attevaluator.py:44-46) as one fp32 MFMA GEMM."""
    require_device(qf, 'qf'); require_device(gf, 'gf')
    qf, gf = qf.contiguous(), gf.contiguous()
    m, k = qf.shape
    n = gf.shape[0]
    out = _new((m, n), qf)
    return gemm(qf, gf, out, m, n, k, epilogue=EPI_NEGDOT, math=MATH_F32)


def pairwise_distance_tensor(x, y):
    """sqrt(clamp(|x|^2 + |y|^2 - 2 x.y^T, 1e-12))  (attevaluator.py:33-41)."""
    require_device(x, 'x'); require_device(y, 'y')
    m, n = x.shape[0], y.shape[0]
    x, y = x.contiguous().view(m, -1), y.contiguous().view(n, -1)
    k = x.shape[1]
    rn, cn = _new((m,), x), _new((n,), x)
    _call('grl_row_sqnorm', ptr(x), ptr(rn), m, k, k)
    _call('grl_row_sqnorm', ptr(y), ptr(cn), n, k, k)
    out = _new((m, n), x)
    return gemm(x, y, out, m, n, k, epilogue=EPI_EUCLID, rnorm=rn, cnorm=cn, math=MATH_F32)


def rank_rows(distmat):
    """Row-wise ascending argsort on the GPU (int32 [rows][n]); ties to the smaller index.
    Replaces the host `np.argsort(distmat, axis=1)` of eva_functions.py:139: one LDS bitonic
    network per row up to 16384 columns (MARS: 11310), the chunked network beyond (galleries of
    up to 2^24 entries; rows are processed in slabs so the workspace stays below ~1 GB)."""
    require_device(distmat, 'distmat')
    distmat = distmat.contiguous()
    rows, n = distmat.shape
    idx = torch.empty((rows, n), dtype=torch.int32, device=distmat.device)
    if n <= 16384:
        _call('grl_row_argsort', ptr(distmat), n, rows, n, ptr(idx))
        return idx
    lib = _lib.load()
    per_row = lib.grl_row_argsort_workspace_bytes(1, n)
    slab = max(1, min(rows, 65535, (1 << 30) // per_row))
    ws = torch.empty(lib.grl_row_argsort_workspace_bytes(slab, n), dtype=torch.uint8, device=distmat.device)
    for r0 in range(0, rows, slab):
        r = min(slab, rows - r0)
        _call('grl_row_argsort_wide', ptr(distmat[r0:]), n, r, n, ptr(idx[r0:]), ptr(ws))
    return idx


def rank_metrics(indices, q_pids, g_pids, q_camids, g_camids, max_rank=100):
    """CMC curve and mAP of eva_functions.evaluate (eva_functions.py:134-184) from a device
    argsort (``rank_rows``): `grl_rank_metrics` leaves (first match rank, #matches, AP) per query,
    the host only averages nq numbers.  Returns (cmc[max_rank] float32, mAP float)."""
    import numpy as np
    if not (torch.is_tensor(indices) and indices.is_cuda and indices.dtype == torch.int32):
        raise _lib.GrlHipError('rank_metrics needs the int32 device index matrix of rank_rows')
    indices = indices.contiguous()
    nq, ng = indices.shape
    dev = indices.device

    def ids(a, n, what):
        a = np.asarray(a).reshape(-1)
        if a.size != n:
            raise ValueError('%s: expected %d entries, got %d' % (what, n, a.size))
        return torch.from_numpy(a.astype(np.int32)).to(dev)
    qp, qc = ids(q_pids, nq, 'q_pids'), ids(q_camids, nq, 'q_camids')
    gp, gc = ids(g_pids, ng, 'g_pids'), ids(g_camids, ng, 'g_camids')
    first = torch.empty(nq, dtype=torch.int32, device=dev)
    nhit = torch.empty(nq, dtype=torch.int32, device=dev)
    ap = torch.empty(nq, dtype=torch.float64, device=dev)
    _call('grl_rank_metrics', ptr(indices), ng, ptr(qp), ptr(qc), ptr(gp), ptr(gc), nq, ng, ptr(first), ptr(nhit),
          ptr(ap))
    first, nhit, ap = first.cpu().numpy(), nhit.cpu().numpy(), ap.cpu().numpy()
    valid = nhit > 0
    assert valid.any(), "Error: all query identities do not appear in gallery"
    if ng < max_rank:
        max_rank = ng
        print("Note: number of gallery samples is quite small, got {}".format(ng))
    hit_by = (first[valid][:, None] <= np.arange(max_rank)[None, :]).astype(np.float32)
    return hit_by.sum(0) / float(valid.sum()), float(np.mean(ap[valid]))

