"""Cross-layer fusion of the bf16-storage trunk (grl_amd/csrc/fuse_bf16.hip) on a real MI355X, through the C ABI:
conv3 + residual + ReLU and the next block's conv1 in one launch, against torch fp32 on the same bf16 inputs and against
the unfused pipeline (two launches of the bf16-storage GEMM)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    from grl_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


class _C(object):
    """Stand-in for engine._Conv (1x1 conv + folded BatchNorm) with seeded weights."""

    def __init__(self, N, K, g, dev):
        from grl_amd import engine
        self.N, self.K, self.k, self.ldw, self.cin, self.stride = N, K, 1, K, K, 1
        self.w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        self.scale = (torch.rand(N, generator=g) + 0.5).to(dev)
        self.shift = (torch.randn(N, generator=g) * 0.3).to(dev)
        self._wb = self._wperm = None
        self.wb = engine._Conv.wb.__get__(self)
        self.wperm = engine._Conv.wperm.__get__(self)


def _close_bf16(got, ref, ulps=1.0):
    """got (bf16 values as fp32) is `ref` (fp32) rounded to bf16, give or take fp32 summation-order noise: the error
    is at most `ulps` bf16 half-spacings of |ref| plus 1e-4 absolute (a sum of O(1) terms that cancels to ~0 still
    carries ~1e-6 of the terms' magnitude)."""
    err = (got - ref).abs()
    tol = ref.abs() * (ulps * 2.0 ** -8) + 1e-4
    bad = err > tol
    assert not bool(bad.any()), (float(err.max()), int(bad.sum()))


@pytest.mark.parametrize('P,C4,Pn', [(64, 256, 64), (64, 256, 128), (128, 512, 128), (128, 512, 256), (64, 256, 0), (128, 512, 0)])
@pytest.mark.parametrize('M', [256 * 9, 16 * 7 + 5, 70000])
def test_bottleneck_tail_matches_torch_and_the_unfused_launches(dev, P, C4, Pn, M):
    from grl_amd import engine
    g = torch.Generator().manual_seed(P + C4 + Pn + M)
    c3 = _C(C4, P, g, dev)
    c1 = _C(Pn, C4, g, dev) if Pn else None
    t2 = torch.randn(M, P, generator=g).clamp_min(0).to(dev).to(BF)
    res = torch.randn(M, C4, generator=g).to(dev).to(BF)
    y, u = engine.bneck_tail_bf16(t2, c3, res, c1, M)
    torch.cuda.synchronize()
    # torch fp32 on the same bf16 operands
    w3 = c3.wb().float()
    yr = torch.relu(t2.float() @ w3.t() * c3.scale + c3.shift + res.float())
    _close_bf16(y.float(), yr)                                     # the fp32 value rounded once
    assert float((y.float() - yr).abs().max() / yr.abs().max()) < 6e-3
    # the unfused launches of the bf16-storage GEMM: conv3 (+res, ReLU), then conv1 on ITS output
    y0 = torch.empty(M, C4, dtype=BF, device=dev)
    engine.gemm(t2, c3.wb(), y0, M, C4, P, scale=c3.scale, shift=c3.shift, res=res, relu=True, math=engine.MATH_BF16S)
    _close_bf16(y.float(), y0.float(), ulps=2.0)                   # two roundings of almost the same fp32 value
    frac_equal = float((y == y0).float().mean())
    assert frac_equal > 0.995, frac_equal                           # (different MFMA shape: the fp32 sums differ in the last bit)
    if Pn:
        ur = torch.relu(y.float() @ c1.wb().float().t() * c1.scale + c1.shift)      # from the kernel's own y (bf16)
        _close_bf16(u.float(), ur)
        assert float((u.float() - ur).abs().max() / ur.abs().max()) < 6e-3
        u0 = torch.empty(M, Pn, dtype=BF, device=dev)
        engine.gemm(y, c1.wb(), u0, M, Pn, C4, scale=c1.scale, shift=c1.shift, relu=True, math=engine.MATH_BF16S)
        _close_bf16(u.float(), u0.float(), ulps=2.0)
    else:
        assert u is None


@pytest.mark.parametrize('M', [256 * 7, 16 * 5 + 3, 50000])
def test_bottleneck_tail_with_the_downsample_branch_in_the_same_launch(dev, M):
    """Layer 1's first block: res = bn_d(conv_d(x0)) is computed inside the launch -- against the unfused sequence
    (downsample GEMM writes bf16 res, the tail reads it): same roundings, so the same values up to fp32 summation order."""
    from grl_amd import engine
    g = torch.Generator().manual_seed(M)
    c3, c1, dn = _C(256, 64, g, dev), _C(64, 256, g, dev), _C(256, 64, g, dev)
    t2 = torch.randn(M, 64, generator=g).clamp_min(0).to(dev).to(BF)
    x0 = torch.randn(M, 64, generator=g).clamp_min(0).to(dev).to(BF)
    y, u = engine.bneck_tail_bf16(t2, c3, None, c1, M, down=dn, x0=x0)
    res = torch.empty(M, 256, dtype=BF, device=dev)
    engine.gemm(x0, dn.wb(), res, M, 256, 64, scale=dn.scale, shift=dn.shift, relu=False, math=engine.MATH_BF16S)
    y0, u0 = engine.bneck_tail_bf16(t2, c3, res, c1, M)
    # the residual is rounded to bf16 in both paths; where its fp32 value sits on a rounding boundary the two paths may
    # round it apart by one bf16 step OF THE RESIDUAL -- which can exceed a step of y where the sum cancels
    err = (y.float() - y0.float()).abs()
    assert bool((err <= (y0.float().abs() + res.float().abs()) * 2.0 ** -7 + 1e-4).all()), float(err.max())
    assert float((y == y0).float().mean()) > 0.99
    assert float((u.float() - u0.float()).norm() / u0.float().norm()) < 1e-3
    resr = ((x0.float() @ dn.wb().float().t()) * dn.scale + dn.shift)
    yr = torch.relu(t2.float() @ c3.wb().float().t() * c3.scale + c3.shift + resr)
    assert float((y.float() - yr).norm() / yr.norm()) < 4e-3


def test_bottleneck_tail_rows_do_not_depend_on_the_batch(dev):
    """A pixel's outputs are the same bits whatever tile / grid / batch it is computed in (each wave owns whole
    pixels): rows [0, 300) of a 5000-row launch equal a 300-row launch."""
    from grl_amd import engine
    g = torch.Generator().manual_seed(3)
    c3, c1 = _C(512, 128, g, dev), _C(128, 512, g, dev)
    t2 = torch.randn(5000, 128, generator=g).clamp_min(0).to(dev).to(BF)
    res = torch.randn(5000, 512, generator=g).to(dev).to(BF)
    y, u = engine.bneck_tail_bf16(t2, c3, res, c1, 5000)
    ys, us = engine.bneck_tail_bf16(t2[:300].contiguous(), c3, res[:300].contiguous(), c1, 300)
    assert torch.equal(y[:300], ys) and torch.equal(u[:300], us)


def test_bf16s_eval_with_and_without_the_fused_trunk(dev, synth_models):
    """End to end: the bf16-storage feature rows with the fused trunk agree with the one-launch-per-conv pipeline to
    bf16 rounding noise (well inside the 3e-2 bf16-storage tolerance vs the fp32 oracle) and stay batch independent."""
    from grl_amd import engine
    from grl_amd.synthetic import synth_clips
    cnn, siam = synth_models[0], synth_models[1]
    cnn, siam = cnn.to(dev).eval(), siam.to(dev).eval()
    clips = synth_clips(4, 4, seed=5).to(dev)
    with engine.math_mode('bf16s'):
        f1 = engine.extract_features(cnn, siam, clips)
        f1b = engine.extract_features(cnn, siam, clips[:2].contiguous())
        old, engine.FUSE_BNECK = engine.FUSE_BNECK, False
        try:
            f0 = engine.extract_features(cnn, siam, clips)
        finally:
            engine.FUSE_BNECK = old
    assert torch.equal(f1[:2], f1b)
    rel = float((f1 - f0).norm() / f0.norm())
    assert rel < 1e-2, rel


@pytest.mark.parametrize('P,C4,Pn', [(64, 256, 64), (64, 256, 128), (128, 512, 128), (64, 256, 0), (128, 512, 0)])
@pytest.mark.parametrize('M', [256 * 5, 32 * 3 + 7, 40000])
def test_bottleneck_tail_f32_is_bit_identical_to_the_two_gemm_launches(dev, P, C4, Pn, M):
    """Exact-fp32 twin (fuse_f32.hip): the transposed MFMA keeps gemm_f32_kernel's documented k-ordered fmaf chain,
    so y and u equal the unfused launches BIT FOR BIT (ragged M, every shape)."""
    from grl_amd import engine
    g = torch.Generator().manual_seed(P + C4 + Pn + M)
    c3 = _C(C4, P, g, dev)
    c1 = _C(Pn, C4, g, dev) if Pn else None
    t2 = torch.randn(M, P, generator=g).clamp_min(0).to(dev)
    res = torch.randn(M, C4, generator=g).to(dev)
    with engine.math_mode('f32'):
        y, u = engine.bneck_tail_f32(t2, c3, res, c1, M)
        y0 = torch.empty(M, C4, device=dev)
        engine.gemm(t2, c3.w, y0, M, C4, P, scale=c3.scale, shift=c3.shift, res=res, relu=True)
        assert torch.equal(y, y0), float((y - y0).abs().max())
        if Pn:
            u0 = torch.empty(M, Pn, device=dev)
            engine.gemm(y0, c1.w, u0, M, Pn, C4, scale=c1.scale, shift=c1.shift, relu=True)
            assert torch.equal(u, u0), float((u - u0).abs().max())
    yr = torch.relu((t2.double() @ c3.w.double().t()) * c3.scale.double() + c3.shift.double() + res.double())
    assert float((y.double() - yr).abs().max() / yr.abs().max()) < 1e-5


def test_f32_eval_features_identical_with_and_without_the_fused_trunk(dev, synth_models):
    """End to end, exact fp32 (the headline datapath): the 6144-d feature rows do not change by one bit."""
    from grl_amd import engine
    from grl_amd.synthetic import synth_clips
    cnn, siam = synth_models[0].to(dev).eval(), synth_models[1].to(dev).eval()
    clips = synth_clips(3, 4, seed=7).to(dev)
    with engine.math_mode('f32'):
        f1 = engine.extract_features(cnn, siam, clips)
        old, engine.FUSE_BNECK = engine.FUSE_BNECK, False
        try:
            f0 = engine.extract_features(cnn, siam, clips)
        finally:
            engine.FUSE_BNECK = old
    assert torch.equal(f1, f0)


@pytest.mark.parametrize('n,H', [(3, 64), (1, 8), (5, 16)])
def test_conv3x3_c64_matches_the_generic_kernel_and_torch(dev, n, H):
    """Layer 1's 3x3 (64 -> 64, W = 32): the LDS-resident-weight kernel against the generic implicit-GEMM launch it
    replaces and against torch conv2d in fp32 on the same bf16 operands (zero padding at the frame borders, tiles of 8
    rows, frames of different heights)."""
    import torch.nn.functional as F
    from grl_amd import engine
    g = torch.Generator().manual_seed(n * 100 + H)
    W = 32
    w4 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(dev)
    wp = torch.empty(64, 576, device=dev)
    engine._call('grl_pack_conv_weight', engine.ptr(w4), engine.ptr(wp), 64, 64, 3, 3)
    c = _C(64, 576, g, dev)
    c.w, c.k, c.cin = wp, 3, 64
    x = torch.randn(n * H * W, 64, generator=g).to(dev).to(BF)
    old, engine.FUSE_C64 = engine.FUSE_C64, True
    try:
        y, _, _ = engine._conv_b16(x, c, n, H, W)
        engine.FUSE_C64 = False
        y0, _, _ = engine._conv_b16(x, c, n, H, W)
    finally:
        engine.FUSE_C64 = old
    xr = x.float().view(n, H, W, 64).permute(0, 3, 1, 2)
    wr = c.wb().float().view(64, 3, 3, 64).permute(0, 3, 1, 2)
    yr = torch.relu(F.conv2d(xr, wr, padding=1) * c.scale.view(1, -1, 1, 1) + c.shift.view(1, -1, 1, 1))
    yr = yr.permute(0, 2, 3, 1).reshape(-1, 64)
    _close_bf16(y.float(), yr, ulps=1.0)
    _close_bf16(y.float(), y0.float(), ulps=2.0)
    assert float((y == y0).float().mean()) > 0.99


@pytest.mark.parametrize('n,u8', [(3, False), (5, True), (40, False), (130, True), (512, False)])
def test_stem_with_the_max_pool_in_the_same_launch_is_bit_identical(dev, synth_models, n, u8):
    """grl_stem_pool_bf16 (stem 7x7/s2 + folded BN + ReLU + 3x3/s2 max-pool, one launch, the stem map never written)
    against the two launches it replaces: rounding commutes with max, so the pooled map is the same bits -- float and raw
    uint8 inputs, every strip (top strip without / lower strips with a warm-up tile).  Round 5: the launch is the second
    form (stem_pool2_b16_kernel: weights in registers, fragments straight from the staged rows, pooling on bf16 bit
    patterns); strips of 8 (n < 32), 16 / 32 / 64 and 128 (n = 512: one workgroup per frame) stem rows; GRL_STEM_POOL2=0 in
    the environment runs the first form through the same test."""
    from grl_amd import engine
    from grl_amd.synthetic import synth_clips
    cnn = synth_models[0].to(dev).eval()
    plan = engine._plan(cnn, engine.GrlEvalPlan)
    x = synth_clips(1, n, seed=11, raw=u8)[0].to(dev).contiguous()
    assert x.shape[0] == n
    H, W = 256, 128
    stem = torch.empty(n * 128 * 64, 64, dtype=BF, device=dev)
    if u8:
        engine._call('grl_stem_conv7x7_u8_bf16', engine.ptr(x), engine.ptr(engine.input_mean_std(dev)), engine.ptr(plan.stem_w),
                     engine.ptr(plan.stem_scale), engine.ptr(plan.stem_shift), engine.ptr(stem), n, H, W, 1, engine.ptr(plan.stem_wpb))
    else:
        engine._call('grl_stem_conv7x7_bf16', engine.ptr(x), engine.ptr(plan.stem_w), engine.ptr(plan.stem_scale),
                     engine.ptr(plan.stem_shift), engine.ptr(stem), n, H, W, 1, engine.ptr(plan.stem_wpb))
    ref = torch.empty(n * 64 * 32, 64, dtype=BF, device=dev)
    engine._call('grl_maxpool3x3s2_bf16', engine.ptr(stem), engine.ptr(ref), n, 128, 64, 64)
    got = torch.full((n * 64 * 32, 64), -1.0, dtype=BF, device=dev)
    engine._call('grl_stem_pool_bf16', engine.ptr(x), 1 if u8 else 0, engine.ptr(engine.input_mean_std(dev)) if u8 else None,
                 engine.ptr(plan.stem_scale), engine.ptr(plan.stem_shift), engine.ptr(got), n, H, W, engine.ptr(plan.stem_wpb))
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())


def test_fused_kernels_are_deterministic_run_to_run(dev):
    """No atomics, fixed summation orders, every wave owns whole pixels: two launches on the same inputs give the same bits
    (streaming variants with their chunk barriers and hidden LDS-DMA included)."""
    from grl_amd import engine
    g = torch.Generator().manual_seed(17)
    for (P, C4, Pn) in ((64, 256, 64), (128, 512, 128), (128, 512, 256)):
        c3, c1 = _C(C4, P, g, dev), _C(Pn, C4, g, dev)
        M = 256 * 40 + 37
        t2 = torch.randn(M, P, generator=g).clamp_min(0).to(dev).to(BF)
        res = torch.randn(M, C4, generator=g).to(dev).to(BF)
        outs = [engine.bneck_tail_bf16(t2, c3, res, c1, M) for _ in range(3)]
        assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
        if (P, C4, Pn) != (128, 512, 256):
            t2f, resf = t2.float(), res.float()
            with engine.math_mode('f32'):
                outs = [engine.bneck_tail_f32(t2f, c3, resf, c1, M) for _ in range(3)]
            assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
