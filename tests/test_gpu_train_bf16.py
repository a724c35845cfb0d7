"""bf16-STORAGE training (train_engine.set_math('bf16s'), BASELINE configs[2] as a training batch) on a real MI355X.

Per kernel: every bf16-storage twin (train_bf16.hip, the bf16-in weight gradient, the statistics epilogue of the
bf16-storage GEMM) against its fp32 twin on bf16-REPRESENTABLE inputs -- the arithmetic is the same fp32 arithmetic, so
elementwise kernels must equal round_bf16(fp32 result) bit for bit and reductions agree to fp32 summation order.
End to end: the whole CNN forward + backward against the exact-fp32 step and against the reference's conditioned
fixture at a stated bf16 tolerance, and a 64 x 8 (configs[2]) step through size-independent properties."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope='module')
def dev():
    from grl_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _rep(*shape, seed=0, scale=1.0, shift=0.0):
    """bf16-representable fp32 tensor"""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).bfloat16().float()


def test_elementwise_twins_equal_rounded_fp32(dev):
    from grl_amd import train_engine as TE
    from grl_amd.engine import _call
    from grl_amd._lib import ptr
    M, Cc = 777, 256
    z, res, dy = (_rep(M, Cc, seed=s).to(dev) for s in (1, 2, 3))
    mean, scale, beta = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
    for relu, r in ((1, res), (0, None), (1, None)):
        y32 = torch.empty(M, Cc, device=dev); y16 = torch.empty(M, Cc, device=dev, dtype=BF)
        rb, zb = None if r is None else r.bfloat16(), z.bfloat16()       # (temporaries must outlive the launch)
        _call('grl_bn_apply_centered', ptr(z), ptr(mean), ptr(scale), ptr(beta), ptr(r), ptr(y32), M, Cc, relu, None)
        _call('grl_bn_apply_centered_bf16', ptr(zb), ptr(mean), ptr(scale), ptr(beta), ptr(rb), ptr(y16), M, Cc, relu, None)
        assert torch.equal(y16, y32.bfloat16())
    act = torch.relu(z)
    for acc in (0, 1):
        o32 = res.clone(); o16 = res.bfloat16()
        _call('grl_relu_bwd', ptr(dy), ptr(act), ptr(o32), dy.numel(), acc)
        dyb, actb = dy.bfloat16(), act.bfloat16()
        _call('grl_relu_bwd_bf16', ptr(dyb), ptr(actb), ptr(o16), dy.numel(), acc)
        assert torch.equal(o16, o32.bfloat16())
    o32 = torch.empty_like(z); o16 = torch.empty_like(z, dtype=BF)
    _call('grl_axpby', ptr(z), ptr(res), ptr(o32), C.c_float(1.0), C.c_float(1.0), z.numel())
    zb, resb = z.bfloat16(), res.bfloat16()
    _call('grl_axpby_bf16', ptr(zb), ptr(resb), ptr(o16), C.c_float(1.0), C.c_float(1.0), z.numel())
    assert torch.equal(o16, o32.bfloat16())
    # gate apply / backward (one wave per pixel row)
    Mg, Cg = 300, 2048
    x = _rep(Mg, Cg, seed=5).to(dev); y3 = _rep(Mg, 64, seed=6).to(dev)
    outs = []
    for b16 in (False, True):
        dt = BF if b16 else torch.float32
        cm = torch.empty(Mg, device=dev); xc = torch.empty(Mg, Cg, device=dev, dtype=dt); xu = torch.empty_like(xc)
        y3d, xd = y3.to(dt), x.to(dt)
        _call('grl_gate_apply' + ('_bf16' if b16 else ''), ptr(y3d), 64, ptr(xd), ptr(cm), ptr(xc), ptr(xu), Mg, Cg)
        outs.append((cm, xc, xu))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][1], outs[0][1].bfloat16()) and torch.equal(outs[1][2], outs[0][2].bfloat16())
    dxc, dxu = _rep(Mg, Cg, seed=7).to(dev), _rep(Mg, Cg, seed=8).to(dev)
    res = []
    for b16 in (False, True):
        dt = BF if b16 else torch.float32
        dx = torch.empty(Mg, Cg, device=dev, dtype=dt); dyy = torch.zeros(Mg, 64, device=dev, dtype=dt)
        dxcd, dxud, xd = dxc.to(dt), dxu.to(dt), x.to(dt)
        _call('grl_gate_bwd' + ('_bf16' if b16 else ''), ptr(dxcd), ptr(dxud), ptr(xd), ptr(outs[0][0]), ptr(dx), 0,
              ptr(dyy), 64, Mg, Cg)
        res.append((dx, dyy))
    assert torch.equal(res[1][0], res[0][0].bfloat16())
    assert float((res[1][1].float() - res[0][1]).abs().max() / res[0][1].abs().max()) < 1e-2     # (one bf16 rounding of a 2048-term sum)
    # dilate2 (all three modes), strided axpy, row broadcast, squared-difference backward
    n, Ho, Wo, Cd = 3, 4, 5, 64
    small = _rep(n * Ho * Wo, Cd, seed=9).to(dev)
    for acc, oy, ox in ((0, 0, 0), (1, 1, 0), (2, 1, 1)):
        up32 = _rep(n * 2 * Ho * 2 * Wo, Cd, seed=10).to(dev); up16 = up32.bfloat16()
        _call('grl_dilate2', ptr(small), ptr(up32), n, Ho, Wo, 2 * Ho, 2 * Wo, Cd, acc, oy, ox)
        smallb = small.bfloat16()
        _call('grl_dilate2_bf16', ptr(smallb), ptr(up16), n, Ho, Wo, 2 * Ho, 2 * Wo, Cd, acc, oy, ox)
        assert torch.equal(up16, up32.bfloat16())
    b, t, frame = 4, 3, 1024
    dst32 = _rep(b, t, frame, seed=11).to(dev); dst16 = dst32.bfloat16(); src = _rep(b, frame, seed=12).to(dev)
    _call('grl_axpy_strided', ptr(dst32.view(-1)[frame:]), t * frame, ptr(src), frame, b, frame, C.c_float(0.5), 1)
    srcb = src.bfloat16()
    _call('grl_axpy_strided_bf16', ptr(dst16.view(-1)[frame:]), t * frame, ptr(srcb), frame, b, frame, C.c_float(0.5), 1)
    assert torch.equal(dst16, dst32.bfloat16())
    v = torch.randn(b, 256, device=dev)
    d32 = _rep(b * 8, 256, seed=13).to(dev); d16 = d32.bfloat16()
    TE.add_rowbcast(d32, v, b * 8, 256, 8, 0.125, 1); TE.add_rowbcast(d16, v, b * 8, 256, 8, 0.125, 1)
    assert torch.equal(d16, d32.bfloat16())
    vb = _rep(b, 256, seed=14).to(dev)
    d32 = torch.empty(b * 8, 256, device=dev); d16 = torch.empty(b * 8, 256, device=dev, dtype=BF)
    TE.add_rowbcast(d32, vb, b * 8, 256, 8, 0.25, 0); TE.add_rowbcast(d16, vb.bfloat16(), b * 8, 256, 8, 0.25, 0)
    assert torch.equal(d16, d32.bfloat16())
    rows, Cs = 16, 128
    f1, f2 = _rep(b * rows, Cs, seed=15).to(dev), _rep(b * t * rows, Cs, seed=16).to(dev)
    dd = torch.randn(b, Cs, device=dev)
    r = []
    for b16 in (False, True):
        dt = BF if b16 else torch.float32
        df1 = torch.empty(b * rows, Cs, device=dev, dtype=dt); df2 = _rep(b * t * rows, Cs, seed=17).to(dev).to(dt)
        f1d, f2d = f1.to(dt), f2.to(dt)
        _call('grl_sqdiff_bwd' + ('_bf16' if b16 else ''), ptr(f1d), ptr(f2d[rows:]), ptr(dd), ptr(df1), ptr(df2[rows:]),
              b, rows, Cs, t * rows * Cs, 1)
        r.append((df1, df2))
    assert torch.equal(r[1][0], r[0][0].bfloat16()) and torch.equal(r[1][1], r[0][1].bfloat16())
    # bf16 -> fp32
    y = torch.empty(M, Cc, device=dev)
    zb = z.bfloat16()
    _call('grl_cast_f32', ptr(zb), ptr(y), z.numel())
    assert torch.equal(y, z)


@pytest.mark.parametrize('M,Cc', [(4096, 64), (1000, 128), (777, 512), (300, 2048), (2048, 32)])
def test_bn_backward_and_column_statistics_twins(dev, M, Cc):
    """grl_bn_bwd_bf16 (reduce, finalize, apply; activation mask, mask recomputed from z, residual gradient) and
    grl_col_stats_bf16 against the fp32 kernels on representable inputs."""
    from grl_amd import train_engine as TE
    from grl_amd.engine import _call
    from grl_amd._lib import ptr
    import torch.nn as nn
    g = torch.Generator().manual_seed(M + Cc)
    bn = nn.BatchNorm1d(Cc).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5); bn.bias.copy_(torch.randn(Cc, generator=g) * 0.3)
    z = _rep(M, Cc, seed=1, scale=2.0, shift=0.7).to(dev)
    dy = _rep(M, Cc, seed=2).to(dev)
    rows = TE._lib.load().grl_col_stats_rows(M)
    slab32 = torch.empty(rows, 2, Cc, device=dev); slab16 = torch.empty_like(slab32)
    _call('grl_col_stats', ptr(z), ptr(slab32), M, Cc, Cc, ptr(z))
    zb, piv = z.bfloat16(), z[0].contiguous()
    _call('grl_col_stats_bf16', ptr(zb), ptr(slab16), M, Cc, Cc, ptr(piv))
    assert float((slab32.sum(0) - slab16.sum(0)).abs().max() / slab32.sum(0).abs().max()) < 1e-5
    st = TE.bn_finalize(slab32, rows, Cc, M, bn, dev, pivot=z)
    a32 = torch.empty_like(z)
    TE.bn_apply(z, st, None, a32, M, Cc, True)
    for mode in ('act', 'from_z', 'none', 'gres', 'inplace'):
        outs = []
        for b16 in (False, True):
            dt = BF if b16 else torch.float32
            dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
            dyd = dy.to(dt).clone()
            gres = torch.empty(M, Cc, device=dev, dtype=dt) if mode == 'gres' else (dyd if mode == 'inplace' else None)
            act = a32.to(dt) if mode in ('act', 'gres', 'inplace') else None
            dz = TE.bn_backward(dyd, z.to(dt), act, st, bn.weight, dg, db, M, Cc, gres=gres, mask_from_z=(mode == 'from_z'))
            outs.append((dz.float(), dg, db, None if gres is None else gres.float()))
        if mode == 'gres':
            ref_gres = outs
        if mode == 'inplace':          # gres aliased to dy (masked in place) == the two-buffer form, bit for bit
            for o, r in zip(outs, ref_gres):
                assert all(torch.equal(x, y) for x, y in zip(o, r))
        (dz32, dg32, db32, gr32), (dz16, dg16, db16, gr16) = outs
        rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
        assert rel(dg16, dg32) < 2e-5 and rel(db16, db32) < 2e-5, mode
        assert rel(dz16, dz32) < 6e-3, mode                       # one bf16 rounding of dz
        if gr32 is not None:
            assert torch.equal(gr16, gr32.bfloat16().float())


@pytest.mark.parametrize('M,N,K,conv', [(4096, 128, 2048, None), (1000, 64, 64, None), (5000, 256, 64, None), (260, 2048, 128, None),
                                        (1024, 64, 160, None),
                                        (4096, 512, 2048, None), (5000, 264, 520, None), (33, 256, 256, None),     # (the 256 x 256 LDS-DMA tile)
                                        (4 * 16 * 8, 512, 9 * 512, (16, 8, 512, 16, 8, 3, 3, 1, 1)),
                                        (2 * 16 * 8, 128, 9 * 128, (16, 8, 128, 16, 8, 3, 3, 1, 1)),
                                        (3 * 16 * 8, 64, 9 * 64, (16, 8, 64, 16, 8, 3, 3, 1, 1)),
                                        (2 * 8 * 8, 256, 9 * 128, (16, 16, 128, 8, 8, 3, 3, 2, 1)),
                                        (2 * 8 * 4, 512, 256, (16, 8, 256, 8, 4, 1, 1, 2, 0))])
def test_wgrad_bf16_operands(dev, M, N, K, conv):
    """The bf16-in weight gradient (one 128 x 128 tile shape for every layer; tiles that straddle taps when C = 64;
    zero-filled edges) against the fp32 kernel on representable operands."""
    from grl_amd import train_engine as TE
    rng = np.random.default_rng(M + N + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    dz = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16().float().to(dev)
    x = torch.from_numpy(np.maximum(rng.standard_normal((rows_in, cin)), -0.5).astype(np.float32)).bfloat16().float().to(dev)
    kh = conv[5] if conv else 1

    def run(a, b):
        dw = torch.zeros((N, cin, kh, kh) if conv else (N, K), device=dev)
        TE.wgrad(a, b, dw, M, N, K, conv=conv, accumulate=0)
        return dw
    ref = run(dz, x).double()
    got = run(dz.bfloat16(), x.bfloat16()).double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 3e-6
    # accumulate into an existing gradient, and the stem's padded K (k_out)
    if conv is None and K == 160:
        dw = torch.ones(N, 147, device=dev)
        TE.wgrad(dz.bfloat16(), x.bfloat16(), dw, M, N, K, k_out=147, accumulate=1)
        assert float((dw.double() - 1.0 - ref[:, :147]).abs().max() / ref.abs().max()) < 3e-6


@pytest.mark.parametrize('M,N,K,conv', [(2048, 256, 256, None), (4096, 64, 576, (32, 16, 64, 32, 16, 3, 3, 1, 1)), (1000, 2048, 512, None),
                                        (16384, 64, 256, None),
                                        (49152 + 136, 256, 512, None), (65536, 512, 128, None),          # (the 256 x 256 kernel's epilogue)
                                        (48 * 32 * 32, 256, 9 * 64, (32, 32, 64, 32, 32, 3, 3, 1, 1))])
def test_gemm_bf16_storage_statistics_epilogue(dev, M, N, K, conv):
    """Train-mode BatchNorm statistics out of the bf16-storage GEMM: per-channel sum / sum of squares of the RAW
    fp32 accumulators (taken before the output is rounded to bf16) against the fp32 kernel's on representable
    operands; the stored output is the rounded accumulator."""
    from grl_amd import engine
    from grl_amd.engine import MATH_BF16S
    g = torch.Generator().manual_seed(M + N + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    a = (torch.randn(rows_in, cin, generator=g) + 0.3).bfloat16().float().to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16().float().to(dev)
    y32 = torch.empty(M, N, device=dev); y16 = torch.empty(M, N, device=dev, dtype=BF)
    _, s32 = engine.gemm(a, w, y32, M, N, K, stats=True, conv=conv, kblock=True)
    _, s16 = engine.gemm(a.bfloat16(), w.bfloat16(), y16, M, N, K, stats=True, conv=conv, math=MATH_BF16S)
    t32, t16 = s32.sum(0), s16.sum(0)
    assert float((t32[0] - t16[0]).abs().max() / t32[0].abs().max()) < 2e-5
    assert float((t32[1] - t16[1]).abs().max() / t32[1].abs().max()) < 2e-5
    assert float((y16.float() - y32).abs().max() / y32.abs().max()) < 5e-3


def _fresh(profile='conditioned'):
    import contextlib, io
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile=profile))
    return cnn.cuda().train()


@pytest.mark.parametrize('fname,med_tol,cos_min,cos_med', [('grl_train_cond_b8t4.npz', 0.2, 0.97, 0.99),
                                                            ('grl_train_cond_b4t8.npz', 0.3, 0.9, 0.97),
                                                            ('grl_train_cond_b32t4.npz', 0.15, 0.98, 0.99)])
def test_bf16_storage_step_against_reference_fixture(golden, fname, med_tol, cos_min, cos_med):
    """The whole CNN forward + backward in bf16 storage against the reference's fp32 run on the conditioned fixtures
    (8 x 4; 4 x 8 = the T = 8 recurrence of BASELINE configs[2]; the full 32 x 4).  STATED bf16 TOLERANCE: every stored activation carries one 2^-9
    rounding, ~50 train-mode layers deep: outputs within 3e-2 relative L2 (measured 1.6-1.8e-2); parameter
    gradients -- random projections of an L2-normalised, batch-normalised output, which amplify forward noise
    ~10x -- at cosine >= 0.97 for every >= 2-D weight and a median relative L2 error <= 0.2 (measured: median
    0.12-0.14, cosine median 0.993, min 0.987); BatchNorm running statistics within 5e-2 (the variance over the B = 8 rows of
    the pooled feature moves 2 % with the forward's 8e-3).  The exact-fp32 path keeps the 1e-3 pin.
    Measured per fixture (median relative L2 / min cosine / median cosine): 8 x 4: 0.107 / 0.981 / 0.995; 32 x 4: 0.084 /
    0.991 / 0.997; 4 x 8: 0.198 / 0.939 / 0.983 -- the per-clip BatchNorm1d layers normalise over FOUR rows there, which
    amplifies the forward's bf16 noise more (stated, looser bounds for that fixture)."""
    import train_cond_check as TC
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    g = golden(fname)
    B, T = int(g['meta.B']), int(g['meta.T'])
    clip_seed = int(g['meta.clip_seed']) if 'meta.clip_seed' in g.files else 3
    su = int(g['meta.xu_stride']) if 'meta.xu_stride' in g.files else 1
    sc = int(g['meta.xc_stride']) if 'meta.xc_stride' in g.files else 4
    cnn = _fresh()
    r1, r2 = TC.upstream(B, T)
    old = TE.set_math('bf16s')
    try:
        xu, xc = cnn(synth_clips_structured(B, T, seed=clip_seed).cuda())
        ((xu * r1.cuda()).sum() + (xc * r2.cuda()).sum()).backward()
    finally:
        TE.set_math(old)
    l2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    e_u, e_c = l2(xu.detach().cpu().numpy()[..., ::su], g['x_uncorr']), l2(xc.detach().cpu().numpy()[..., ::sc], g['x_corr_s4'])
    print('bf16s vs reference fp32: outputs rel L2 %.2e %.2e' % (e_u, e_c))
    assert e_u < 3e-2 and e_c < 3e-2
    errs, cosw = [], []
    named = dict(cnn.named_parameters())
    for k in [str(k) for k in g['meta.keys']]:
        f = named[k].grad.detach().reshape(-1).double()
        assert bool(torch.isfinite(f).all()), k
        idx = torch.linspace(0, f.numel() - 1, min(256, f.numel())).long().to(f.device)
        s = f[idx].cpu().numpy(); val = g['g.%s.val' % k].astype(np.float64)
        errs.append(np.linalg.norm(s - val) / max(np.linalg.norm(val), 1e-300))
        if named[k].dim() >= 2:
            cosw.append(float(np.dot(s, val) / (np.linalg.norm(s) * np.linalg.norm(val) + 1e-300)))
    errs = np.array(sorted(errs))
    print('bf16s gradients vs reference fp32: relative L2 median %.2e p90 %.2e; weight cosine min %.4f median %.4f' % (
        np.median(errs), errs[int(0.9 * len(errs))], min(cosw), float(np.median(cosw))))
    assert np.median(errs) < med_tol and min(cosw) > cos_min and np.median(cosw) > cos_med
    sd = cnn.state_dict()
    for k in [k for k in g.files if k.startswith('stat.') and 'num_batches' not in k]:
        assert np.abs(sd[k[5:]].double().cpu().numpy() - g[k]).max() / max(np.abs(g[k]).max(), 1e-30) < 5e-2, k
    for k in [k for k in g.files if k.startswith('stat.') and 'num_batches' in k]:
        assert int(sd[k[5:]]) == int(g[k])


# Per-tap budget of the bf16-storage TRAINING forward against the exact-fp32 forward on the same clips and weights
# (relative L2 of every stored activation the tape exposes).  One bf16 rounding is 2^-9 relative per element; every
# train-mode layer adds one, BatchNorm re-centres, ReLU keeps the error relative -- measured on MI355X (round 3,
# a per-tap probe): stem 2e-3, layer 4 9e-3, pooled features 6-8e-3.  The budgets are 2x those: a SINGLE layer
# that is wrong by a few percent (a dropped k-step, a misplaced scale) breaks its tap's budget and every later one, which
# the end-to-end "median gradient error 0.2" tolerance of the step test could absorb.
TAP_BUDGET = {'stem': 4e-3, 'pool': 4e-3, 'layer1': 8e-3, 'layer2': 1.2e-2, 'layer3': 1.6e-2, 'layer4': 2e-2,
              'x_glo': 1e-2, 'glo': 2e-2, 'corr_map': 2e-2, 'f_uncorr': 2e-2, 'f_corr': 2.5e-2, 'x_uncorr': 2e-2, 'x_corr': 2.5e-2}


def test_bf16_storage_forward_stays_inside_a_per_tap_budget():
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    dev = torch.device('cuda:0')
    clips = synth_clips_structured(8, 4, seed=3).to(dev)
    taps = {}
    for math in ('f32', 'bf16s'):
        cnn = _fresh()
        cnn._grl_taps = {}
        old = TE.set_math(math)
        try:
            with torch.no_grad():
                xu, xc = cnn(clips)
        finally:
            TE.set_math(old)
        t = dict(cnn._grl_taps)
        t['x_uncorr'], t['x_corr'] = xu, xc
        taps[math] = {k: (v.float() if torch.is_tensor(v) else torch.stack([w.float() for w in v])) for k, v in t.items()}
    seen = []
    for k, a in taps['f32'].items():
        b = taps['bf16s'][k].double()
        e = float((a.double() - b).norm() / a.double().norm())
        budget = TAP_BUDGET.get(k, 3e-2)
        seen.append((k, e, budget))
        print('bf16s tap %-12s rel L2 %.2e (budget %.1e)' % (k, e, budget))
    assert all(k in dict((s[0], 1) for s in seen) for k in ('stem', 'layer1', 'layer2', 'layer3', 'layer4'))
    bad = [s for s in seen if s[1] > s[2]]
    assert not bad, bad


def test_bf16_storage_step_at_configs2_size_properties():
    """BASELINE configs[2] as a training batch: P x K = 16 x 4 = 64 clips, T = 8, bf16 storage.  Size-independent
    properties: finite outputs with unit-norm rows, every parameter that gets a gradient in fp32 gets a finite one,
    BatchNorm bookkeeping (the TRL memo BatchNorms ran 8 times), the backward is exactly linear in the upstream
    gradient for powers of two (every bf16 / fp32 rounding commutes with a scaling by 2), and run-to-run determinism."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips
    B, T = 64, 8
    cnn = _fresh('default')
    clips = synth_clips(B, T, seed=4).cuda()
    g = torch.Generator().manual_seed(2)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    nb0 = int(cnn.temporal_learning_block.uncorr_memo_forward.bn1.num_batches_tracked)
    old = TE.set_math('bf16s')
    grads = []
    try:
        for scale in (1.0, 2.0, 1.0):
            cnn.zero_grad(set_to_none=True)
            xu, xc = cnn(clips)
            ((xu * r1).sum() * scale + (xc * r2).sum() * scale).backward()
            grads.append({k: p.grad.clone() for k, p in cnn.named_parameters() if p.grad is not None})
    finally:
        TE.set_math(old)
    assert bool(torch.isfinite(xu).all()) and bool(torch.isfinite(xc).all())
    assert float((xu.norm(dim=1) - 1).abs().max()) < 1e-4 and float((xc.norm(dim=2) - 1).abs().max()) < 1e-4
    assert int(cnn.temporal_learning_block.uncorr_memo_forward.bn1.num_batches_tracked) == nb0 + 3 * T
    assert len(grads[0]) >= 194 and all(bool(torch.isfinite(v).all()) for v in grads[0].values())
    # (running statistics move between the passes, batch statistics do not: the passes see the same forward)
    for k in grads[0]:
        assert torch.equal(grads[1][k], grads[0][k] * 2), k
        assert torch.equal(grads[2][k], grads[0][k]), k


@pytest.mark.parametrize('math', ['bf16s', 'f32'])
def test_graphed_train_step_equals_eager_bit_for_bit(math):
    """grl_amd.train_graph.GraphedTrainStep: the whole SEQTrainer step (forward, 5-term loss, HIP backward on three
    streams, fused SGD) captured into a HIP graph.  Three replays on changing batches leave the parameters, BatchNorm
    statistics and OIM tables bit-identical to three eager steps from the same start."""
    import contextlib, io
    from grl_amd import train_engine as TE
    from grl_amd.train_graph import GraphedTrainStep
    from grl_amd.reid import models
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.synthetic import synth_state_dict, synth_clips_structured
    dev = torch.device('cuda:0')
    B, T = 8, 4

    def build():
        with contextlib.redirect_stdout(io.StringIO()):
            cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
        siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
        siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
        cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'))
        siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
        siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
        mods = [m.to(dev).train() for m in (cnn, siam, siamv)]
        crits = [OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev) for _ in range(2)]
        tr = SEQTrainer(mods[0], mods[1], mods[2], PairLoss().to(dev), crits[0], crits[1], None)
        opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
        return tr, opt, mods, crits

    batches = [(synth_clips_structured(B, T, seed=60 + i).to(dev), (torch.tensor([5, 5, 9, 9, 300, 300, 77, 77]) + i).to(dev))
               for i in range(5)]

    def snap(mods, crits):
        return [v.detach().clone() for m in mods for v in m.state_dict().values()] + [c.lut.detach().clone() for c in crits]
    old = TE.set_math(math)
    try:
        tr, opt, mods, crits = build()                      # eager: 2 warm-up steps + 3 steps
        losses_e = []
        for i, (c, p) in enumerate(batches):
            out = tr._forward([c], p, 0, 0)
            opt.zero_grad(set_to_none=True); out[0].backward(); opt.step()
            losses_e.append(float(out[0].detach()))
        ref = snap(mods, crits)
        tr, opt, mods, crits = build()                      # graphed: two real warm-up steps inside the constructor, then 3 replays
        step = GraphedTrainStep(tr, opt, batches[2][0], batches[2][1], warmup_batches=batches[:2])
        for i in (2, 3, 4):
            out = step(*batches[i])
            assert float(out[0].detach()) == losses_e[i], (i, float(out[0].detach()), losses_e[i])
        got = snap(mods, crits)
    finally:
        TE.set_math(old)
    assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref))


@pytest.mark.parametrize('math', ['f32', 'bf16s'])
def test_weight_prep_table_equals_per_layer_path(math):
    """train_engine.WeightPrep: from the second step on every packed / transposed / bf16-cast weight of the step comes
    out of ONE grl_weight_prep launch (a table of gathers logged during the first step).  Three steps with it leave
    parameters and gradients bit-identical to three steps on the per-layer path, and the table covers the step."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 4, 4
    clips = [synth_clips_structured(B, T, seed=70 + i).cuda() for i in range(3)]
    g = torch.Generator().manual_seed(3)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    outs = []
    old = TE.set_math(math)
    try:
        for prep in (True, False):
            TE.WEIGHT_PREP = prep
            cnn = _fresh()
            opt = torch.optim.SGD(cnn.parameters(), lr=1e-2, momentum=0.9)
            for c in clips:
                xu, xc = cnn(c)
                opt.zero_grad(set_to_none=True)
                ((xu * r1).sum() + (xc * r2).sum()).backward()
                opt.step()
            outs.append(([p.detach().clone() for p in cnn.parameters()], [p.grad.clone() for p in cnn.parameters() if p.grad is not None]))
            if prep:
                wp = cnn.__dict__['_grl_weight_prep'][math == 'bf16s']
                assert wp.ready and wp.count >= (120 if math == "bf16s" else 80), wp.count
    finally:
        TE.WEIGHT_PREP = True
        TE.set_math(old)
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["f32", "bf16s"])
def test_fused_stem_tail_equals_separate_passes(math):
    """train_engine.STEM_TAIL_FUSED: BatchNorm apply + ReLU + max-pool of the stem in one pass that records the windows'
    first-maximum positions (the post-ReLU map is never written), max-pool backward through the positions, ReLU mask
    recomputed from z -- against bn_apply + grl_maxpool3x3s2 / grl_maxpool3x3s2_bwd on the stored activation: outputs,
    every parameter gradient and the running statistics bit-identical over two steps (odd map sizes included at the
    kernel level below)."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 4, 4
    clips = [synth_clips_structured(B, T, seed=90 + i).cuda() for i in range(2)]
    g = torch.Generator().manual_seed(6)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    outs = []
    old = TE.set_math(math)
    try:
        for fused in (True, False):
            TE.STEM_TAIL_FUSED = fused
            cnn = _fresh()
            opt = torch.optim.SGD(cnn.parameters(), lr=1e-2, momentum=0.9)
            keep = []
            for c in clips:
                xu, xc = cnn(c)
                opt.zero_grad(set_to_none=True)
                ((xu * r1).sum() + (xc * r2).sum()).backward()
                keep += [xu.detach().clone(), xc.detach().clone()]
                opt.step()
            outs.append(keep + [v.detach().clone() for v in cnn.state_dict().values()] +
                        [p.grad.clone() for p in cnn.parameters() if p.grad is not None])
    finally:
        TE.STEM_TAIL_FUSED = True
        TE.set_math(old)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("b16", [False, True])
@pytest.mark.parametrize("n,H,W", [(2, 9, 7), (3, 16, 8), (1, 5, 12)])
def test_bn_relu_maxpool_with_positions_matches_torch(b16, n, H, W):
    """grl_bn_relu_maxpool3x3s2(_bf16) + grl_maxpool3x3s2_bwd_idx(_bf16) against torch (BatchNorm apply -> ReLU ->
    MaxPool2d(3, 2, 1) and its autograd) on odd map sizes: pooled values equal, the routed gradient equals torch's
    max-pool backward (ties among zeros aside: their gradient is masked by the ReLU anyway, so compare after the mask)."""
    from grl_amd import _lib
    from grl_amd._lib import ptr
    lib = _lib.load()
    dev = torch.device('cuda:0')
    Cc = 64
    g = torch.Generator().manual_seed(H * 100 + W)
    dt = torch.bfloat16 if b16 else torch.float32
    sfx = '_bf16' if b16 else ''
    z = torch.randn(n * H * W, Cc, generator=g).to(dev).to(dt)
    mean, scale, beta = (torch.randn(Cc, generator=g) * 0.1).to(dev), (torch.rand(Cc, generator=g) + 0.5).to(dev), (torch.randn(Cc, generator=g) * 0.2).to(dev)
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    y = torch.empty(n * Hp * Wp, Cc, device=dev, dtype=dt)
    idx = torch.empty(n * Hp * Wp, Cc, device=dev, dtype=torch.uint8)
    _lib.check(getattr(lib, 'grl_bn_relu_maxpool3x3s2' + sfx)(ptr(z), ptr(mean), ptr(scale), ptr(beta), ptr(y), ptr(idx), n, H, W, Cc, _lib.stream()))
    a = torch.relu((z.float() - mean) * scale + beta).to(dt).float()            # the stored activation of the separate passes
    a4 = a.view(n, H, W, Cc).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    p = torch.nn.functional.max_pool2d(a4, 3, 2, 1)
    assert torch.equal(y.float().view(n, Hp, Wp, Cc).permute(0, 3, 1, 2), p.detach())
    dp = torch.randn(n * Hp * Wp, Cc, generator=g).to(dev).to(dt)
    da = torch.empty(n * H * W, Cc, device=dev, dtype=dt)
    _lib.check(getattr(lib, 'grl_maxpool3x3s2_bwd_idx' + sfx)(ptr(idx), ptr(dp), ptr(da), n, H, W, Cc, _lib.stream()))
    ref = torch.autograd.grad(p, a4, dp.float().view(n, Hp, Wp, Cc).permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1).reshape(n * H * W, Cc)
    mask = (a > 0).float()
    got = da.float() * mask
    want = (ref * mask).to(dt).float() if b16 else ref * mask
    assert torch.allclose(got, want, rtol=0, atol=1e-2 if b16 else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["f32", "bf16s"])
def test_relu_mask_bits_equal_reading_the_activation(math):
    """train_engine.RELU_BITS: the residual BatchNorms (y = relu(bn(z) + res)) record (y > 0) as one bit per output in the
    forward apply pass and their backward reads those bytes instead of the activation (grl_bn_bwd's relu_bits, both the
    in-place and the accumulating form occur in a bottleneck stack): outputs, every gradient and the running statistics
    bit-identical to the act-reading path over two steps."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 4, 4
    clips = [synth_clips_structured(B, T, seed=95 + i).cuda() for i in range(2)]
    g = torch.Generator().manual_seed(7)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    outs = []
    old = TE.set_math(math)
    fused_was, TE.BN_REDUCE_FUSED = TE.BN_REDUCE_FUSED, False      # (the fused reduce needs the bits for the residual BatchNorms:
    try:                                                           #  with it on, the two runs would differ in summation order)
        for bits in (True, False):
            TE.RELU_BITS = bits
            cnn = _fresh()
            opt = torch.optim.SGD(cnn.parameters(), lr=1e-2, momentum=0.9)
            keep = []
            for c in clips:
                xu, xc = cnn(c)
                opt.zero_grad(set_to_none=True)
                ((xu * r1).sum() + (xc * r2).sum()).backward()
                keep += [xu.detach().clone(), xc.detach().clone()]
                opt.step()
            outs.append(keep + [v.detach().clone() for v in cnn.state_dict().values()] +
                        [p.grad.clone() for p in cnn.parameters() if p.grad is not None])
    finally:
        TE.RELU_BITS = True
        TE.BN_REDUCE_FUSED = fused_was
        TE.set_math(old)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["f32", "mixed"])
def test_bn_backward_reduce_in_the_gemm_epilogue(golden, math):
    """train_engine.BN_REDUCE_FUSED (GrlGemm.bn_z): the data-gradient GEMM that completes the gradient of a bottleneck
    activation masks it and leaves the BatchNorm backward's two column sums in its epilogue; grl_bn_bwd_finish does the
    rest.  Same mathematics as the separate reduce pass in another summation order: the forward is bit-identical,
    every parameter gradient agrees to 2e-4 relative L2 (fp32 sums of 16384-262144 signed terms in two orders: measured
    worst 3.5e-5, a BatchNorm bias; `mixed`: 1e-3, the split-bf16 products see differently rounded inputs), and the fused path is actually
    taken (the reduce kernel runs less often)."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 4, 4
    clip = synth_clips_structured(B, T, seed=97).cuda()
    g = torch.Generator().manual_seed(8)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    outs, fused_calls = [], []
    old = TE.set_math(math)
    orig = TE._call
    try:
        for fused in (True, False):
            TE.BN_REDUCE_FUSED = fused
            n = [0]

            def spy(name, *a, _n=n):
                if name == 'grl_bn_bwd_finish':
                    _n[0] += 1
                return orig(name, *a)
            TE._call = spy
            cnn = _fresh()
            xu, xc = cnn(clip)
            ((xu * r1).sum() + (xc * r2).sum()).backward()
            fused_calls.append(n[0])
            outs.append(([xu.detach().clone(), xc.detach().clone()],
                         {k: p.grad.clone() for k, p in cnn.named_parameters() if p.grad is not None}))
    finally:
        TE._call = orig
        TE.BN_REDUCE_FUSED = True
        TE.set_math(old)
    torch.cuda.synchronize()
    assert fused_calls[0] >= 30 and fused_calls[1] == 0, fused_calls     # 16 blocks x (bn1, bn2, bn3) minus stride-2 / last
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    worst = 0.0
    live = [str(k) for k in golden('grl_train_cond_b8t4.npz')['meta.keys']]     # (not the analytically-zero gradients:
    for k in live:                                                               #  rounding noise in either order)
        ga, gb = outs[0][1][k], outs[1][1][k]
        e = float((ga.double() - gb.double()).norm() / gb.double().norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < (1e-3 if math == 'mixed' else 2e-4), (k, e)
    print('fused BN reduce: %d BatchNorms, worst gradient deviation %.1e' % (fused_calls[0], worst))


@pytest.mark.gpu
def test_bn_backward_reduce_in_the_gemm_epilogue_bf16_storage():
    """Round 5: the fused BatchNorm-backward reduce on bf16 storage (GrlGemm.bn_z with GRL_MATH_BF16S -- the interior
    epilogue of the 128-row tiles and of gemm_bf16_256_kernel, grl_bn_bwd_finish_bf16) against the separate
    bn_bwd_reduce pass.  The forward is bit-identical.  The two backward passes are NOT each other's reference: the
    separate pass sums the gradient after it was rounded to bf16, the epilogue sums the fp32 value before the rounding
    -- on near-cancelling sums (the bn3 biases: 262144 signed terms, |sum| ~ 2e-3) the rounded terms' noise is the
    size of the sum itself.  Both are therefore held against the exact-fp32 step of the same inputs: the fused path
    must be at least as close to it as the separate pass (median and 90th percentile over all parameter tensors), and
    it must actually be taken."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 4, 4
    clip = synth_clips_structured(B, T, seed=97).cuda()
    g = torch.Generator().manual_seed(8)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    outs, fused_calls = {}, {}
    orig = TE._call
    try:
        for name, math, fused in (('f32', 'f32', True), ('fused', 'bf16s', True), ('separate', 'bf16s', False)):
            old = TE.set_math(math)
            TE.BN_REDUCE_FUSED_BF16 = fused
            n = [0]

            def spy(fn, *a, _n=n):
                if fn == 'grl_bn_bwd_finish_bf16':
                    _n[0] += 1
                return orig(fn, *a)
            TE._call = spy
            try:
                cnn = _fresh()
                xu, xc = cnn(clip)
                ((xu * r1).sum() + (xc * r2).sum()).backward()
            finally:
                TE.set_math(old)
            fused_calls[name] = n[0]
            outs[name] = ([xu.detach().clone(), xc.detach().clone()],
                          {k: p.grad.double() for k, p in cnn.named_parameters() if p.grad is not None})
    finally:
        TE._call = orig
        TE.BN_REDUCE_FUSED_BF16 = True
    torch.cuda.synchronize()
    assert fused_calls['fused'] >= 30 and fused_calls['separate'] == 0, fused_calls
    assert all(torch.equal(a, b) for a, b in zip(outs['fused'][0], outs['separate'][0]))
    ef, es = [], []
    for k, gt in outs['f32'][1].items():
        if float(gt.norm()) < 1e-12:
            continue
        ef.append(float((outs['fused'][1][k] - gt).norm() / gt.norm()))
        es.append(float((outs['separate'][1][k] - gt).norm() / gt.norm()))
    ef.sort(); es.sort()
    med = lambda v: v[len(v) // 2]
    p90 = lambda v: v[len(v) * 9 // 10]
    print('bf16s BN reduce vs the fp32 step: fused median %.2e p90 %.2e max %.2e | separate median %.2e p90 %.2e max %.2e (%d BatchNorms fused)' % (
        med(ef), p90(ef), ef[-1], med(es), p90(es), es[-1], fused_calls['fused']))
    assert med(ef) <= 1.05 * med(es) and p90(ef) <= 1.05 * p90(es)


@pytest.mark.gpu
def test_statistics_slabs_are_written_completely():
    """engine.gemm(stats=True) no longer zero-fills its slab (79 fill launches per bf16-storage training step): with
    every slab poisoned (engine.SLAB_CHECK) a training forward + backward in both storages must leave no poison --
    i.e. the row count grl_conv_gemm_f32_stat_rows reports is the row count the kernel that takes the launch writes."""
    from grl_amd import engine, train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    clip = synth_clips_structured(4, 4, seed=5).cuda()
    engine.SLAB_CHECK = True
    try:
        for math in ('bf16s', 'f32'):
            old = TE.set_math(math)
            try:
                cnn = _fresh()
                xu, xc = cnn(clip)
                (xu.sum() + xc.sum()).backward()
                assert bool(torch.isfinite(xu).all()) and bool(torch.isfinite(xc).all())
            finally:
                TE.set_math(old)
    finally:
        engine.SLAB_CHECK = False


@pytest.mark.gpu
@pytest.mark.parametrize("math,only_corr", [("f32", False), ("f32", True), ("bf16s", False)])
def test_trl_stacked_weight_gradients_equal_per_step_products(golden, math, only_corr):
    """train_engine.WGRAD_STACK (round 5): the TRL recurrences apply f1 and their bottleneck T times with the same weights;
    the operands of the T weight gradients are row blocks of one buffer and dW is one product over T * B * 128 rows.  Same
    sums in another order: the forward and every activation gradient are bit-identical (the stacks only move buffers),
    the parameter gradients agree to 2e-4 relative L2 (fp32 accumulation, bf16s too) -- and 48 + 48 launches of the
    weight-gradient kernels and their slab reductions (recurrence 24, channel-attention MLP 16 ... per T = 4) become 12 + 12.  ``only_corr``: only f_corr carries a gradient, so the
    LAST step's bottleneck of each direction gets none -- its block of the stack must count as zeros, not stall the product."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 4, 4
    clip = synth_clips_structured(B, T, seed=99).cuda()
    g = torch.Generator().manual_seed(9)
    r1, r2 = torch.randn(B, 2048, generator=g).cuda(), torch.randn(B, T, 2048, generator=g).cuda()
    outs, launches = [], []
    old = TE.set_math(math)
    orig = TE.wgrad
    try:
        for stacked in (True, False):
            TE.WGRAD_STACK = stacked
            n = [0]

            def spy(*a, _n=n, **kw):
                _n[0] += 1
                return orig(*a, **kw)
            TE.wgrad = spy
            cnn = _fresh()
            xu, xc = cnn(clip)
            ((xc * r2).sum() if only_corr else (xu * r1).sum() + (xc * r2).sum()).backward()
            launches.append(n[0])
            outs.append(([xu.detach().clone(), xc.detach().clone()],
                         {k: p.grad.clone() for k, p in cnn.named_parameters() if p.grad is not None}))
    finally:
        TE.wgrad = orig
        TE.WGRAD_STACK = True
        TE.set_math(old)
    torch.cuda.synchronize()
    assert launches[1] - launches[0] == 2 * (4 + 2) * (T - 1), launches      # per direction: f1 + three convs + the attention MLP's two layers
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    live = [str(k) for k in golden('grl_train_cond_b8t4.npz')['meta.keys']]
    worst = 0.0
    for k in live:
        ga, gb = outs[0][1][k], outs[1][1][k]
        if 'temporal_learning_block' not in k:
            assert torch.equal(ga, gb), k            # (everything upstream of the TRL sees the same activation gradients)
            continue
        e = float((ga.double() - gb.double()).norm() / gb.double().norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    print('stacked TRL weight gradients: %d -> %d launches, worst deviation %.1e' % (launches[1], launches[0], worst))
