"""Per-kernel parity on a real MI355X, through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from grl_amd.synthetic import synth_clips

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    from grl_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('M,N,K', [(256, 128, 64), (128, 64, 32), (300, 200, 96), (33, 1024, 2048),
                                   (1000, 70, 160), (4096, 512, 2048), (129, 129, 32)])
def test_gemm_bit_exact_vs_fma_chain(dev, M, N, K):
    """Dense GEMM equals the C oracle's k-ordered fmaf chain bit for bit (all tile
    configurations, ragged M/N edges)."""
    from grl_amd import engine
    from oracle.ref_c import chain_gemm
    rng = np.random.default_rng(M * 7 + N)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    y = torch.empty(M, N, device=dev)
    engine.gemm(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), y, M, N, K)
    got = y.cpu().numpy()
    ref = chain_gemm(a, w)
    assert _rel(got, a.astype(np.float64) @ w.astype(np.float64).T) < 1e-5
    assert np.array_equal(got, ref), 'max diff %g' % np.abs(got - ref).max()


@pytest.mark.parametrize('M,N,K', [(32, 1024, 2048), (64, 2048, 1024), (256, 32, 2048), (5, 128, 2048), (33, 68, 1568)])
def test_gemm_splitk_equals_one_workgroup_kblock(dev, M, N, K):
    """Skinny K-blocked GEMMs (per-clip linears of GCE / TRL / the Siamese heads in train mode) run their 512-k
    segments as separate workgroups when the caller hands scratch (GrlGemm.splitk_ws); the finish kernel adds the
    segments in the chain's own association, so the result -- epilogue included -- is the one-workgroup result bit
    for bit."""
    from grl_amd import engine
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) + 0.5).to(dev)
    w = torch.randn(N, K, generator=g).to(dev)
    sc, sh = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    lib = engine._lib.load()
    for kw in (dict(), dict(scale=sc, shift=sh, relu=True), dict(shift=sh, res=res), dict(scale=sc, res=res, relu=True)):
        outs = []
        for split in (True, False):
            engine.SPLITK = split
            try:
                y = torch.full((M, N), float('nan'), device=dev)
                engine.gemm(a, w, y, M, N, K, kblock=True, **kw)
                outs.append(y)
            finally:
                engine.SPLITK = True
        assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all()), kw.keys()
    d = engine.GrlGemm()
    d.M, d.N, d.K, d.kblock, d.math = M, N, K, 1, 0
    assert lib.grl_conv_gemm_f32_workspace_floats(C.byref(d)) == -(-K // 512) * M * N
    d.M = 4096
    assert lib.grl_conv_gemm_f32_workspace_floats(C.byref(d)) == 0


@pytest.mark.parametrize('M,N,K,conv', [(300, 200, 2048, None), (64, 130, 4608, None), (4096, 512, 1024, None),
                                        (2 * 16 * 8, 96, 9 * 128, (16, 8, 128, 16, 8, 3, 3, 1, 1)),
                                        (128, 64, 512, None), (32, 1024, 2048, None), (256, 32, 2048, None),
                                        (7, 100, 1024, None)])
def test_gemm_kblock_bit_exact_and_more_accurate(dev, M, N, K, conv):
    """GrlGemm.kblock (the train-mode forward's K-blocked accumulation: the chain cut every 512 k,
    segments summed in order) equals the C oracle's blocked chain bit for bit on every tile shape,
    dense and implicit-GEMM, and is closer to the exact product than the single chain on
    positive-mean operands (where a long fp32 chain drifts)."""
    from grl_amd import engine
    from oracle.ref_c import chain_gemm
    rng = np.random.default_rng(M + N + K)
    w = rng.standard_normal((N, K)).astype(np.float32)
    if conv is None:
        a = (rng.standard_normal((M, K)) + 1.5).astype(np.float32)
        cols = a
        xa = torch.from_numpy(a).to(dev)
    else:
        H, W, Cc = conv[0], conv[1], conv[2]
        n = M // (H * W)
        x = (rng.standard_normal((n, H, W, Cc)) + 1.5).astype(np.float32)
        xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        cols = np.stack([xp[:, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], 3).reshape(M, K)
        xa = torch.from_numpy(x).to(dev)
    wd = torch.from_numpy(w).to(dev)
    y1, y0 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    engine.gemm(xa, wd, y1, M, N, K, conv=conv, kblock=True)
    engine.gemm(xa, wd, y0, M, N, K, conv=conv)
    assert np.array_equal(y1.cpu().numpy(), chain_gemm(cols, w, kblock=True))
    assert np.array_equal(y0.cpu().numpy(), chain_gemm(cols, w))
    if K > 512:
        exact = cols.astype(np.float64) @ w.astype(np.float64).T
        e1 = np.linalg.norm(y1.cpu().numpy() - exact); e0 = np.linalg.norm(y0.cpu().numpy() - exact)
        assert e1 < e0, (e1, e0)
    else:
        assert torch.equal(y0, y1)


def test_gemm_epilogue_affine_residual_relu_gbias(dev):
    from grl_amd import engine
    rng = np.random.default_rng(3)
    M, N, K = 512, 192, 128
    a, w = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    gb = rng.standard_normal((M // 128, N)).astype(np.float32)
    rs = rng.uniform(0, 1, M).astype(np.float32)
    t = lambda x: torch.from_numpy(x).to(dev)
    y = torch.empty(M, N, device=dev)
    engine.gemm(t(a), t(w), y, M, N, K, scale=t(sc), shift=t(sh), res=t(res), gbias=t(gb),
                rows_per_group=128, rowscale=t(rs), relu=True)
    acc = a.astype(np.float64) @ w.T.astype(np.float64)
    ref = np.maximum((acc * rs[:, None] + np.repeat(gb, 128, 0)) * sc + sh + res, 0)
    assert _rel(y.cpu().numpy(), ref) < 1e-5
    # strided output (writes into a slice of a wider row) and strided W (ldw > K)
    wide = torch.zeros(M, N + 64, device=dev)
    wpad = torch.from_numpy(np.concatenate([rng.standard_normal((N, 32)).astype(np.float32), w], 1)).to(dev)
    engine.gemm(t(a), wpad[:, 32:], wide[:, 64:], M, N, K, ldw=K + 32, ldy=N + 64)
    assert _rel(wide[:, 64:].cpu().numpy(), acc) < 1e-5
    assert float(wide[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize('cin,cout,k,stride,H,W,n', [
    (64, 64, 3, 1, 16, 8, 3), (128, 128, 3, 2, 16, 16, 2), (256, 512, 1, 2, 8, 8, 4),
    (64, 256, 1, 1, 8, 4, 2), (512, 512, 3, 1, 16, 8, 2)])
def test_conv_implicit_gemm_vs_torch(dev, cin, cout, k, stride, H, W, n):
    """Implicit-GEMM conv (+ folded BN, ReLU, residual) vs F.conv2d on CPU."""
    from grl_amd import engine
    rng = np.random.default_rng(cin + cout + k)
    x = torch.from_numpy(rng.standard_normal((n, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    ref = F.conv2d(x, w, stride=stride, padding=k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    ref = F.relu(ref + res)
    Ho, Wo = ref.shape[2:]
    xl = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = torch.empty(cout, k * k * cin, device=dev)
    from grl_amd._lib import ptr
    wd = w.to(dev)
    engine._call('grl_pack_conv_weight', ptr(wd), ptr(wp), cout, cin, k, k)
    assert torch.equal(wp.cpu().view(cout, k * k, cin), w.view(cout, cin, k * k).permute(0, 2, 1))
    y = torch.empty(n * Ho * Wo, cout, device=dev)
    engine.gemm(xl, wp, y, n * Ho * Wo, cout, k * k * cin, scale=sc.to(dev), shift=sh.to(dev),
                res=res.permute(0, 2, 3, 1).contiguous().to(dev), relu=True,
                conv=(H, W, cin, Ho, Wo, k, k, stride, k // 2))
    got = y.view(n, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
    assert _rel(got.numpy(), ref.numpy()) < 1e-5


def test_gemm_train_stats_slab(dev):
    from grl_amd import engine, _lib
    rng = np.random.default_rng(5)
    M, N, K = 1000, 96, 64
    a, w = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    d = _lib.GrlGemm(); d.M, d.N, d.K = M, N, K
    rows = _lib.load().grl_conv_gemm_f32_stat_rows(C.byref(d))
    stats = torch.zeros(rows, 2, N, device=dev)
    y = torch.empty(M, N, device=dev)
    engine.gemm(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), y, M, N, K, stats=stats)
    yy = y.cpu().double().numpy()
    s = stats.sum(0).cpu().numpy()
    assert _rel(s[0], yy.sum(0)) < 1e-5 and _rel(s[1], (yy ** 2).sum(0)) < 1e-5


def test_stem_and_maxpool_vs_torch(dev):
    from grl_amd import engine
    from grl_amd._lib import ptr
    rng = np.random.default_rng(9)
    n, H, W = 2, 64, 32
    x = torch.from_numpy(rng.standard_normal((n, 3, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, 64).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(64).astype(np.float32) * 0.1)
    ref = F.relu(F.conv2d(x, w, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    y = torch.empty(n * (H // 2) * (W // 2), 64, device=dev)
    xd, wd, scd, shd = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)     # keep alive across the launch
    engine._call('grl_stem_conv7x7', ptr(xd), ptr(wd), ptr(scd), ptr(shd), ptr(y), n, H, W, 1, None)
    got = y.view(n, H // 2, W // 2, 64).permute(0, 3, 1, 2).cpu()
    assert _rel(got.numpy(), ref.numpy()) < 1e-5
    pooled = torch.empty(n * (H // 4) * (W // 4), 64, device=dev)
    engine._call('grl_maxpool3x3s2', ptr(y), ptr(pooled), n, H // 2, W // 2, 64)
    refp = F.max_pool2d(got, 3, stride=2, padding=1)
    assert torch.equal(pooled.view(n, H // 4, W // 4, 64).permute(0, 3, 1, 2).cpu(), refp)


@pytest.mark.parametrize('n,u8', [(2, False), (3, True), (40, False), (128, False)])
def test_stem_with_the_max_pool_in_one_launch_fp32(dev, n, u8):
    """grl_stem_pool_f32 (round 5: stem 7x7/s2 + folded BN + ReLU + 3x3/s2 max-pool in one launch, exact fp32 MFMA, the
    stem map never written) against torch (conv2d + max_pool2d in float64, 1e-5 like the stem itself) and against the two
    launches it replaces (same products, another fp32 summation order: 2e-6 of the map's scale) -- float and raw uint8
    input, strips of 4 / 8 / 16 pooled rows with their warm-up iteration (n = 128: the bench's 16-row strips)."""
    from grl_amd import engine
    from grl_amd._lib import ptr
    rng = np.random.default_rng(19)
    H, W = 256, 128
    if u8:
        xr = torch.from_numpy(rng.integers(0, 256, (n, 3, H, W), dtype=np.uint8))
        ms = engine.input_mean_std(dev)
        xf = ((xr.float() / 255.0) - ms[:3].cpu().view(1, 3, 1, 1)) / ms[3:].cpu().view(1, 3, 1, 1)
    else:
        xr = torch.from_numpy(rng.standard_normal((n, 3, H, W)).astype(np.float32))
        xf = xr
    w = torch.from_numpy((rng.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(-1.5, 1.5, 64).astype(np.float32))           # (negative scales too: max and BN do not commute)
    sh = torch.from_numpy(rng.standard_normal(64).astype(np.float32) * 0.1)
    xd, wd, scd, shd = xr.to(dev).contiguous(), w.to(dev), sc.to(dev), sh.to(dev)
    wq = torch.empty(64 * 168, device=dev)
    engine._call('grl_stem_pack_weight_pool', ptr(wd), ptr(wq))
    got = torch.full((n * 64 * 32, 64), -1.0, device=dev)
    engine._call('grl_stem_pool_f32', ptr(xd), 1 if u8 else 0, ptr(engine.input_mean_std(dev)) if u8 else None, ptr(scd), ptr(shd),
                 ptr(got), n, H, W, ptr(wq))
    stem = torch.empty(n * 128 * 64, 64, device=dev)
    if u8:
        engine._call('grl_stem_conv7x7_u8', ptr(xd), ptr(engine.input_mean_std(dev)), ptr(wd), ptr(scd), ptr(shd), ptr(stem), n, H, W, 1, None)
    else:
        engine._call('grl_stem_conv7x7', ptr(xd), ptr(wd), ptr(scd), ptr(shd), ptr(stem), n, H, W, 1, None)
    two = torch.empty(n * 64 * 32, 64, device=dev)
    engine._call('grl_maxpool3x3s2', ptr(stem), ptr(two), n, 128, 64, 64)
    scale = float(two.abs().max())
    assert float((got - two).abs().max()) < 2e-6 * scale, float((got - two).abs().max()) / scale
    m = min(n, 4)                                              # (torch reference on the first and last frames)
    for sl in (slice(0, m), slice(n - m, n)):
        ref = F.max_pool2d(F.relu(F.conv2d(xf[sl].double(), w.double(), stride=2, padding=3) * sc.double().view(1, -1, 1, 1) +
                                  sh.double().view(1, -1, 1, 1)), 3, stride=2, padding=1)
        g = got.view(n, 64, 32, 64)[sl].permute(0, 3, 1, 2).cpu().double()
        assert _rel(g.numpy(), ref.numpy()) < 1e-5


def test_pointwise_kernels_vs_torch(dev):
    from grl_amd import engine
    from grl_amd._lib import ptr
    rng = np.random.default_rng(11)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    b, T, P, Cc = 3, 4, 128, 2048
    x = t(b, T, P, Cc); xd = x.to(dev)
    # group mean (GAP over pixels, GAP over pixels and T, accumulate)
    y = torch.empty(b * T, Cc, device=dev)
    engine._call('grl_group_mean', ptr(xd), ptr(y), b * T, P, Cc, Cc, C.c_float(1.0), 0)
    assert _rel(y.cpu().numpy(), x.mean(2).view(b * T, Cc).numpy()) < 1e-5
    y2 = torch.empty(b, Cc, device=dev)
    engine._call('grl_group_mean', ptr(xd), ptr(y2), b, T * P, Cc, Cc, C.c_float(1.0), 0)
    engine._call('grl_group_mean', ptr(xd), ptr(y2), b, T * P, Cc, Cc, C.c_float(1.0), 1)
    assert _rel(y2.cpu().numpy(), 2 * x.mean((1, 2)).numpy()) < 1e-5
    # temporal mean / add_strided / sqdiff_mean / mean_T
    tm = torch.empty(b, P * Cc, device=dev)
    engine._call('grl_temporal_mean', ptr(xd), ptr(tm), b, T, P * Cc)
    assert _rel(tm.cpu().numpy(), x.mean(1).view(b, -1).numpy()) < 1e-6
    a = t(b, P, Cc); ad = a.to(dev)
    s = torch.empty(b, P * Cc, device=dev)
    engine._call('grl_add_strided', ptr(ad), ptr(xd.view(-1)[2 * P * Cc:]), ptr(s), b, P * Cc, T * P * Cc)
    assert torch.equal(s.cpu().view(b, P, Cc), a + x[:, 2])
    d = torch.empty(b, Cc, device=dev)
    engine._call('grl_sqdiff_mean', ptr(ad), ptr(xd.view(-1)[1 * P * Cc:]), ptr(d), b, P, Cc, T * P * Cc)
    assert _rel(d.cpu().numpy(), (a - x[:, 1]).pow(2).mean(1).numpy()) < 1e-5
    f = t(b, T, Cc); mt = torch.zeros(b, 3 * Cc, device=dev); fd = f.to(dev)
    engine._call('grl_mean_T', ptr(fd), ptr(mt[:, Cc:]), b, T, Cc, 3 * Cc)
    assert _rel(mt[:, Cc:2 * Cc].cpu().numpy(), f.mean(1).numpy()) < 1e-6 and float(mt[:, :Cc].abs().max()) == 0
    # GCE gate
    M = 37
    h, w3, xx = t(M, 256), t(256) * 0.1, t(M, Cc)
    bs, bh = torch.tensor([0.8]), torch.tensor([-0.1])
    cm, xc, xu = torch.empty(M, device=dev), torch.empty(M, Cc, device=dev), torch.empty(M, Cc, device=dev)
    hd, w3d, bsd, bhd, xxd = h.to(dev), w3.to(dev), bs.to(dev), bh.to(dev), xx.to(dev)
    engine._call('grl_gce_gate', ptr(hd), ptr(w3d), ptr(bsd), ptr(bhd), ptr(xxd),
                 ptr(cm), ptr(xc), ptr(xu), M, 256, Cc)
    g = torch.sigmoid((h @ w3) * 0.8 - 0.1)
    assert _rel(cm.cpu().numpy(), g.numpy()) < 1e-5
    assert _rel(xc.cpu().numpy(), (xx * g[:, None]).numpy()) < 1e-5
    assert _rel(xu.cpu().numpy(), (xx * (1 - g[:, None])).numpy()) < 1e-5
    # channel attention + f_step accumulate
    dv, w1, w2 = t(b, Cc).abs(), t(128, Cc) * 0.02, t(Cc, 128) * 0.1
    gap = t(b, T, Cc); fs = t(b, T, Cc); fsd = fs.clone().to(dev)
    ca = torch.empty(b, Cc, device=dev); hidw = torch.empty(b, 128, device=dev)
    dvd, w1d, w2td, gapd = dv.to(dev), w1.to(dev), w2.t().contiguous().to(dev), gap.to(dev)
    engine._call('grl_channel_atte', ptr(dvd), ptr(w1d), ptr(w2td),
                 ptr(gapd[:, 1]), T * Cc, ptr(ca), ptr(fsd.view(b * T, Cc)[2:]), T * Cc, 1, b, Cc, 128, ptr(hidw))
    cref = torch.sigmoid(F.relu(dv @ w1.t()) @ w2.t())
    assert _rel(ca.cpu().numpy(), cref.numpy()) < 1e-5
    fref = fs.clone(); fref[:, 2] += gap[:, 1] * cref + gap[:, 1]
    assert _rel(fsd.cpu().numpy(), fref.numpy()) < 1e-5
    # affine + l2norm into a strided destination, row_sqnorm
    v, sc, sh = t(5, Cc), t(Cc), t(Cc)
    out = torch.zeros(5, 3 * Cc, device=dev)
    vd, scd, shd = v.to(dev), sc.to(dev), sh.to(dev)
    engine._call('grl_affine_l2norm', ptr(vd), ptr(scd), ptr(shd), ptr(out[:, Cc:]), 5, Cc, 3 * Cc)
    assert _rel(out[:, Cc:2 * Cc].cpu().numpy(), F.normalize(v * sc + sh, dim=1).numpy()) < 1e-5
    rn = torch.empty(5, device=dev)
    engine._call('grl_row_sqnorm', ptr(vd), ptr(rn), 5, Cc, Cc)
    assert _rel(rn.cpu().numpy(), v.pow(2).sum(1).numpy()) < 1e-5
    # bn fold
    gm, bt, mu, var, bias = t(Cc).abs() + 0.5, t(Cc), t(Cc), t(Cc).abs() + 0.5, t(Cc)
    so, ho = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    dd = [z.to(dev) for z in (gm, bt, mu, var, bias)]
    engine._call('grl_bn_fold', ptr(dd[0]), ptr(dd[1]), ptr(dd[2]), ptr(dd[3]), ptr(dd[4]),
                 C.c_float(1e-5), ptr(so), ptr(ho), Cc)
    sref = gm / torch.sqrt(var + 1e-5)
    assert _rel(so.cpu().numpy(), sref.numpy()) < 1e-6
    assert _rel(ho.cpu().numpy(), (bt - mu * sref + bias * sref).numpy()) < 1e-5


def test_empty_and_bad_arguments_raise(dev):
    from grl_amd import engine
    from grl_amd._lib import GrlHipError, ptr
    x = torch.zeros(64, 48, device=dev)
    with pytest.raises(GrlHipError):          # K not a multiple of 32
        engine.gemm(x, x, torch.empty(64, 64, device=dev), 64, 64, 48)
    with pytest.raises(GrlHipError):          # empty
        engine.gemm(x, x, x, 0, 64, 32)
    with pytest.raises(GrlHipError):          # T > 16
        engine._call('grl_siamese_attn', ptr(x), ptr(x), ptr(x), 1, 17, 512, 2048, 2048)


@pytest.mark.parametrize('mode,tol', [('bf16x3', 3e-5), ('bf16', 2e-2)])
def test_gemm_bf16_math_modes(dev, mode, tol):
    """Alternative multiplier datapaths: split-bf16 is fp32-class (error relative to
    sum |a||b| ~ 2^-16), plain bf16 is the BASELINE configs[2] mode."""
    from grl_amd import engine, _lib
    rng = np.random.default_rng(17)
    for (M, N, K, conv) in ((512, 256, 2048, None), (300, 200, 96, None), (128 * 3, 128, 9 * 64, (16, 8, 64, 16, 8, 3, 3, 1, 1))):
        a = rng.standard_normal((M, K if conv is None else 64)).astype(np.float32)
        w = rng.standard_normal((N, K)).astype(np.float32)
        ad, wd = torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev)
        y = torch.empty(M, N, device=dev)
        y32 = torch.empty(M, N, device=dev)
        engine.gemm(ad, wd, y, M, N, K, conv=conv, math={'bf16x3': 3, 'bf16': 1}[mode])
        engine.gemm(ad, wd, y32, M, N, K, conv=conv, math=0)
        ref = y32.cpu().double().numpy()
        if conv is None:
            denom = (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T).max()
        else:
            denom = np.abs(ref).max() * 10
        assert np.abs(y.cpu().double().numpy() - ref).max() / denom < tol, (mode, M, N, K)


def test_row_argsort_matches_numpy(dev):
    """GPU row argsort == np.argsort(kind='stable'), including exact ties, ragged widths and
    the MARS gallery width."""
    from grl_amd import engine
    rng = np.random.default_rng(4)
    for rows, n in ((7, 1), (5, 37), (16, 1024), (9, 4097), (12, 11310)):
        d = rng.standard_normal((rows, n)).astype(np.float32)
        d[:, ::5] = np.round(d[:, ::5], 1)                  # plenty of exact ties
        got = engine.rank_rows(torch.from_numpy(d).to(dev)).cpu().numpy()
        assert np.array_equal(got, np.argsort(d, axis=1, kind='stable')), (rows, n)
    # beyond one LDS network (galleries > 16384): the chunked bitonic network, same stable order;
    # widths just past a chunk, a non-power-of-two, three chunk levels, NaN / +-0 / inf rows
    for rows, n in ((3, 16385), (4, 20000), (2, 32768), (2, 70001)):
        d = rng.standard_normal((rows, n)).astype(np.float32)
        d[:, ::7] = np.round(d[:, ::7], 1)
        d[0, :5] = [np.inf, -np.inf, 0.0, -0.0, np.inf]
        got = engine.rank_rows(torch.from_numpy(d).to(dev)).cpu().numpy()
        assert np.array_equal(got, np.argsort(d, axis=1, kind='stable')), (rows, n)


def test_evaluator_ranks_a_gallery_wider_than_one_lds_network(dev):
    """17000-entry gallery through ATTEvaluator's device path (rank_rows + rank_metrics): same CMC /
    mAP as the host evaluate() over numpy's stable argsort of the same distance matrix."""
    from grl_amd import engine
    from grl_amd.reid.evaluator.eva_functions import evaluate
    from grl_amd.synthetic import synth_eval_features
    qf, gf, qp, qc, gp, gc = synth_eval_features(24, 17000, seed=3, n_ids=200, noise=4.0)
    dist = engine.cosin_dist(qf.to(dev), gf.to(dev))
    idx = engine.rank_rows(dist)
    cmc_d, map_d = engine.rank_metrics(idx, qp, gp, qc, gc)
    cmc_h, map_h = evaluate(dist.cpu().numpy(), qp, gp, qc, gc)
    assert np.array_equal(idx.cpu().numpy(), np.argsort(dist.cpu().numpy(), axis=1, kind='stable'))
    assert np.allclose(cmc_d, cmc_h, atol=1e-7) and abs(map_d - map_h) < 1e-9


@pytest.mark.parametrize('cin,cout,k,stride,H,W,n', [(64, 64, 3, 1, 16, 8, 2), (128, 96, 3, 2, 16, 16, 1),
                                                      (256, 128, 1, 2, 8, 8, 2)])
def test_conv_bit_exact_vs_fma_chain(dev, cin, cout, k, stride, H, W, n):
    """The implicit-GEMM gather feeds the same k-ordered fmaf chain as the dense kernel:
    conv output == C oracle chain over the (tap-major, channel-minor) im2col matrix, bit for bit."""
    from grl_amd import engine
    from grl_amd._lib import ptr
    from oracle.ref_c import chain_gemm
    rng = np.random.default_rng(cin * 3 + k)
    x = rng.standard_normal((n, H, W, cin)).astype(np.float32)            # channels-last
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xp = np.zeros((n, H + 2 * pad, W + 2 * pad, cin), np.float32)
    xp[:, pad:pad + H, pad:pad + W] = x
    cols = np.empty((n, Ho, Wo, k * k, cin), np.float32)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, :, ky * k + kx] = xp[:, ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride]
    wp = np.ascontiguousarray(w.reshape(cout, cin, k * k).transpose(0, 2, 1)).reshape(cout, k * k * cin)
    ref = chain_gemm(cols.reshape(n * Ho * Wo, k * k * cin), wp)
    y = torch.empty(n * Ho * Wo, cout, device=dev)
    engine.gemm(torch.from_numpy(x).to(dev), torch.from_numpy(wp).to(dev), y, n * Ho * Wo, cout, k * k * cin,
                conv=(H, W, cin, Ho, Wo, k, k, stride, pad))
    assert np.array_equal(y.cpu().numpy(), ref)


@pytest.mark.parametrize('M,N,K,conv', [(512, 256, 2048, None), (300, 200, 128, None),
                                        (128 * 3, 128, 9 * 64, (16, 8, 64, 16, 8, 3, 3, 1, 1)),
                                        (2 * 64, 192, 9 * 128, (16, 16, 128, 8, 8, 3, 3, 2, 1))])
def test_gemm_bf16_storage(dev, M, N, K, conv):
    """bf16-storage datapath (GRL_MATH_BF16S): bf16 operands, residual and output in HBM, fp32
    accumulate/epilogue.  With bf16-representable inputs the only rounding is the final store."""
    from grl_amd import engine
    rng = np.random.default_rng(23 + M)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    a = torch.from_numpy(rng.standard_normal((rows_in, cin)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).bfloat16()
    res = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    y32 = torch.empty(M, N, dtype=torch.float32, device=dev)
    ad, wd, rd, scd, shd = a.to(dev), w.to(dev), res.to(dev), sc.to(dev), sh.to(dev)
    engine.gemm(ad, wd, y, M, N, K, scale=scd, shift=shd, res=rd, relu=True, conv=conv, math=2)
    engine.gemm(ad, wd, y32, M, N, K, scale=scd, shift=shd, res=rd, relu=True, conv=conv, math=2, out_f32=True)
    # fp32 reference through the exact path on the same (bf16-representable) values
    ref = torch.empty(M, N, device=dev)
    engine.gemm(ad.float(), wd.float(), ref, M, N, K, scale=scd, shift=shd, res=rd.float(), relu=True, conv=conv, math=0)
    assert _rel(y32.cpu().numpy(), ref.cpu().numpy()) < 2e-6            # same products, fp32 accumulate
    assert torch.equal(y, y32.bfloat16())                               # one rounding on the store


@pytest.mark.parametrize('M,N,K', [(4096 + 77, 512 + 40, 1024), (600, 200, 256)])
@pytest.mark.parametrize('relu', [False, True])
def test_gemm_bf16_storage_nan_accumulator(dev, M, N, K, relu):
    """the bf16-storage epilogues apply ReLU as one v_max against a floor.  Without ReLU a NaN accumulator
    (a diverged data-gradient GEMM) must come out as NaN in the interior fast path exactly as in the edge tiles' select
    (round 5 stored -inf there: floor -inf; now the floor is a quiet NaN, v_max returns the other operand); with ReLU
    both paths give 0, as `t > 0 ? t : 0` always did.  Shapes: interior + ragged tiles of the 256x256 kernel and of the
    128-row family; every other element equals the run without the poisoned row."""
    from grl_amd import engine
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).bfloat16().to(dev)
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).bfloat16().to(dev)
    sh = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    clean = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    engine.gemm(a, w, clean, M, N, K, shift=sh, relu=relu, math=2)
    bad_rows = [3, 131, M - 1]                                             # first tile, an interior one, the ragged last
    a2 = a.clone()
    a2[bad_rows, 7] = float('nan')
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    engine.gemm(a2, w, y, M, N, K, shift=sh, relu=relu, math=2)
    yb = y[bad_rows].float()
    if relu:
        assert bool((yb == 0).all())
    else:
        assert bool(torch.isnan(yb).all()), 'a NaN accumulator left the no-ReLU epilogue as %r' % yb.flatten()[:4].tolist()
    keep = torch.ones(M, dtype=torch.bool, device=dev)
    keep[bad_rows] = False
    assert torch.equal(y[keep], clean[keep])


@pytest.mark.parametrize('M,N,K,conv', [
    (512, 256, 2048, None), (300, 200, 256, None), (129, 72, 64, None), (5000, 136, 320, None), (8192, 512, 128, None),
    (5 * 16 * 8, 128, 9 * 64, (16, 8, 64, 16, 8, 3, 3, 1, 1)), (3 * 8 * 8, 96, 9 * 128, (16, 16, 128, 8, 8, 3, 3, 2, 1)),
    (7 * 5 * 3, 136, 256, (10, 6, 256, 5, 3, 1, 1, 2, 0))])
def test_gemm_bf16_storage_ring_tile_equals_two_stage_tiles(dev, M, N, K, conv):
    """The bf16-storage 128 x 64 LDS-DMA kernel runs a THREE-stage ring (pieces two stages ahead, counted waits, a tile's
    second stage requested with its first, K of 1 / 2 / 3 / many stages, conv gather with the tap state advanced two stages
    ahead); the 128 x 128 and 64 x 64 tiles run the two-stage loops.  Same MFMA, same k order: all three bit-identical,
    with a residual / ReLU / per-channel affine epilogue, ragged M and N, more tiles than resident workgroups; and equal
    to the exact fp32 path on the same values up to the summation order inside the bf16 MFMA."""
    from grl_amd import engine, _lib
    lib = _lib.load()
    rng = np.random.default_rng(5 + M + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    a = torch.from_numpy(rng.standard_normal((rows_in, cin)).astype(np.float32)).bfloat16().to(dev)
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).bfloat16().to(dev)
    res = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16().to(dev)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    outs = {}
    old = lib.grl_gemm_bf16_tile_mode(0)                       # keep the 256 x 256 kernel out of it
    try:
        for tile in ((128, 128), (128, 64), (64, 64)):
            lib.grl_gemm_force_tile(*tile)
            y = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=dev)
            engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, res=res, relu=True, conv=conv, math=2)
            outs[tile] = y
    finally:
        lib.grl_gemm_force_tile(0, 0)
        lib.grl_gemm_bf16_tile_mode(old)
    assert torch.equal(outs[(128, 64)], outs[(128, 128)]) and torch.equal(outs[(128, 64)], outs[(64, 64)])
    # ... and right: the exact fp32 path on the same (bf16-representable) values, up to the products' summation order
    ref, y32 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    engine.gemm(a.float(), w.float(), ref, M, N, K, scale=sc, shift=sh, res=res.float(), relu=True, conv=conv, math=0)
    lib.grl_gemm_bf16_tile_mode(0); lib.grl_gemm_force_tile(128, 64)
    try:
        engine.gemm(a, w, y32, M, N, K, scale=sc, shift=sh, res=res, relu=True, conv=conv, math=2, out_f32=True)
    finally:
        lib.grl_gemm_force_tile(0, 0); lib.grl_gemm_bf16_tile_mode(old)
    assert _rel(y32.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert torch.equal(outs[(128, 64)], y32.bfloat16())


@pytest.mark.parametrize('M,N,K,conv', [(49152, 256, 512, None), (65536, 512, 128, None), (49152 + 136, 264, 192, None),
                                        (48 * 32 * 32, 256, 9 * 64, (32, 32, 64, 32, 32, 3, 3, 1, 1))])
def test_gemm_bf16_256_kernel_selected_vs_torch_fp32(dev, M, N, K, conv):
    """Direct check of gemm_bf16_256_kernel at sizes the AUTOMATIC heuristic hands to it (>= 192 tiles of 256 x 256,
    N >= 256; verified through grl_gemm_bf16_tile_mode): against a plain torch fp32 product of the same
    bf16-representable operands -- fp32 accumulation on both sides, one rounding on the store."""
    import torch.nn.functional as F
    from grl_amd import engine, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    a = torch.randn(rows_in, cin, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(dev)
    res = torch.randn(M, N, generator=g).bfloat16().to(dev)
    sc, sh = (torch.rand(N, generator=g) + 0.5).to(dev), torch.randn(N, generator=g).to(dev)
    outs = {}
    for mode in (-1, 0):                           # automatic (the 256-tile kernel at these sizes) and never
        old = lib.grl_gemm_bf16_tile_mode(mode)
        try:
            y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, res=res, relu=True, conv=conv, math=2)
            outs[mode] = y
        finally:
            lib.grl_gemm_bf16_tile_mode(old)
    if conv is None:
        acc = a.float() @ w.float().t()
    else:
        H, W, Cc = conv[0], conv[1], conv[2]
        x = a.float().view(-1, H, W, Cc).permute(0, 3, 1, 2)
        wt = w.float().view(N, 3, 3, Cc).permute(0, 3, 1, 2)           # packed [N][tap][C] -> torch [N][C][kh][kw]
        acc = F.conv2d(x, wt, padding=1).permute(0, 2, 3, 1).reshape(M, N)
    ref = torch.relu(acc * sc + sh + res.float())
    err = float((outs[-1].float() - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err                          # one bf16 rounding of the output (2^-8 of the largest value)
    assert float((outs[-1].float() - ref).abs().mean() / ref.abs().mean()) < 2e-3
    assert torch.equal(outs[-1], outs[0])           # and bit-identical to the 128 x 128 family


@pytest.mark.parametrize('M,N,K,conv,gb', [
    (700, 520, 256, None, False),                                     # ragged M and N (N % 8 == 0), 3 x 3 tiles
    (2 * 128 * 3, 256, 2048, None, True),                             # per-clip bias (GCE corr0), rows_per_group = 256
    (3 * 16 * 8, 264, 9 * 64, (16, 8, 64, 16, 8, 3, 3, 1, 1), False),   # 3x3 stride 1, zero-page taps
    (2 * 8 * 8, 256, 9 * 128, (16, 16, 128, 8, 8, 3, 3, 2, 1), False),  # 3x3 stride 2
    (2 * 8 * 4, 512, 256, (16, 8, 256, 8, 4, 1, 1, 2, 0), False),       # 1x1 stride 2 (downsample)
    (256, 256, 128, None, False)])
def test_gemm_bf16_256_tile_equals_128_family(dev, M, N, K, conv, gb):
    """The 256 x 256 LDS-DMA kernel (gemm_bf16.hip) against the 128 x 128 register-staged family on
    the same GRL_MATH_BF16S call: same MFMA, same k order => bit-identical bf16 outputs, with
    residual, ReLU, scale/shift, per-clip bias, ragged edges and every implicit-GEMM geometry."""
    from grl_amd import engine, _lib
    lib = _lib.load()
    rng = np.random.default_rng(M + N + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    a = torch.from_numpy(rng.standard_normal((rows_in, cin)).astype(np.float32)).bfloat16().to(dev)
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).bfloat16().to(dev)
    res = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16().to(dev)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    kw = dict(scale=sc, shift=sh, res=res, relu=True, conv=conv, math=2)
    if gb:
        kw.update(gbias=torch.from_numpy(rng.standard_normal((M // 256, N)).astype(np.float32)).to(dev), rows_per_group=256)
    ys = []
    try:
        for mode in (0, 1):
            lib.grl_gemm_bf16_tile_mode(mode)
            y = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev)      # guard row: no write past M
            engine.gemm(a, w, y, M, N, K, **kw)
            ys.append(y)
    finally:
        lib.grl_gemm_bf16_tile_mode(-1)
    assert torch.equal(ys[0], ys[1])
    assert bool((ys[1][M] == 7.0).all())
    ref = torch.empty(M, N, device=dev)
    kw32 = dict(kw, res=res.float(), math=0)
    engine.gemm(a.float(), w.float(), ref, M, N, K, **kw32)
    assert torch.equal(ys[1][:M], ref.bfloat16()) or _rel(ys[1][:M].float().cpu().numpy(), ref.cpu().numpy()) < 8e-3


@pytest.mark.parametrize('M,N,K,n,res', [
    (8192, 512, 2048, 2, False),        # the TRL memo block's conv1 of both directions (256 x 128 ring tiles, 256 of them)
    (8192, 512, 512, 2, False),         # ... conv2
    (2048, 2048, 512, 2, True),         # residual, 256 x 256 ring tiles
    (700, 520, 192, 3, True),           # ragged M and N, three problems, edge-tile epilogue
    (256, 64, 64, 4, False),            # one tile per problem, K = one stage
    (300, 200, 128, 2, True)])          # N % 8 == 0 only
def test_gemm_group_equals_separate_launches(dev, M, N, K, n, res):
    """grl_conv_gemm_f32_group (round 5: the bf16 ring kernel over several problems of one shape) against one
    grl_conv_gemm_f32 call per problem: bit-identical outputs, nothing written past a problem's M rows."""
    from grl_amd import engine
    rng = np.random.default_rng(M + N + K + n)
    calls, refs = [], []
    for g in range(n):
        a = torch.from_numpy(np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)).bfloat16().to(dev)
        w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).bfloat16().to(dev)
        r = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).bfloat16().to(dev) if res else None
        sc = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).to(dev)
        sh = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
        y = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev)
        ref = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev)
        engine.gemm(a, w, ref, M, N, K, scale=sc, shift=sh, res=r, relu=True, math=2)
        calls.append(dict(a=a, w=w, y=y, M=M, N=N, K=K, scale=sc, shift=sh, res=r, relu=True, math=2))
        refs.append(ref)
    engine.gemm_group(calls)
    for c, ref in zip(calls, refs):
        assert torch.equal(c['y'], ref)
        assert bool((c['y'][M] == 7.0).all())


def test_gemm_group_falls_back_for_ungroupable_calls(dev):
    """Different shapes, or the exact-fp32 datapath: the group call runs the problems one by one (same results)."""
    from grl_amd import engine
    rng = np.random.default_rng(5)
    calls, refs = [], []
    for (M, N, K) in ((128, 64, 64), (256, 128, 32)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev)
        w = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).to(dev)
        sc = torch.ones(N, device=dev)
        sh = torch.zeros(N, device=dev)
        y, ref = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        engine.gemm(a, w, ref, M, N, K, scale=sc, shift=sh, math=0)
        calls.append(dict(a=a, w=w, y=y, M=M, N=N, K=K, scale=sc, shift=sh, math=0))
        refs.append(ref)
    engine.gemm_group(calls)
    for c, ref in zip(calls, refs):
        assert torch.equal(c['y'], ref)


def test_bf16_storage_pointwise_twins(dev):
    """bf16-storage twins of the bandwidth-bound kernels against their fp32 twins on
    bf16-representable data: identical fp32 results for the reductions, one rounding on bf16
    outputs."""
    from grl_amd import engine
    from grl_amd._lib import ptr
    rng = np.random.default_rng(31)
    tb = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).bfloat16().to(dev)
    # max pool (exact)
    n, H, W, Cc = 2, 12, 10, 64
    x = tb(n * H * W, Cc)
    y16 = torch.empty(n * 6 * 5, Cc, dtype=torch.bfloat16, device=dev)
    y32 = torch.empty(n * 6 * 5, Cc, device=dev)
    xf = x.float()
    engine._call('grl_maxpool3x3s2_bf16', ptr(x), ptr(y16), n, H, W, Cc)
    engine._call('grl_maxpool3x3s2', ptr(xf), ptr(y32), n, H, W, Cc)
    assert torch.equal(y16.float(), y32)
    # group mean / sqdiff mean (fp32 outputs)
    b, T, P, C2 = 3, 2, 128, 2048
    a = tb(b, T, P, C2); af = a.float()
    g16, g32 = torch.zeros(b * T, C2, device=dev), torch.zeros(b * T, C2, device=dev)
    engine._call('grl_group_mean_bf16', ptr(a), ptr(g16), b * T, P, C2, C2, C.c_float(1.0), 0)
    engine._call('grl_group_mean', ptr(af), ptr(g32), b * T, P, C2, C2, C.c_float(1.0), 0)
    assert _rel(g16.cpu().numpy(), g32.cpu().numpy()) < 1e-5
    f1 = tb(b, P, C2); f1f = f1.float()
    d16, d32 = torch.empty(b, C2, device=dev), torch.empty(b, C2, device=dev)
    engine._call('grl_sqdiff_mean_bf16', ptr(f1), ptr(a.view(-1)[P * C2:]), ptr(d16), b, P, C2, T * P * C2)
    engine._call('grl_sqdiff_mean', ptr(f1f), ptr(af.view(-1)[P * C2:]), ptr(d32), b, P, C2, T * P * C2)
    assert _rel(d16.cpu().numpy(), d32.cpu().numpy()) < 1e-5
    # temporal mean / add_strided (bf16 outputs = rounded fp32 results)
    tm16 = torch.empty(b, P * C2, dtype=torch.bfloat16, device=dev); tm32 = torch.empty(b, P * C2, device=dev)
    engine._call('grl_temporal_mean_bf16', ptr(a), ptr(tm16), b, T, P * C2)
    engine._call('grl_temporal_mean', ptr(af), ptr(tm32), b, T, P * C2)
    assert torch.equal(tm16, tm32.bfloat16())
    s16 = torch.empty(b, P * C2, dtype=torch.bfloat16, device=dev); s32 = torch.empty(b, P * C2, device=dev)
    engine._call('grl_add_strided_bf16', ptr(f1), ptr(a.view(-1)[P * C2:]), ptr(s16), b, P * C2, T * P * C2)
    engine._call('grl_add_strided', ptr(f1f), ptr(af.view(-1)[P * C2:]), ptr(s32), b, P * C2, T * P * C2)
    assert torch.equal(s16, s32.bfloat16())
    # gate
    M = 37
    h, xx = tb(M, 256), tb(M, C2)
    w3 = torch.from_numpy((rng.standard_normal(256) * 0.1).astype(np.float32)).to(dev)
    bs, bh = torch.tensor([0.8], device=dev), torch.tensor([-0.1], device=dev)
    cm16, cm32 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    xc16, xu16 = torch.empty(M, C2, dtype=torch.bfloat16, device=dev), torch.empty(M, C2, dtype=torch.bfloat16, device=dev)
    xc32, xu32 = torch.empty(M, C2, device=dev), torch.empty(M, C2, device=dev)
    hf, xxf = h.float(), xx.float()
    engine._call('grl_gce_gate_bf16', ptr(h), ptr(w3), ptr(bs), ptr(bh), ptr(xx), ptr(cm16), ptr(xc16), ptr(xu16), M, 256, C2)
    engine._call('grl_gce_gate', ptr(hf), ptr(w3), ptr(bs), ptr(bh), ptr(xxf), ptr(cm32), ptr(xc32), ptr(xu32), M, 256, C2)
    assert _rel(cm16.cpu().numpy(), cm32.cpu().numpy()) < 1e-5
    assert _rel(xc16.float().cpu().numpy(), xc32.cpu().numpy()) < 8e-3 and _rel(xu16.float().cpu().numpy(), xu32.cpu().numpy()) < 8e-3
    # stem (bf16 MFMA) vs the fp32 stem on bf16-representable pixels/weights
    n, H, W = 2, 64, 32
    xi = torch.from_numpy(rng.standard_normal((n, 3, H, W)).astype(np.float32)).bfloat16().float().to(dev)
    ws = torch.from_numpy((rng.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32)).bfloat16().float().to(dev)
    sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.1
    o32 = torch.empty(n * 32 * 16, 64, device=dev); o16 = torch.empty(n * 32 * 16, 64, dtype=torch.bfloat16, device=dev)
    wpb = torch.empty(64 * 184, dtype=torch.bfloat16, device=dev)
    engine._call('grl_stem_pack_weight_bf16', ptr(ws), ptr(wpb))
    engine._call('grl_stem_conv7x7', ptr(xi), ptr(ws), ptr(sc), ptr(sh), ptr(o32), n, H, W, 1, None)
    engine._call('grl_stem_conv7x7_bf16', ptr(xi), ptr(ws), ptr(sc), ptr(sh), ptr(o16), n, H, W, 1, ptr(wpb))
    assert _rel(o16.float().cpu().numpy(), o32.cpu().numpy()) < 8e-3
    o16b = torch.empty_like(o16)
    engine._call('grl_stem_conv7x7_bf16', ptr(xi), ptr(ws), ptr(sc), ptr(sh), ptr(o16b), n, H, W, 1, None)
    assert torch.equal(o16, o16b)                     # packed image == in-kernel conversion


def _im2col(x, k_h, k_w, stride, pad):
    n, H, W, cin = x.shape
    Ho, Wo = (H + 2 * pad - k_h) // stride + 1, (W + 2 * pad - k_w) // stride + 1
    xp = np.zeros((n, H + 2 * pad + k_h, W + 2 * pad + k_w, cin), np.float32)     # slack rows stay zero
    xp[:, pad:pad + H, pad:pad + W] = x
    cols = np.empty((n, Ho, Wo, k_h * k_w, cin), np.float32)
    for ky in range(k_h):
        for kx in range(k_w):
            cols[:, :, :, ky * k_w + kx] = xp[:, ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride]
    return cols.reshape(n * Ho * Wo, k_h * k_w * cin), Ho, Wo


def test_gemm_and_conv_fuzz_bit_exact(dev):
    """Seeded random shapes against the C oracle's fmaf chain, bit for bit: ragged M / N (scalar and
    float4 epilogues), every tile shape, more tiles than resident workgroups (a persistent
    workgroup then walks several tiles with the next tile's stage prefetched), rectangular and
    even-sized conv kernels, pad 0, stride 2, and the forced tile shapes."""
    from grl_amd import engine, _lib
    from oracle.ref_c import chain_gemm
    lib = _lib.load()
    rng = np.random.default_rng(2024)
    dense = [(int(rng.integers(1, 700)), int(rng.integers(1, 300)), 32 * int(rng.integers(1, 9))) for _ in range(10)]
    dense += [(70000, 192, 64), (33000, 72, 96), (1100 * 128 // 8, 64, 32)]       # > 768 / 1024 tiles
    for M, N, K in dense:
        a = rng.standard_normal((M, K)).astype(np.float32)
        w = rng.standard_normal((N, K)).astype(np.float32)
        ref = chain_gemm(a, w)
        for tile in (None, (64, 64), (128, 64), (128, 128)):
            if tile and M * N > 4e6:
                continue
            lib.grl_gemm_force_tile(*(tile or (0, 0)))         # (per call; GRL_GEMM_TILE is read once per process)
            try:
                y = torch.full((M, N), 7.0, device=dev)
                engine.gemm(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), y, M, N, K)
            finally:
                lib.grl_gemm_force_tile(0, 0)
            assert np.array_equal(y.cpu().numpy(), ref), (M, N, K, tile)
    convs = [(32, 40, 3, 3, 1, 1, 9, 7, 3), (64, 64, 1, 2, 1, 0, 8, 8, 2), (32, 96, 2, 2, 1, 0, 6, 10, 2),
             (64, 33, 3, 3, 2, 1, 12, 8, 5), (32, 64, 2, 1, 1, 0, 7, 5, 4), (96, 130, 1, 1, 2, 0, 10, 6, 3),
             (64, 64, 3, 3, 1, 1, 64, 32, 70)]                                     # 143360 rows: 2240 tiles
    for cin, cout, kh, kw, stride, pad, H, W, n in convs:
        x = rng.standard_normal((n, H, W, cin)).astype(np.float32)
        w = (rng.standard_normal((cout, kh * kw * cin)) / np.sqrt(cin * kh * kw)).astype(np.float32)
        cols, Ho, Wo = _im2col(x, kh, kw, stride, pad)
        ref = chain_gemm(cols, w)
        y = torch.empty(n * Ho * Wo, cout, device=dev)
        engine.gemm(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), y, n * Ho * Wo, cout, kh * kw * cin,
                    conv=(H, W, cin, Ho, Wo, kh, kw, stride, pad))
        assert np.array_equal(y.cpu().numpy(), ref), (cin, cout, kh, kw, stride, pad, H, W, n)


@pytest.mark.parametrize('tile', [(128, 128), (128, 64)])
def test_hand_scheduled_stage_loop_ragged_edges_bit_exact(dev, tile):
    """The LDS-DMA kernels' hand-scheduled stage loop (gemm_f32.hip PIPE: hidden DMA pieces, fragments carried across the
    stage barrier, per-tile conv validity bits) on every edge it has: M and N that end inside a tile (clamped rows),
    K = 256 .. 2304 (8 .. 72 stages, odd and even), more tiles than resident workgroups (next tile's first stage issued
    behind the epilogue), strided A (lda > K), implicit-GEMM windows with stride 2 / no padding / images that end inside
    a tile -- each on both 128-row tiles, bit for bit against the C oracle's fmaf chain."""
    from grl_amd import engine, _lib
    from oracle.ref_c import chain_gemm
    lib = _lib.load()
    rng = np.random.default_rng(7 + tile[1])
    lib.grl_gemm_force_tile(*tile)
    try:
        # (N = 2048 / 3072: 16 / 24 column tiles on the 128-wide tile, 32 / 48 on the 64-wide one -- the column-panel tile walk)
        for M, N, K in [(300, 200, 256), (129, 257, 512), (1000, 130, 288), (77, 64, 2304), (40000, 136, 320),
                        (1200, 2048, 256), (700, 3072, 288)]:
            a = rng.standard_normal((M, K)).astype(np.float32)
            w = rng.standard_normal((N, K)).astype(np.float32)
            y = torch.full((M, N), 7.0, device=dev)
            engine.gemm(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), y, M, N, K)
            assert np.array_equal(y.cpu().numpy(), chain_gemm(a, w)), (M, N, K)
        # strided A: the GEMM reads K columns of a wider matrix
        M, N, K, lda = 500, 96, 384, 640
        a = rng.standard_normal((M, lda)).astype(np.float32)
        w = rng.standard_normal((N, K)).astype(np.float32)
        y = torch.empty(M, N, device=dev)
        engine.gemm(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev), y, M, N, K, lda=lda)
        assert np.array_equal(y.cpu().numpy(), chain_gemm(np.ascontiguousarray(a[:, :K]), w))
        for cin, cout, kh, kw, stride, pad, H, W, n in [(64, 72, 3, 3, 1, 1, 9, 7, 5), (96, 40, 3, 3, 2, 1, 11, 13, 3),
                                                         (256, 130, 1, 1, 2, 0, 10, 6, 7), (32, 64, 3, 2, 1, 0, 12, 9, 4),
                                                         (128, 64, 3, 3, 1, 1, 16, 8, 33)]:
            x = rng.standard_normal((n, H, W, cin)).astype(np.float32)
            w = (rng.standard_normal((cout, kh * kw * cin)) / np.sqrt(cin * kh * kw)).astype(np.float32)
            cols, Ho, Wo = _im2col(x, kh, kw, stride, pad)
            y = torch.empty(n * Ho * Wo, cout, device=dev)
            engine.gemm(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), y, n * Ho * Wo, cout, kh * kw * cin,
                        conv=(H, W, cin, Ho, Wo, kh, kw, stride, pad))
            assert np.array_equal(y.cpu().numpy(), chain_gemm(cols, w)), (cin, cout, kh, kw, stride, pad, H, W, n)
    finally:
        lib.grl_gemm_force_tile(0, 0)
    assert lib.grl_gemm_force_tile(64, 128) < 0 and lib.grl_gemm_force_tile(0, 0) == 0      # illegal pair refused, state untouched


def test_device_augmentation_matches_reference_transforms(dev, golden):
    """grl_augment_normalize_u8 (flip + erase + ToTensor + Normalize in one pass over raw uint8
    clips) against the output of the reference's own PIL pipeline (tests/golden/augment.npz,
    seqtransforms.py:92-216 as composed in dataloader.py:51-57), bit for bit; then at the real
    256 x 128 size against the oracle restatement with every branch forced."""
    import random
    from grl_amd import engine
    from grl_amd.reid.data.augment import draw_clip_params, pack_params
    from oracle import grl_oracle as O
    g = golden('augment.npz')
    N, T, H, W = [int(v) for v in g['shape']]
    u8 = synth_clips(N, T, seed=int(g['clips_seed']), h=H, w=W, raw=True)
    random.seed(int(g['seed']))
    params = pack_params([draw_clip_params(T, H, W) for _ in range(N)])
    out = engine.augment_normalize_u8(u8.to(dev), params)
    assert np.array_equal(out.cpu().numpy(), g['out'])
    # full-size frames, T = 4: patches clipped by the right / bottom border, flips, untouched frames
    B, T = 6, 4
    u8 = synth_clips(B, T, seed=3, raw=True)
    rnd = random.Random(11)
    plist = [draw_clip_params(T, 256, 128, rnd) for _ in range(B)]
    plist[0][0] = 1; plist[0][1:9] = [1, 100, 200, 100, 200, 7, 8, 9]        # clipped on both borders
    plist[1] = [0] * (1 + 8 * T)                                             # nothing happens
    params = pack_params(plist)
    out = engine.augment_normalize_u8(u8.to(dev), params.to(dev))
    assert np.array_equal(out.cpu().numpy(), O.augment_apply(u8.numpy(), params.numpy()))
    assert torch.equal(out[1], engine.normalize_u8(u8[1:2].to(dev))[0])      # = plain ToTensor + Normalize
    with pytest.raises(ValueError):
        engine.augment_normalize_u8(u8.to(dev), params[:, :-1])


def test_trainer_consumes_raw_clips_with_device_augmentation(dev):
    """A loader that yields (uint8 clips, pids, camids, augmentation block): the trainer's
    _parse_data hands the model the flipped / erased / normalised float clips."""
    from torch.utils.data import DataLoader
    from grl_amd import engine
    from grl_amd.reid.data import SyntheticPairs
    from grl_amd.reid.train.trainer import SEQTrainer
    from oracle import grl_oracle as O
    loader = DataLoader(SyntheticPairs(2, 2, seed=4, augment=True), batch_size=4)
    batch = next(iter(engine.DevicePrefetcher(loader, dev)))
    assert len(batch) == 4 and batch[0].dtype == torch.uint8 and batch[3].shape == (4, 1 + 8 * 2)
    tr = SEQTrainer.__new__(SEQTrainer)
    tr.device = dev
    (imgs,), pids = tr._parse_data(batch)
    assert imgs.dtype == torch.float32 and pids.is_cuda
    assert np.array_equal(imgs.cpu().numpy(), O.augment_apply(batch[0].cpu().numpy(), batch[3].cpu().numpy()))


def test_device_rect_scale_matches_reference_and_pil(dev, golden):
    """grl_resize_bilinear_u8 against the reference's RectScale output (tests/golden/augment.npz:
    six input sizes -> 64 x 32) and, at the real 256 x 128 target, against Pillow itself."""
    from grl_amd import engine
    g = golden('augment.npz')
    for k in range(6):
        hh, ww = [int(v) for v in g['rect.%d.shape' % k]]
        src = np.random.Generator(np.random.PCG64(100 + k)).integers(0, 256, (hh, ww, 3), dtype=np.uint8)
        x = torch.from_numpy(np.ascontiguousarray(src.transpose(2, 0, 1))).to(dev)
        y = engine.rect_scale_u8(x, 64, 32)
        assert np.array_equal(y.cpu().numpy().transpose(1, 2, 0), g['rect.%d.out' % k]), (hh, ww)
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(9)
    for hh, ww in ((128, 64), (300, 150), (277, 131)):
        frames = rng.integers(0, 256, (2, 3, hh, ww), dtype=np.uint8)                 # [B*T, 3, H, W]
        y = engine.rect_scale_u8(torch.from_numpy(frames).to(dev)).cpu().numpy()
        for n in range(2):
            ref = np.asarray(Image.fromarray(np.ascontiguousarray(frames[n].transpose(1, 2, 0)), 'RGB').resize((128, 256), Image.BILINEAR))
            assert np.array_equal(y[n].transpose(1, 2, 0), ref), (hh, ww)
    x = torch.from_numpy(rng.integers(0, 256, (1, 2, 3, 256, 128), dtype=np.uint8)).to(dev)
    assert engine.rect_scale_u8(x) is x


@pytest.mark.parametrize('b,t', [(2, 3), (5, 2)])
def test_gemm_sqdiff_epilogue_matches_unfused_path(dev, b, t):
    """GRL_EPI_SQDIFF (TRL step: GAP((ReLU(conv_f1(memo)) - f2_t)^2) reduced inside the GEMM epilogue,
    the conv output never stored) against the unfused launches (GEMM -> grl_sqdiff_mean) and a
    float64 reference; a clip's result does not depend on the batch around it."""
    from grl_amd import engine
    from grl_amd._lib import ptr, EPI_SQDIFF
    rng = np.random.default_rng(b * 10 + t)
    Cc, PIX = 256, 128
    memo = torch.from_numpy(rng.standard_normal((b * PIX, Cc)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((Cc, Cc)) / 16).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32)).to(dev)
    f2 = torch.from_numpy(np.maximum(rng.standard_normal((b * t * PIX, Cc)), 0).astype(np.float32)).to(dev)
    ti = t - 1
    dpart = torch.empty(b * 4, Cc, device=dev)
    engine.gemm(memo, w, dpart, b * PIX, Cc, Cc, shift=bias, epilogue=EPI_SQDIFF, res=f2[ti * PIX:], res_rows=PIX,
                res_gstride=t * PIX)
    d = torch.empty(b, Cc, device=dev)
    engine._call('grl_group_mean', ptr(dpart), ptr(d), b, 4, Cc, Cc, C.c_float(1.0 / 32.0), 0)
    f1 = torch.empty(b * PIX, Cc, device=dev)
    engine.gemm(memo, w, f1, b * PIX, Cc, Cc, shift=bias, relu=True)
    d_ref = torch.empty(b, Cc, device=dev)
    engine._call('grl_sqdiff_mean', ptr(f1), ptr(f2[ti * PIX:]), ptr(d_ref), b, PIX, Cc, t * PIX * Cc)
    exact = ((f1.double().view(b, PIX, Cc) - f2.double().view(b, t, PIX, Cc)[:, ti]) ** 2).mean(1)
    assert _rel(d.cpu().numpy(), exact.cpu().numpy()) < 1e-6
    assert _rel(d.cpu().numpy(), d_ref.cpu().numpy()) < 1e-6
    # clip 0 alone: identical bits
    dp1 = torch.empty(4, Cc, device=dev)
    engine.gemm(memo[:PIX], w, dp1, PIX, Cc, Cc, shift=bias, epilogue=EPI_SQDIFF, res=f2[ti * PIX:], res_rows=PIX,
                res_gstride=t * PIX)
    assert torch.equal(dp1, dpart[:4])


def test_small_kernels_next_to_a_gemm_on_another_stream():
    """Regression for the packed-fp32 corruption found while running the two TRL directions on two HIP streams
    (tools/hw_probe/README.md): one channel-attention step on stream B while a bf16-storage GEMM retires on stream
    A must reproduce its quiet result bit for bit -- with `v_pk_mul_f32` in the kernel one term of the 128-term
    dot product went missing in 16 lanes in ~3 of 4 runs; the library is built without packed fp32."""
    from grl_amd.engine import ptr, _call, gemm, MATH_BF16S
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(5)
    b, t, Cc, Hd, Mb = 32, 4, 2048, 128, 4096
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    shift, dvec, gapc = rnd(Cc) * 0.1, rnd(b, Cc).abs(), rnd(b * t, Cc).abs()
    w1, w2t = rnd(Hd, Cc) * 0.05, rnd(Hd, Cc) * 0.05
    a, w = (rnd(Mb, Cc) * 0.5).bfloat16(), (rnd(Cc, Cc) * 0.02).bfloat16()
    y = torch.empty(Mb, Cc, device=dev, dtype=torch.bfloat16)
    fc, hid = torch.empty(b, t, Cc, device=dev), torch.empty(b, Hd, device=dev)

    def step():
        _call('grl_channel_atte', ptr(dvec), ptr(w1), ptr(w2t), ptr(gapc), t * Cc, None, ptr(fc), t * Cc, 0, b, Cc,
              Hd, ptr(hid))
    step()
    gemm(a, w, y, Mb, Cc, Cc, shift=shift, relu=True, math=MATH_BF16S)
    torch.cuda.synchronize()
    want_fc, want_y = fc[:, 0].clone(), y.clone()
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for _ in range(40):
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            gemm(a, w, y, Mb, Cc, Cc, shift=shift, relu=True, math=MATH_BF16S)
        with torch.cuda.stream(sb):
            step()
        torch.cuda.synchronize()
        assert torch.equal(fc[:, 0], want_fc) and torch.equal(y, want_y)
