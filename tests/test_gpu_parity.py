"""End-to-end parity of the HIP path (through the reid API + C ABI) with the
reference-generated golden vectors and the oracle, on a real MI355X.
Tolerance: north_star asks <= 1e-3 relative fp32; the eval path is held to 1e-4."""
import numpy as np
import pytest
import torch

from grl_amd.synthetic import synth_clips, synth_eval_features

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope='module')
def gpu_models(synth_models):
    assert torch.cuda.is_available()
    cnn, siam, siamv = synth_models
    dev = torch.device('cuda:0')
    return cnn.to(dev).eval(), siam.to(dev).eval(), siamv.to(dev).eval()


def _sample_check(t, g, key, tol=TOL):
    f = t.detach().cpu().contiguous().reshape(-1).double()
    assert tuple(t.shape) == tuple(g[key + '.shape'])
    assert _rel(f[torch.from_numpy(g[key + '.idx'])].numpy(), g[key + '.val']) < tol
    assert abs(f.sum().item() - g[key + '.sum']) <= tol * g[key + '.abssum']


def test_eval_forward_matches_reference_golden(golden, gpu_models):
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    g = golden('grl_eval_b2t4.npz')
    clips = synth_clips(2, 4, seed=0).cuda()
    taps = {}
    xu, xc = engine.grl_forward(cnn, clips, taps=taps)
    for k in ('stem', 'pool', 'layer1', 'layer2', 'layer3', 'layer4'):
        _sample_check(taps[k], g, 'tap.' + k)
    assert _rel(taps['corr_map'].cpu().numpy(), g['corr_map']) < TOL
    assert _rel(torch.stack(taps['fwd_catte'])[:, :, ::16].cpu().numpy(), g['catte.fwd']) < TOL
    assert _rel(torch.stack(taps['bwd_catte'])[:, :, ::16].cpu().numpy(), g['catte.bwd']) < TOL
    assert _rel(taps['f_uncorr'].cpu().numpy(), g['f_uncorr']) < TOL
    assert _rel(taps['f_corr'].cpu().numpy(), g['f_corr']) < TOL
    assert _rel(xu.cpu().numpy(), g['x_uncorr']) < TOL
    assert _rel(xc.cpu().numpy(), g['x_corr']) < TOL
    # The product path (no taps): stem + max-pool are ONE launch there (round 5, grl_stem_pool_f32: the same products in
    # another fp32 summation order), so it is pinned to the golden outputs by itself and agrees with the tapped run, whose
    # stem is the two-launch form, to rounding -- not to the bit.
    xu1, xc1 = engine.grl_forward(cnn, clips)
    assert _rel(xu1.cpu().numpy(), g['x_uncorr']) < TOL and _rel(xc1.cpu().numpy(), g['x_corr']) < TOL
    assert _rel(xu1.cpu().numpy(), xu.cpu().numpy()) < 1e-5 and _rel(xc1.cpu().numpy(), xc.cpu().numpy()) < 1e-5
    with pytest.MonkeyPatch.context() as mp:                   # (the two-launch stem on the product path: the tapped run's bits)
        mp.setattr(engine, 'FUSE_STEM_POOL_F32', False)
        xu0, xc0 = engine.grl_forward(cnn, clips)
    assert torch.equal(xu0, xu) and torch.equal(xc0, xc)
    # the module API (what mars_train.py / the evaluator call) gives the product path's tensors
    xu2, xc2 = cnn(clips)
    assert torch.equal(xu2, xu1) and torch.equal(xc2, xc1)
    feat = engine.extract_features(cnn, siam, clips)
    assert _rel(feat.cpu().numpy(), g['feat']) < TOL
    pooled = siam.self_attention(xc1)
    assert torch.equal(torch.cat((xu1, pooled, xc1.mean(dim=1)), 1)[:, :4096], feat[:, :4096])


def test_eval_forward_matches_oracle_other_shapes(gpu_models):
    """T=8 (reference default seq_len) and an odd batch, against the oracle."""
    from oracle import grl_oracle as O
    cnn, siam, _ = gpu_models
    sd = {k: v.detach().cpu() for k, v in cnn.state_dict().items()}
    ssd = {k: v.detach().cpu() for k, v in siam.state_dict().items()}
    for b, t, seed in ((1, 8, 3), (3, 2, 4)):
        clips = synth_clips(b, t, seed=seed)
        from grl_amd import engine
        feat = engine.extract_features(cnn, siam, clips.cuda())
        assert _rel(feat.cpu().numpy(), O.extract_features(sd, ssd, clips).numpy()) < TOL


def test_features_do_not_depend_on_batch_composition(gpu_models):
    """Eval BN is folded, clips are independent and every GEMM element is one
    k-ordered accumulation chain whatever the tile shape, so a clip's feature row is
    bit-identical in a batch of 32 (BASELINE size) and in a batch of 2."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    clips = synth_clips(32, 4, seed=5).cuda()
    big = engine.extract_features(cnn, siam, clips)
    assert big.shape == (32, 6144) and bool(torch.isfinite(big).all())
    small = engine.extract_features(cnn, siam, clips[10:12].contiguous())
    assert torch.equal(big[10:12], small)
    nrm = big.view(32, 3, 2048).norm(dim=2)
    assert float((nrm[:, :2] - 1).abs().max()) < 1e-5       # unit-norm blocks
    again = engine.extract_features(cnn, siam, clips)
    assert torch.equal(big, again)                          # run-to-run deterministic


def test_siamese_heads_match_reference_golden(golden, gpu_models):
    _, siam, siamv = gpu_models
    g = golden('siamese_b4t4.npz')
    x = torch.from_numpy(g['x']).cuda()
    assert _rel(siam.self_attention(x).cpu().numpy(), g['eval.attn']) < TOL
    cls, out = siam(x)
    assert _rel(cls.cpu().numpy(), g['eval.cls']) < TOL and _rel(out.cpu().numpy(), g['eval.out']) < TOL
    cls, out = siamv(x[:, 0].contiguous())
    assert _rel(cls.cpu().numpy(), g['eval.v_cls']) < TOL and _rel(out.cpu().numpy(), g['eval.v_out']) < TOL
    x8 = torch.randn(2, 16, 2048, generator=torch.Generator().manual_seed(1))
    from oracle import grl_oracle as O
    ssd = {k: v.detach().cpu() for k, v in siam.state_dict().items()}
    assert _rel(siam.self_attention(x8.cuda()).cpu().numpy(), O.self_attention(ssd, x8).numpy()) < TOL


def test_evaluator_matches_reference_golden(golden):
    from grl_amd import engine
    from grl_amd.reid.evaluator.eva_functions import evaluate
    from oracle.ref_c import chain_gemm
    g = golden('evaluator_q40_g400.npz')
    qf, gf, qp, qc, gp, gc = synth_eval_features(40, 400, seed=1, n_ids=24, noise=7.0)
    d = engine.cosin_dist(qf.cuda(), gf.cuda()).cpu().numpy()
    assert _rel(d, g['dist']) < 2e-5
    # bit-exact against the fma-chain oracle => ranking indices bit-exact
    ref = chain_gemm(qf.numpy(), gf.numpy(), mode=1)
    assert np.array_equal(d, ref)
    assert np.array_equal(np.argsort(d, axis=1), np.argsort(ref, axis=1))
    # against the reference's own BLAS ranking: only rounding-level neighbours may swap
    resorted = np.take_along_axis(g['dist'], np.argsort(d, axis=1), 1)
    assert (resorted[:, :-1] - resorted[:, 1:]).max() <= 3e-5
    cmc, mAP = evaluate(d, qp, gp, qc, gc)
    assert np.allclose(cmc[:20], g['cmc'], atol=1e-6) and abs(mAP - float(g['mAP'])) < 1e-6
    # ranking + CMC/AP entirely on the device (grl_row_argsort + grl_rank_metrics)
    cmc_d, map_d = evaluate(None, qp, gp, qc, gc, indices=engine.rank_rows(engine.cosin_dist(qf.cuda(), gf.cuda())))
    assert np.array_equal(cmc_d, cmc) and abs(map_d - mAP) < 1e-12
    assert np.allclose(cmc_d[:20], g['cmc'], atol=1e-6) and abs(map_d - float(g['mAP'])) < 1e-6
    e = engine.pairwise_distance_tensor(qf.cuda(), qf.cuda()).cpu().numpy()
    assert _rel(e ** 2, g['euclid_qq'] ** 2) < 1e-5
    qd = qf.view(20, 2, -1).mean(1); gd = torch.cat((qd, gf[40:240]), 0)
    assert _rel(engine.cosin_dist(qd.cuda(), gd.cuda()).cpu().numpy(), g['dist_dense']) < 2e-5
    e = engine.pairwise_distance_tensor(qd.cuda(), gd.cuda()).cpu().numpy()
    assert _rel(e ** 2, g['euclid_dense'] ** 2) < 1e-5


def test_gpu_ranking_against_the_reference_index_fixtures(golden, capsys):
    """Verdict r5 item 1: the DEVICE ranking (grl_conv_gemm_f32 NEGDOT / EUCLID epilogue + grl_row_argsort)
    against the reference's OWN rankings stored in the fixture -- ``indices`` (cosin_dist,
    attevaluator.py:44-46), ``idx_dense`` (the non-unit-norm dense mode, attevaluator.py:84,95) and
    ``idx_dense_euclid`` (pairwise_distance_tensor, attevaluator.py:33-41), each np.argsort'ed as
    eva_functions.py:139 does.  Contract (tests/ranking_check.py): positions may differ only where the
    reference's own distances of the two entries are within 3e-5 (neighbour swaps below fp32 noise), an
    entry moves at most 3 places, and no query's CMC / AP input changes.  The mismatch counts are printed."""
    from grl_amd import engine
    from grl_amd.reid.evaluator.eva_functions import evaluate
    import ranking_check as R
    g = golden('evaluator_q40_g400.npz')
    qf, gf, qp, qc, gp, gc = synth_eval_features(40, 400, seed=1, n_ids=24, noise=7.0)
    qd, gd, qpd, qcd, gpd, gcd = R.dense_case(qf, gf, qp, qc, gp, gc)
    cases = [
        ('cosine  40x400 (indices)', engine.cosin_dist(qf.cuda(), gf.cuda()), g['indices'], g['dist'], qp, gp, qc, gc),
        ('cosine  dense 20x220 (idx_dense)', engine.cosin_dist(qd.cuda(), gd.cuda()), g['idx_dense'], g['dist_dense'],
         qpd, gpd, qcd, gcd),
        ('euclid  dense 20x220 (idx_dense_euclid)', engine.pairwise_distance_tensor(qd.cuda(), gd.cuda()),
         g['idx_dense_euclid'], g['euclid_dense'], qpd, gpd, qcd, gcd),
    ]
    report = []
    for name, dmat, ref_idx, ref_dist, a, b, c, d in cases:
        idx_dev = engine.rank_rows(dmat)                                    # device argsort of the device matrix
        ours = idx_dev.cpu().numpy()
        assert np.array_equal(ours, np.argsort(dmat.cpu().numpy(), axis=1, kind='stable'))
        r = R.compare(ours, ref_idx, ref_dist, a, b, c, d, tol=3e-5)
        report.append((name, r))
        # and the metric computed ON THE DEVICE from the device ranking equals the one the reference's
        # ranking gives on the host
        cmc_ref, map_ref = evaluate(ref_dist, a, b, c, d, indices=ref_idx.astype(np.int64))
        cmc_dev, map_dev = evaluate(None, a, b, c, d, indices=idx_dev)
        assert np.array_equal(cmc_dev, cmc_ref) and abs(map_dev - map_ref) < 1e-12
    with capsys.disabled():
        for name, r in report:
            print('\n[ranking vs reference] %-40s %d of %d positions differ (%d rows), worst reference gap %.2e, '
                  'max shift %d, queries with a changed CMC/AP input: %d'
                  % (name, r['differ'], r['positions'], r['rows_touched'], r['worst_ref_gap'], r['max_shift'],
                     r['metric_changes']))
    assert sum(r['differ'] for _, r in report) <= 16


def test_evaluator_full_mars_size_properties():
    """BASELINE config 5: 1980 x 11310 x 6144.  Bit-exact against the C oracle on a
    row sample, symmetry D(q,g) == D(g,q)^T, self-distance and Euclid/-dot
    consistency on unit-norm-triple rows (|q-g|^2 = 6 - 2 q.g)."""
    from grl_amd import engine
    from oracle.ref_c import chain_gemm
    qf, gf, qp, qc, gp, gc = synth_eval_features(1980, 11310, seed=1)
    qd, gd = qf.cuda(), gf.cuda()
    d = engine.cosin_dist(qd, gd)
    assert d.shape == (1980, 11310)
    rows = np.array([0, 1, 31, 32, 127, 128, 777, 1919, 1951, 1979])
    ref = chain_gemm(qf.numpy()[rows], gf.numpy(), mode=1)
    got = d.cpu().numpy()
    assert np.array_equal(got[rows], ref)
    assert np.array_equal(np.argsort(got[rows], axis=1), np.argsort(ref, axis=1))
    dt = engine.cosin_dist(gd, qd)
    assert torch.equal(dt.t(), d)
    diag = got[np.arange(1980), np.arange(1980)]
    assert np.abs(diag + 3.0).max() < 5e-5 and (got.argmin(1) == np.arange(1980)).all()
    e = engine.pairwise_distance_tensor(qd[:256], gd)
    assert float((e ** 2 - (6 + 2 * d[:256])).abs().max()) < 1e-4


# ----------------------------------------------------------------------------
# train mode: batch-stat BN forward + HIP backward vs the reference golden
# ----------------------------------------------------------------------------
def _fresh_models():
    import contextlib, io
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    return cnn.cuda(), siam.cuda(), siamv.cuda()


def _fresh_cnn_conditioned():
    import contextlib, io
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'))
    return cnn.cuda()


@pytest.mark.parametrize('math', ['f32', 'bf16s'])
def test_default_step_adopts_every_residual_gradient_in_place(math):
    """Tape.owns() keys on storage identity; a tape-owned gradient that shared a storage with a foreign
    tensor would silently lose in-place adoption (a perf cliff, not a wrong result).  On the default step every one of
    the 16 trunk + 2 x T TRL residual blocks must adopt (grad(res) = the masked da buffer, no copy), and every stacked
    weight gradient must have been issued (Tape.backward raises otherwise)."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    cnn = _fresh_cnn_conditioned()
    cnn.train()
    old = TE.set_math(math)
    try:
        TE.ADOPT_STATS[:] = [0, 0]
        xu, xc = cnn(synth_clips_structured(4, 4, seed=3).cuda())
        (xu.sum() + xc.sum()).backward()
        torch.cuda.synchronize()
    finally:
        TE.set_math(old)
    adopted, copied = TE.ADOPT_STATS
    assert copied == 0 and adopted >= 16 + 2 * 4, TE.ADOPT_STATS
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for n, p in cnn.named_parameters()
               if 'temporal_learning_block' in n and n.endswith('conv1.weight'))


@pytest.mark.parametrize('fname,math', [('grl_train_cond_b8t4.npz', 'f32'), ('grl_train_cond_b8t4.npz', 'mixed'),
                                        ('grl_train_cond_b4t8.npz', 'f32'), ('grl_train_cond_b32t4.npz', 'f32'),
                                        ('grl_train_cond_b32t4.npz', 'mixed'), ('grl_train_cond_b64t4.npz', 'f32')])
def test_train_forward_backward_matches_reference_golden_1e3(golden, fname, math):
    """(``math``: the default exact-fp32 step, and 'mixed' = the same forward with split-bf16 backward GEMMs.)
    THE end-to-end backward pin: HIP train-mode forward + backward of the whole CNN against the
    reference's fp32 autograd run (tests/golden/grl_train_cond_b8t4.npz), B x T = 8 x 4, EVERY
    parameter gradient at <= 1e-3 relative L2 (north_star's figure), outputs at <= 1e-4, whole
    tensors covered by norm + projection checksums, BN running statistics at <= 1e-4.  The
    'conditioned' weights / structured clips keep the ReLU-flip floor (what ANY two fp32 runs
    differ by) at ~2e-4 -- measured in the fixture as the reference fp32 vs its own float64 -- so
    a 1 % gradient bug in any layer fails here (tolerance model: tests/train_cond_check.py).
    The chaotic default-weight fixtures below stay as stress tests.  Round 3: the same pin at B x T = 4 x 8 (the
    T = 8 recurrence of BASELINE configs[2]: the memo-block BatchNorms run 8 times per forward) and at configs[1]'s
    FULL size 32 x 4 (grl_train_cond_b32t4.npz: the reference's fp32 + float64 runs at that size); and BASELINE
    configs[3]'s per-GPU batch 64 x 4 (grl_train_cond_b64t4.npz)."""
    import train_cond_check as TC
    from grl_amd.synthetic import synth_clips_structured
    g = golden(fname)
    B, T = int(g['meta.B']), int(g['meta.T'])
    clip_seed = int(g['meta.clip_seed']) if 'meta.clip_seed' in g.files else 3
    cnn = _fresh_cnn_conditioned()
    cnn.train()
    r1, r2 = TC.upstream(B, T)
    from grl_amd import train_engine as TE
    old = TE.set_math(math)
    try:
        xu, xc = cnn(synth_clips_structured(B, T, seed=clip_seed).cuda())
        ((xu * r1.cuda()).sum() + (xc * r2.cuda()).sum()).backward()
    finally:
        TE.set_math(old)
    grads = {k: p.grad for k, p in cnn.named_parameters() if p.grad is not None}
    TC.check(g, xu, xc, grads, cnn.state_dict(), out_tol=1e-4, grad_tol=1e-3, label='HIP %s %dx%d' % (math, B, T))


@pytest.mark.parametrize('B,T,seed,fname,tol_xu', [(2, 4, 0, 'grl_train_b2t4.npz', 5e-2),
                                                   (4, 2, 2, 'grl_train_b4t2.npz', 4e-3)])
def test_train_forward_backward_chaotic_weights_stress(golden, B, T, seed, fname, tol_xu):
    """One train-mode forward + backward of the CNN: outputs, BN running statistics and
    parameter gradients against the reference (fp32 autograd on CPU).

    Yardstick for the tolerances: the fixtures also hold the SAME graph run by the reference
    model in float64 (f64.* keys).  The fp32 reference itself differs from it by 1.2e-2 (B=2) /
    7e-4 (B=4) on x_uncorr and by 2e-2..4e-2 on individual gradient elements (batch-statistics
    BN over 2-8 samples in front of 50 ReLU layers is that sensitive to rounding; at B = 2
    x_uncorr goes through BatchNorm1d over TWO rows).  The HIP result has to be as close to
    the float64 run as the reference's own fp32 run is (factor 2), which it is since the
    BatchNorm1d statistics are accumulated around a pivot row (grl_col_stats): with plain
    E[x^2] - E[x]^2 in fp32 the B = 2 gradients were 7x further from float64 than torch's."""
    g = golden(fname)
    cnn, _, _ = _fresh_models()
    cnn.train()
    rg = np.random.Generator(np.random.PCG64(7))
    r1 = torch.from_numpy(rg.standard_normal((B, 2048)).astype(np.float32)).cuda()
    r2 = torch.from_numpy(rg.standard_normal((B, T, 2048)).astype(np.float32)).cuda()
    clips = synth_clips(B, T, seed=seed).cuda()
    xu, xc = cnn(clips)
    # uncorr_bn / glo_fc.1 are BatchNorm1d over B = 2 rows here: (x - mean)/sqrt(var + eps) with
    # var = (x0 - x1)^2 / 4 amplifies fp32 rounding of near-equal rows, in the reference too.
    print('fwd rel err: x_uncorr %.2e x_corr %.2e' % (_rel(xu.detach().cpu().numpy(), g['x_uncorr']),
                                                     _rel(xc.detach().cpu().numpy(), g['x_corr'])))
    assert _rel(xu.detach().cpu().numpy(), g['x_uncorr']) < tol_xu
    assert _rel(xc.detach().cpu().numpy(), g['x_corr']) < 2e-3
    loss = (xu * r1).sum() + (xc * r2).sum()
    loss.backward()
    st = cnn.state_dict()
    for k in [k for k in g.files if k.startswith('stat.')]:
        assert _rel(st[k[5:]].double().cpu().numpy(), g[k]) < 1e-4, k
    named = dict(cnn.named_parameters())
    keys = sorted({k[5:].rsplit('.', 1)[0] for k in g.files if k.startswith('grad.')})
    worst = {}
    for k in keys:
        if k == 'input':
            continue                    # the clip itself needs no gradient in training
        gr = named[k].grad
        assert gr is not None, k
        f = gr.detach().cpu().reshape(-1).double()
        ref = g['grad.' + k + '.val']
        err = np.abs(f[torch.from_numpy(g['grad.' + k + '.idx'])].numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        worst[k] = (err, abs(f.abs().sum().item() - g['grad.' + k + '.abssum']) / g['grad.' + k + '.abssum'])
    for k, v in worst.items():
        print('%-70s sample rel err %.2e  abssum rel err %.2e' % (k, v[0], v[1]))
    # The yardstick for this ill-conditioned graph: the reference model run in float64
    # (f64.* keys).  The HIP gradients must be as close to it as the reference's own float32 run.
    e_ref, e_hip = {}, {}
    for k in keys:
        if k == 'input':
            continue
        f = named[k].grad.detach().cpu().reshape(-1).double()
        t64 = g['f64.grad.' + k + '.val']
        scale = max(np.abs(t64).max(), 1e-30)
        e_ref[k] = np.abs(g['grad.' + k + '.val'].astype(np.float64) - t64).max() / scale
        e_hip[k] = np.abs(f[torch.from_numpy(g['grad.' + k + '.idx'])].numpy() - t64).max() / scale
    for k in e_ref:
        print('  %-66s vs f64: ref32 %.2e  hip %.2e' % (k, e_ref[k], e_hip[k]))
    print('vs float64: reference fp32 worst %.2e, HIP worst %.2e' % (max(e_ref.values()), max(e_hip.values())))
    assert max(e_hip.values()) <= 2.0 * max(e_ref.values()) + 1e-3, (max(e_hip.values()), max(e_ref.values()))
    assert _rel(xu.detach().cpu().numpy(), g['f64.x_uncorr']) <= 2.0 * _rel(g['x_uncorr'], g['f64.x_uncorr']) + 1e-4
    assert _rel(xc.detach().cpu().numpy(), g['f64.x_corr']) <= 2.0 * _rel(g['x_corr'], g['f64.x_corr']) + 1e-4
    bad = {k: v for k, v in worst.items() if v[0] > 8e-2 or v[1] > 4e-2}
    assert not bad, bad


# B = 4 clips = 2 pairs on the default (un-conditioned) weights, gradients of (out . rr + cls . rc) against the reference's
# autograd run.  Rounds 1-3 held these to 2e-3; measured on MI355X (round 4): 2.7e-7, 6.7e-7, 2.4e-7 -- the head's kernels
# follow torch's summation closely -- so the assert now sits where a real error would show.
SIAMESE_B4T4_GRAD_TOL = 2e-5


def test_siamese_train_matches_reference_golden(golden):
    g = golden('siamese_b4t4.npz')
    _, siam, _ = _fresh_models()
    siam.train()
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    cls, out = siam(x)
    assert _rel(cls.detach().cpu().numpy(), g['train.cls']) < 1e-3
    assert _rel(out.detach().cpu().numpy(), g['train.out']) < 1e-4
    ((out * torch.from_numpy(g['train.rr']).cuda()).sum() + (cls * torch.from_numpy(g['train.rc']).cuda()).sum()).backward()
    e = (_rel(x.grad.cpu().numpy(), g['train.grad_x']), _rel(siam.featQ.weight.grad[:, ::64].cpu().numpy(), g['train.grad_featQ_w']),
         _rel(siam.classifierlinear.weight.grad.cpu().numpy(), g['train.grad_cls_w']))
    print('siamese_b4t4 gradient errors (grad_x, featQ.weight, classifier.weight): %.2e %.2e %.2e' % e)
    assert max(e) < SIAMESE_B4T4_GRAD_TOL
    assert _rel(siam.featQ_bn.running_mean.cpu().numpy(), g['train.featQ_bn_rm']) < 1e-4


def test_trainer_steps_run_on_hip():
    """Two SEQTrainer iterations (reference loss composition, SGD+nesterov as
    mars_train.py:94-108) on synthetic pairs: finite loss, parameters move, OIM LUT rows
    of the batch identities become unit vectors, eval path still works afterwards."""
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.reid.data import SyntheticPairs
    from torch.utils.data import DataLoader
    cnn, siam, siamv = _fresh_models()
    dev = torch.device('cuda:0')
    crit_c, crit_u = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev)
    trainer = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), crit_c, crit_u, None)
    params = [p for m in (cnn, siam, siamv) for p in m.parameters()]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
    loader = DataLoader(SyntheticPairs(4, 2), batch_size=4, shuffle=False, drop_last=True)
    w0 = cnn.backbone.base[0].weight.detach().clone()
    f0 = cnn.temporal_learning_block.forward_f1[0].weight.detach().clone()
    q0 = siam.featQ.weight.detach().clone()
    trainer.train(0, loader, opt)
    assert len(loader) == 2
    for p in (cnn.backbone.base[0].weight, cnn.temporal_learning_block.forward_f1[0].weight, siam.featQ.weight):
        assert bool(torch.isfinite(p).all())
    assert float((cnn.backbone.base[0].weight - w0).abs().max()) > 0
    assert float((cnn.temporal_learning_block.forward_f1[0].weight - f0).abs().max()) > 0
    assert float((siam.featQ.weight - q0).abs().max()) > 0
    assert siam.featV.weight.grad is None                   # unused upstream as well
    rows = crit_c.lut.norm(dim=1)
    assert int((rows > 0).sum()) >= 1 and float((rows[rows > 0] - 1).abs().max()) < 1e-5
    assert int(cnn.backbone.base[1].num_batches_tracked) == 2
    assert int(cnn.temporal_learning_block.uncorr_memo_forward.bn1.num_batches_tracked) == 4   # T calls per forward
    from grl_amd import engine
    cnn.eval(); siam.eval()
    feat = engine.extract_features(cnn, siam, synth_clips(2, 2, seed=9).cuda())
    assert bool(torch.isfinite(feat).all())


def test_trainer_meters_read_one_step_late_hold_the_same_values(capsys):
    """SEQTrainer.train reads the step's loss / precisions one step late from a pinned buffer (no `loss.item()` stall
    behind the forward); the meters, the writer's scalars and the trained parameters are those of the upstream order
    (GRL_LAZY_METERS=0: trainer.py:70-77), value for value."""
    from grl_amd.reid.train import trainer as T
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.reid.data import SyntheticPairs
    from torch.utils.data import DataLoader
    dev = torch.device('cuda:0')

    class Writer(object):
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, val, it):
            self.rows.append((tag, float(val), int(it)))

    runs = {}
    for lazy in (True, False):
        T.LAZY_METERS = lazy
        try:
            cnn, siam, siamv = _fresh_models()
            crit_c, crit_u = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev)
            tr = T.SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), crit_c, crit_u, None)
            tr.writer = Writer()
            opt = torch.optim.SGD([p for m in (cnn, siam, siamv) for p in m.parameters()], lr=1e-3, momentum=0.9,
                                  weight_decay=5e-4, nesterov=True)
            loader = DataLoader(SyntheticPairs(12, 2), batch_size=4, shuffle=False, drop_last=True)
            tr.train(1, loader, opt)
        finally:
            T.LAZY_METERS = True
        m = tr.meters
        runs[lazy] = ([(k, float(v.val), float(v.avg), int(v.count)) for k, v in sorted(m.items())], tr.writer.rows,
                      cnn.backbone.base[0].weight.detach().clone(), crit_c.lut.clone())
    assert runs[True][0] == runs[False][0] and runs[True][0][0][3] == 24            # 6 steps x 4 clips... per meter
    assert runs[True][1] == runs[False][1] and len(runs[True][1]) == 2 * 6
    assert [r[2] for r in runs[True][1][::2]] == [6 + i for i in range(6)]          # num_iter = len(loader) * epoch + i
    assert torch.equal(runs[True][2], runs[False][2]) and torch.equal(runs[True][3], runs[False][3])


@pytest.mark.parametrize('mode,tol', [('bf16x3', 1e-3), ('bf16', 5e-2), ('bf16s', 2e-2)])
def test_eval_forward_alternative_math_modes(golden, gpu_models, mode, tol):
    """The opt-in bf16 datapaths against the reference golden: split-bf16 stays inside the
    north star's 1e-3 fp32 parity budget; plain bf16 (BASELINE configs[2]) gets its own
    looser bound.  The default (exact fp32) is what every other test pins."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    g = golden('grl_eval_b2t4.npz')
    clips = synth_clips(2, 4, seed=0).cuda()
    with engine.math_mode(mode):
        assert engine.get_math() == mode
        feat = engine.extract_features(cnn, siam, clips)
    assert engine.get_math() == 'f32'
    err = _rel(feat.cpu().numpy(), g['feat'])
    print('%s: feature rel err vs reference %.2e' % (mode, err))
    assert err < tol


def test_attevaluator_end_to_end_both_modes(gpu_models, capsys):
    """ATTEvaluator.evaluate on synthetic loaders: rrs_test mode (mars_train.py) and the
    dense per-tracklet mode of test_all.py (chunks of 8 clips, clip-averaged features,
    attevaluator.py:68-98) against the oracle's feature rows; prints the reference's
    'Mean AP / Rank-k' lines and returns Rank-1."""
    from grl_amd.reid.evaluator import ATTEvaluator
    from oracle import grl_oracle as O
    cnn, siam, _ = gpu_models
    sd = {k: v.detach().cpu() for k, v in cnn.state_dict().items()}
    ssd = {k: v.detach().cpu() for k, v in siam.state_dict().items()}
    T = 2
    rng = np.random.Generator(np.random.PCG64(3))

    def items(n, seed):
        clips = synth_clips(n, T, seed=seed)
        pids = torch.from_numpy(rng.integers(0, 3, n))
        cams = torch.from_numpy(rng.integers(0, 2, n))
        return clips, pids, cams
    q, g = items(4, 21), items(26, 22)          # the reference indexes Rank-20: gallery >= 20
    ev = ATTEvaluator(cnn, siam, only_eval=False)
    qf, qp, qc = ev.extract_feature([q])
    assert _rel(qf.cpu().numpy(), O.extract_features(sd, ssd, q[0]).numpy()) < TOL
    assert list(qp) == list(q[1].numpy())
    import grl_amd.reid.evaluator.eva_functions as EF
    r1 = ev.evaluate(None, None, [q], [g], None, False, False)
    out = capsys.readouterr().out
    assert 'Mean AP:' in out and 'Rank-1' in out and 0.0 <= r1 <= 1.0
    r1r = ev.evaluate(None, None, [q], [g], None, False, True)        # with k-reciprocal re-ranking
    assert 0.0 <= r1r <= 1.0
    # dense mode: one tracklet of 11 clips -> chunks 8 + 3, features averaged over clips
    dense = synth_clips(11, T, seed=31).unsqueeze(0)
    evd = ATTEvaluator(cnn, siam, only_eval=True)
    df, dp, _ = evd.extract_feature([(dense, torch.tensor([7]), torch.tensor([1]))])
    ref = O.extract_features(sd, ssd, dense[0]).mean(dim=0, keepdim=True)
    assert df.shape == (1, 6144) and _rel(df.cpu().numpy(), ref.numpy()) < TOL and int(dp[0]) == 7
    # several tracklets of different lengths: clips are batched ACROSS tracklets (evd.group pending clips, chunks of
    # evd.chunk) -- the per-tracklet means must be the ones a tracklet-by-tracklet run gives, bit for bit
    tracks = [(synth_clips(n, T, seed=40 + n).unsqueeze(0), torch.tensor([n]), torch.tensor([n % 2])) for n in (3, 7, 1, 9, 5)]
    evd.group, evd.chunk = 12, 8
    grouped, gp, gc = evd.extract_feature(tracks)
    evd.group = 1
    single, sp, sc = evd.extract_feature(tracks)
    assert grouped.shape == (5, 6144) and torch.equal(grouped, single)
    assert list(gp) == [3, 7, 1, 9, 5] == list(sp) and list(gc) == list(sc)


@pytest.mark.parametrize('b,t', [(1, 1), (1, 16), (5, 3)])
def test_eval_edge_shapes_match_oracle(gpu_models, b, t):
    """Smallest clip (one frame), the longest the attention kernel takes (T = 16) and a
    ragged batch/length: features against the oracle."""
    from grl_amd import engine
    from oracle import grl_oracle as O
    cnn, siam, _ = gpu_models
    sd = {k: v.detach().cpu() for k, v in cnn.state_dict().items()}
    ssd = {k: v.detach().cpu() for k, v in siam.state_dict().items()}
    clips = synth_clips(b, t, seed=40 + b + t)
    feat = engine.extract_features(cnn, siam, clips.cuda())
    assert _rel(feat.cpu().numpy(), O.extract_features(sd, ssd, clips).numpy()) < TOL


def test_eval_rejects_bad_inputs(gpu_models):
    from grl_amd import engine
    from grl_amd._lib import GrlHipError
    cnn, siam, _ = gpu_models
    with pytest.raises(ValueError):
        cnn(torch.zeros(2, 4, 3, 128, 64, device='cuda'))            # the 16x8 map is hard-wired upstream too
    with pytest.raises(ValueError):
        cnn(torch.zeros(8, 3, 256, 128, device='cuda'))              # missing T axis
    with pytest.raises(GrlHipError):
        cnn(torch.zeros(1, 2, 3, 256, 128, device='cuda', dtype=torch.float16))
    with pytest.raises(GrlHipError):
        siam.self_attention(torch.zeros(1, 17, 2048, device='cuda'))  # T > 16
    with pytest.raises(RuntimeError):
        siam(torch.zeros(3, 4, 2048, device='cuda'))                 # odd batch (Siamese.py:112-113)
    d = engine.cosin_dist(torch.ones(1, 64, device='cuda'), torch.ones(3, 64, device='cuda'))
    assert d.shape == (1, 3) and float(d[0, 0]) == -64.0              # single query row


def test_baseline_config2_bf16_T8_B64(gpu_models):
    """BASELINE configs[2]: bf16 MFMA datapath, T = 8, P x K = 16 x 4 (64 clips).  Full-size
    run checked through size-independent properties (finite, unit-norm blocks, a clip's row is
    bit-identical in the batch of 64 and in a batch of 2) and against the fp32 oracle on two
    clips with the bf16 tolerance."""
    from grl_amd import engine
    from oracle import grl_oracle as O
    cnn, siam, _ = gpu_models
    clips = synth_clips(64, 8, seed=8).cuda()
    with engine.math_mode('bf16s'):
        big = engine.extract_features(cnn, siam, clips)
        small = engine.extract_features(cnn, siam, clips[30:32].contiguous())
    assert big.shape == (64, 6144) and bool(torch.isfinite(big).all())
    assert torch.equal(big[30:32], small)
    assert float((big.view(64, 3, 2048).norm(dim=2)[:, :2] - 1).abs().max()) < 1e-5
    sd = {k: v.detach().cpu() for k, v in cnn.state_dict().items()}
    ssd = {k: v.detach().cpu() for k, v in siam.state_dict().items()}
    ref = O.extract_features(sd, ssd, clips[30:32].cpu())
    err = _rel(small.cpu().numpy(), ref.numpy())
    print('configs[2] bf16 T=8: rel err vs fp32 oracle %.2e' % err)
    assert err < 3e-2


def test_trl_grouped_launches_equal_two_stream_form(gpu_models):
    """bf16 storage: the TRL step's conv1 / conv2 of both directions as grouped launches on one stream (round 5)
    against the two-stream form of rounds 2-4: the same kernels' arithmetic per output, bit-identical features."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    clips = synth_clips(6, 8, seed=11).cuda()
    old = engine.TRL_GROUP
    try:
        with engine.math_mode('bf16s'):
            engine.TRL_GROUP = True
            a = engine.extract_features(cnn, siam, clips)
            engine.TRL_GROUP = False
            b = engine.extract_features(cnn, siam, clips)
    finally:
        engine.TRL_GROUP = old
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_baseline_config4_full_mars_rank1_map():
    """BASELINE configs[4]: the full MARS-size query x gallery matrix on the GPU, then the
    reference's host ranking: Rank-1 / mAP / CMC equal to those of the CPU (BLAS) distance
    matrix of the oracle."""
    from grl_amd import engine
    from grl_amd.reid.evaluator.eva_functions import evaluate
    from oracle import grl_oracle as O
    qf, gf, qp, qc, gp, gc = synth_eval_features(1980, 11310, seed=1, noise=6.0)
    d_dev = engine.cosin_dist(qf.cuda(), gf.cuda())
    d_gpu = d_dev.cpu().numpy()
    d_cpu = O.cosin_dist(qf, gf).numpy()
    assert _rel(d_gpu, d_cpu) < 2e-5
    cmc_g, map_g = evaluate(d_gpu, qp, gp, qc, gc)
    cmc_c, map_c = evaluate(d_cpu, qp, gp, qc, gc)
    print('configs[4]: mAP %.4f Rank-1 %.4f' % (map_g, cmc_g[0]))
    assert abs(map_g - map_c) < 1e-5 and np.abs(cmc_g - cmc_c).max() < 1e-3 and 0.02 < map_g < 0.9999
    # the same protocol with ranking and CMC/AP on the device.  A 11310-entry fp32 row holds a few
    # exactly tied distances; the device sort is stable (ties to the smaller index), numpy's default
    # introsort is not: against the stable host ranking the result is identical (mAP to fp64
    # rounding), against the default one a tied (hit, miss) pair may swap (<= 1e-7 on mAP).
    cmc_d, map_d = evaluate(None, qp, gp, qc, gc, indices=engine.rank_rows(d_dev))
    cmc_s, map_s = evaluate(d_gpu, qp, gp, qc, gc, indices=np.argsort(d_gpu, axis=1, kind='stable'))
    assert np.array_equal(cmc_d, cmc_s) and abs(map_d - map_s) < 1e-12
    assert np.abs(cmc_d - cmc_g).max() < 1e-3 and abs(map_d - map_g) < 1e-7


def test_rank_metrics_edge_cases():
    """grl_rank_metrics vs the host evaluate on: identities that never appear in the gallery
    (query skipped), entries dropped for sharing pid AND camera, a gallery shorter than
    max_rank (flat CMC extension), ragged row lengths that are not a multiple of the 256-entry
    chunk, and the all-queries-invalid assertion."""
    from grl_amd import engine
    from grl_amd.reid.evaluator.eva_functions import evaluate
    rng = np.random.default_rng(5)
    for nq, ng, n_ids in ((7, 37, 5), (33, 300, 40), (64, 1000, 500), (5, 257, 3)):
        d = torch.from_numpy(rng.standard_normal((nq, ng)).astype(np.float32)).cuda()
        qp, gp = rng.integers(0, n_ids, nq), rng.integers(0, n_ids, ng)
        qc, gc = rng.integers(0, 3, nq), rng.integers(0, 3, ng)
        qp[0] = n_ids + 7                                   # never appears -> skipped
        idx = engine.rank_rows(d)
        cmc_h, map_h = evaluate(d.cpu().numpy(), qp, gp, qc, gc, indices=idx.cpu().numpy())
        cmc_d, map_d = evaluate(None, qp, gp, qc, gc, indices=idx)
        assert cmc_d.shape == cmc_h.shape and np.array_equal(cmc_d, cmc_h), (nq, ng)
        assert abs(map_d - map_h) < 1e-12
    with pytest.raises(AssertionError):
        evaluate(None, np.array([9, 9]), np.zeros(10, np.int64), np.zeros(2, np.int64), np.ones(10, np.int64),
                 indices=engine.rank_rows(torch.randn(2, 10).cuda()))


def test_row_argsort_is_a_total_order_on_special_values():
    """NaN (either sign), +-inf, -0.0 / +0.0 and repeated values: the device argsort returns a
    permutation in np.argsort(kind='stable') order (NaN last, -0 == +0), never an out-of-range
    index, and grl_rank_metrics on top of it matches the host loop."""
    from grl_amd import engine
    from grl_amd.reid.evaluator.eva_functions import evaluate
    rng = np.random.default_rng(9)
    for n in (5, 48, 64, 300, 1025):
        d = rng.standard_normal((7, n)).astype(np.float32)
        d[0, :] = np.nan
        d[1, ::3] = np.nan
        d[1, 1::5] = -np.nan
        d[2, ::2] = np.inf
        d[2, 1::4] = -np.inf
        d[3, ::2] = 0.0
        d[3, 1::2] = -0.0
        d[4, :] = np.round(d[4] * 2) / 2               # many exact ties
        d[5, n // 2] = np.nan
        idx = engine.rank_rows(torch.from_numpy(d).cuda()).cpu().numpy()
        assert np.array_equal(np.sort(idx, axis=1), np.tile(np.arange(n), (7, 1)))
        assert np.array_equal(idx, np.argsort(d, axis=1, kind='stable'))
    qp, gp = rng.integers(0, 4, 7), rng.integers(0, 4, n)
    qc, gc = rng.integers(0, 2, 7), rng.integers(0, 2, n)
    cmc_h, map_h = evaluate(d, qp, gp, qc, gc, indices=idx)
    cmc_d, map_d = evaluate(None, qp, gp, qc, gc, indices=engine.rank_rows(torch.from_numpy(d).cuda()))
    assert np.array_equal(cmc_h, cmc_d) and abs(map_h - map_d) < 1e-12


def test_device_re_ranking_matches_reference_golden_and_numpy(golden):
    """k-reciprocal re-ranking on the device (grl_amd/csrc/rerank.hip) against (a) the output of
    the reference's own re_ranking on the stored input matrices and (b) the host numpy
    restatement on larger structured inputs, incl. other (k1, k2, lambda) and k2 == 1."""
    from grl_amd import engine
    from grl_amd.reid.evaluator.rerank import re_ranking
    g = golden('rerank_q16_g120.npz')
    out = re_ranking(torch.from_numpy(g['dist']).cuda(), torch.from_numpy(g['qq']).cuda(),
                     torch.from_numpy(g['gg']).cuda())
    assert out.is_cuda and tuple(out.shape) == (16, 120)
    assert np.abs(out.cpu().numpy() - g['final']).max() < 2e-6
    for (nq, ng, k1, k2, lam, seed) in ((48, 600, 20, 6, 0.3, 3), (33, 257, 7, 1, 0.5, 4), (20, 300, 12, 3, 0.1, 5)):
        qf, gf, qp, qc, gp, gc = synth_eval_features(nq, ng, seed=seed, n_ids=40, noise=4.0)
        qd, gd = qf.cuda(), gf.cuda()
        d, dqq, dgg = engine.cosin_dist(qd, gd), engine.pairwise_distance_tensor(qd, qd), \
            engine.pairwise_distance_tensor(gd, gd)
        dev = re_ranking(d, dqq, dgg, k1=k1, k2=k2, lambda_value=lam).cpu().numpy()
        host = re_ranking(d.cpu().numpy(), dqq.cpu().numpy(), dgg.cpu().numpy(), k1=k1, k2=k2, lambda_value=lam)
        assert np.abs(dev - host).max() < 2e-6, (nq, ng, k1, k2)
    with pytest.raises(ValueError):
        re_ranking(d, dqq[:5], dgg)


def test_raw_uint8_clips_are_normalised_on_the_device(gpu_models):
    """uint8 clips go straight to the stem, which applies ToTensor +
    Normalize while staging its input patch -- features are bit-identical to the float path
    (fp32 and bf16-storage pipelines, eval and train mode), and grl_normalize_u8 is
    bit-identical to the host transform."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    raw = synth_clips(3, 4, seed=11, raw=True).cuda()
    flt = synth_clips(3, 4, seed=11).cuda()
    assert torch.equal(engine.normalize_u8(raw), flt)
    assert torch.equal(engine.extract_features(cnn, siam, raw), engine.extract_features(cnn, siam, flt))
    with engine.math_mode('bf16s'):
        assert torch.equal(engine.extract_features(cnn, siam, raw), engine.extract_features(cnn, siam, flt))
    xu_r, xc_r = cnn(raw)
    xu_f, xc_f = cnn(flt)
    assert torch.equal(xu_r, xu_f) and torch.equal(xc_r, xc_f)
    with pytest.raises(engine._lib.GrlHipError):
        cnn(raw.to(torch.int16))


def test_hip_graph_replay_is_bit_identical(gpu_models):
    """The captured-graph extractor replays the same launches: bit-identical features,
    new inputs are picked up, a second shape gets its own graph."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    gx = engine.GraphedExtractor(cnn, siam)
    a, b = synth_clips(4, 2, seed=50).cuda(), synth_clips(4, 2, seed=51).cuda()
    fa, fb = gx(a), gx(b)
    assert torch.equal(fa, engine.extract_features(cnn, siam, a))
    assert torch.equal(fb, engine.extract_features(cnn, siam, b))
    assert not torch.equal(fa, fb)
    c = synth_clips(2, 3, seed=52).cuda()
    assert torch.equal(gx(c), engine.extract_features(cnn, siam, c))
    assert len(gx._graphs) == 2


def test_trl_two_stream_order_is_bit_identical_to_single_stream(gpu_models, monkeypatch):
    """The two TRL directions run on two HIP streams (engine._TrlFork): same kernels, per-direction
    scratch, f_corr contributions summed at the join -- the features equal the single-stream order
    bit for bit, in fp32 and bf16-storage, eager and replayed from a HIP graph, call after call."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    clips = [synth_clips(b, t, seed=60 + b).cuda() for b, t in ((5, 4), (2, 3), (32, 4))]
    for mode in ('f32', 'bf16s'):
        with engine.math_mode(mode):
            monkeypatch.setattr(engine, 'TRL_STREAMS', False)
            want = [engine.extract_features(cnn, siam, c) for c in clips]
            monkeypatch.setattr(engine, 'TRL_STREAMS', True)
            for _ in range(3):
                for c, w in zip(clips, want):
                    assert torch.equal(engine.extract_features(cnn, siam, c), w)
    gx = engine.GraphedExtractor(cnn, siam)
    monkeypatch.setattr(engine, 'TRL_STREAMS', False)
    want = engine.extract_features(cnn, siam, clips[0])
    monkeypatch.setattr(engine, 'TRL_STREAMS', True)
    assert torch.equal(gx(clips[0]), want) and torch.equal(gx(clips[0]), want)


def test_reference_script_flow_through_dropin(tmp_path):
    """examples/train_synthetic.py follows mars_train.py's call sequence through the drop-in
    `reid`/`utils` packages (DataParallel wrap, module.backbone param groups, SEQTrainer,
    ATTEvaluator, checkpoint save + reload with the 'module.' key prefix)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.path.join(root, 'dropin'))
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'train_synthetic.py'), '--epochs', '1',
                          '--iters', '2', '-b', '4', '--seq_len', '2', '--logs-dir', str(tmp_path)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'Mean AP:' in out.stdout and 'best rank-1 accuracy is' in out.stdout
    ck = torch.load(os.path.join(str(tmp_path), 'cnnmodel_best.pth.tar'), map_location='cpu', weights_only=False)
    assert set(ck) == {'state_dict', 'epoch', 'best_top1'} and len(ck['state_dict']) == 401
    assert all(k.startswith('module.') for k in ck['state_dict'])
    # test_all.py's flow: the checkpoints just written load back (their BN running statistics are
    # two steps old, so the metrics of that run mean nothing), then with the deterministic synthetic
    # weights: rrs-test and dense mode, re-ranking, float and raw-uint8 loaders print the same lines
    ev = os.path.join(root, 'examples', 'eval_synthetic.py')
    base = [sys.executable, ev, '--queries', '8', '--gallery', '40', '--seq_len', '2']
    r = subprocess.run(base + ['--cnn_ckpt', os.path.join(str(tmp_path), 'cnnmodel_best.pth.tar'),
                               '--siamese_ckpt', os.path.join(str(tmp_path), 'siamesemodel_best.pth.tar')],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'Rank-1:' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    seen = {}
    for extra in ([], ['--uint8'], ['--rerank'], ['--dense'], ['--dense', '--uint8']):
        r = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert 'Mean AP:' in r.stdout and 'Rank-1:' in r.stdout
        seen[tuple(extra)] = [l for l in r.stdout.splitlines() if l.startswith(('Mean AP', 'Rank-'))]
    assert seen[()] == seen[('--uint8',)] and seen[('--dense',)] == seen[('--dense', '--uint8')]


def test_trainer_step_matches_reference_trainer_golden_1e3(golden):
    """grl_amd's SEQTrainer on HIP against ONE step of the reference's own SEQTrainer (trainer.py:107-170 `_forward`
    as shipped + loss.backward(); tests/golden/trainer_step_cond_b8t4.npz: B x T = 8 x 4, conditioned CNN weights,
    unit-norm LUTs): the 5-term loss, the model outputs handed to the heads, the gradients that come back into the
    CNN outputs and EVERY parameter gradient of the Siamese head at <= 1e-3 (north_star's figure), the LUT rows of
    the batch identities after the three OIM backwards, the heads' BatchNorm running statistics."""
    import trainer_step_check as TS
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from grl_amd.synthetic import synth_clips_structured
    g = golden('trainer_step_cond_b8t4.npz')
    B, T = int(g['meta.B']), int(g['meta.T'])
    dev = torch.device('cuda:0')
    cnn = _fresh_cnn_conditioned()
    _, siam, siamv = _fresh_models()
    crit_c, crit_u = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev)
    lut_c, lut_u = TS.luts(g)
    crit_c.lut.copy_(lut_c); crit_u.lut.copy_(lut_u)
    trainer = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), crit_c, crit_u, None)
    cnn.train(); siam.train(); siamv.train()
    taps = {}
    h = cnn.register_forward_hook(lambda m, i, o: taps.update(xu=o[0], xc=o[1]))
    loss, p_u, p_v, p_f = trainer._forward([synth_clips_structured(B, T, seed=3).to(dev)], torch.from_numpy(g['pids']).to(dev), 0, 0)
    h.remove()
    taps['xu'].retain_grad(); taps['xc'].retain_grad()
    loss.backward()
    print('trainer step: loss hip %.6f reference %.6f' % (loss.item(), float(g['loss'])))
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    assert [float(p_u), float(p_v), float(p_f)] == list(g['prec'])
    assert TS.rel(taps['xu'].detach().cpu().numpy(), g['x_uncorr']) < 1e-4
    assert TS.rel(taps['xc'].detach().cpu().numpy()[..., ::4], g['x_corr_s4']) < 1e-4
    assert TS.rel(taps['xu'].grad.cpu().numpy(), g['grad.x_uncorr']) < 1e-3
    assert TS.rel(taps['xc'].grad.cpu().numpy()[..., ::4], g['grad.x_corr_s4']) < 1e-3
    TS.check_grads(g, 'gs', {k: p.grad for k, p in siam.named_parameters() if p.grad is not None}, 1e-3, 'HIP')
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in siamv.parameters())     # (as upstream: 'gv' is empty)
    TS.check_grads(g, 'gc', {k: p.grad for k, p in cnn.named_parameters() if p.grad is not None}, 1e-3, 'HIP', cnn_model=True)
    rows = torch.from_numpy(g['lut_rows']).to(dev)
    assert TS.rel(crit_c.lut[rows].cpu().numpy(), g['lut_c1']) < 1e-4 and TS.rel(crit_u.lut[rows].cpu().numpy(), g['lut_u1']) < 1e-4
    for name, m in (('siamese', siam), ('siamese_video', siamv)):
        sd = m.state_dict()
        for k in [k for k in g.files if k.startswith('stat.%s.' % name)]:
            assert TS.rel(sd[k[len('stat.%s.' % name):]].cpu().numpy(), g[k]) < 1e-4, k


def test_trainer_loss_composition_matches_cpu_restatement():
    """SEQTrainer._forward on HIP vs the same 5-term loss assembled on CPU from the oracle's
    train-mode forward (trainer.py:107-170: frame id + clip id (same LUT) + 20 x pair
    verification + batch-hard triplet on the correlated branch, id loss on the uncorrelated
    branch), B x T = 4 x 2, non-zero LUTs."""
    import torch.nn.functional as F
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    from oracle import grl_oracle as O
    cnn, siam, siamv = _fresh_models()
    sd = {k: v.detach().cpu().clone() for k, v in cnn.state_dict().items()}
    ssd = {k: v.detach().cpu().clone() for k, v in siam.state_dict().items()}
    svd = {k: v.detach().cpu().clone() for k, v in siamv.state_dict().items()}
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    lut_c = F.normalize(torch.randn(625, 2048, generator=g), dim=1)
    lut_u = F.normalize(torch.randn(625, 2048, generator=g), dim=1)
    crit_c, crit_u = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev)
    crit_c.lut.copy_(lut_c); crit_u.lut.copy_(lut_u)
    trainer = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), crit_c, crit_u, None)
    cnn.train(); siam.train(); siamv.train()
    clips = synth_clips(4, 2, seed=2)
    pids = torch.tensor([17, 17, 401, 401])
    loss, p_u, p_v, p_f = trainer._forward([clips.to(dev)], pids.to(dev), 0, 0)
    # CPU restatement
    xu, xc = O.grl_forward(sd, clips, train=True)
    l_frame, _ = O.oim_loss(xc.reshape(8, -1), pids.repeat_interleave(2), lut_c.clone(), 30.0, 0.5)
    cls, sout = O.siamese_forward(ssd, xc, train=True)
    target = torch.cat((pids[0::2], pids[1::2]))
    l_vid, _ = O.oim_loss(sout, target, lut_c.clone(), 30.0, 0.5)
    l_tri = O.triplet_soft_batch_hard(sout, target).mean()
    prob = F.softmax(cls.view(-1, 2), dim=-1).view(2, 2, 2)[:, :, 1]
    l_ver, _ = O.pair_loss(prob, pids[0::2], pids[1::2])
    _, vout = O.siamese_video_forward(svd, xu, train=True)
    l_unc, _ = O.oim_loss(vout, target, lut_u.clone(), 30.0, 0.5)
    ref = l_unc + l_frame + l_vid + 20 * l_ver + l_tri
    print('trainer loss hip %.6f cpu %.6f (rel %.2e)' % (loss.item(), ref.item(), abs(loss.item() - ref.item()) / abs(ref.item())))
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))          # (measured 9e-7; was 2e-3 until round 4)


def test_backward_is_linear_in_the_upstream_gradient():
    """Size-independent property of the HIP backward at a BASELINE-like size (B x T = 8 x 4):
    every op of the backward is linear in the incoming gradient and scaling by 2 is exact in
    fp32, so backward(2g) == 2 * backward(g) bit for bit."""
    cnn, _, _ = _fresh_models()
    cnn.train()
    clips = synth_clips(8, 4, seed=6).cuda()
    rg = torch.Generator().manual_seed(1)
    r1, r2 = torch.randn(8, 2048, generator=rg).cuda(), torch.randn(8, 4, 2048, generator=rg).cuda()
    grads = []
    for scale in (1.0, 2.0):
        cnn.zero_grad(set_to_none=True)
        xu, xc = cnn(clips)
        ((xu * r1).sum() * scale + (xc * r2).sum() * scale).backward()
        grads.append({k: p.grad.clone() for k, p in cnn.named_parameters()})
    for k in grads[0]:
        assert torch.equal(grads[1][k], grads[0][k] * 2), k
        assert bool(torch.isfinite(grads[0][k]).all()), k


def test_prefetcher_overlapped_h2d_gives_identical_features(gpu_models):
    """engine.DevicePrefetcher (side-stream H2D of the next batch) feeds the same bytes: features of
    host float32 / uint8 / already-resident batches equal the plain path bit for bit, in order."""
    from grl_amd import engine
    cnn, siam, _ = gpu_models
    batches = [synth_clips(2, 4, seed=20 + i) for i in range(4)]
    raw = [synth_clips(2, 4, seed=20 + i, raw=True) for i in range(4)]
    want = [engine.extract_features(cnn, siam, b.cuda()) for b in batches]
    for src in (batches, [b.pin_memory() for b in batches], raw, [b.cuda() for b in batches]):
        got = []
        for d, pid, cam in engine.DevicePrefetcher(((b, [i], [0]) for i, b in enumerate(src)), 'cuda:0'):
            assert d.is_cuda and pid == [len(got)]
            got.append(engine.extract_features(cnn, siam, d))
        assert len(got) == 4 and all(torch.equal(a, b) for a, b in zip(got, want))
    m = engine.rows_mean(torch.cat(want, 0))
    assert float((m - torch.cat(want, 0).mean(0, keepdim=True)).abs().max()) < 1e-6


def test_train_forward_backward_realistic_batch_vs_float64_oracle():
    """B x T = 16 x 4 (BatchNorm over real batch sizes): HIP forward + backward against the oracle
    run in FLOAT64 on the host -- every parameter gradient in full, not samples.  The same oracle
    in float32 (what the reference computes) gives the scale of acceptable rounding."""
    from oracle import grl_oracle as O
    B, T = 16, 4
    cnn, _, _ = _fresh_models()
    cnn.train()
    rg = np.random.Generator(np.random.PCG64(17))
    r1 = torch.from_numpy(rg.standard_normal((B, 2048)).astype(np.float32))
    r2 = torch.from_numpy(rg.standard_normal((B, T, 2048)).astype(np.float32))
    clips = synth_clips(B, T, seed=5)
    sd0 = {k: v.detach().cpu().clone() for k, v in cnn.state_dict().items()}

    def oracle(dtype):
        sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
        for k, v in sd.items():
            if v.dtype.is_floating_point and 'running' not in k:
                v.requires_grad_(True)
        xu, xc = O.grl_forward(sd, clips.to(dtype), train=True)
        ((xu * r1.to(dtype)).sum() + (xc * r2.to(dtype)).sum()).backward()
        return xu.detach(), xc.detach(), {k: v.grad for k, v in sd.items() if v.dtype.is_floating_point and v.grad is not None}
    torch.set_num_threads(16)
    xu64, xc64, g64 = oracle(torch.float64)
    xu32, xc32, g32 = oracle(torch.float32)
    xu, xc = cnn(clips.cuda())
    ((xu * r1.cuda()).sum() + (xc * r2.cuda()).sum()).backward()
    named = dict(cnn.named_parameters())

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    assert rel(xu.detach().cpu(), xu64) < 2e-3 and rel(xc.detach().cpu(), xc64) < 2e-3
    e_hip = {k: rel(named[k].grad.cpu(), g64[k]) for k in g64 if named[k].grad is not None}
    e_ref = {k: rel(g32[k], g64[k]) for k in e_hip}
    assert len(e_hip) >= 190                                        # every parameter that gets a gradient
    # a few gradients vanish analytically for these weights (both fp32 runs return rounding noise
    # there): judge the keys on which the reference's own fp32 arithmetic is meaningful
    ok = [k for k in e_hip if e_ref[k] < 0.3]
    assert len(ok) >= 180
    qh = np.quantile([e_hip[k] for k in ok], [0.5, 0.9, 1.0])
    qr = np.quantile([e_ref[k] for k in ok], [0.5, 0.9, 1.0])
    ratio = max(e_hip[k] / max(e_ref[k], 1e-4) for k in ok)
    print('full-batch grads vs float64 over %d tensors: HIP median/p90/max %.2e %.2e %.2e; fp32 oracle %.2e %.2e %.2e; '
          'worst per-tensor ratio %.1f' % (len(ok), qh[0], qh[1], qh[2], qr[0], qr[1], qr[2], ratio))
    for k in sorted(ok, key=lambda k: e_hip[k] / max(e_ref[k], 1e-4))[-6:]:
        print('   %-60s hip %.2e  fp32 oracle %.2e' % (k, e_hip[k], e_ref[k]))
    assert qh[0] <= 2.0 * qr[0] + 1e-4 and qh[1] <= 2.0 * qr[1] + 1e-4 and qh[2] <= 3.0 * qr[2] + 1e-3
    assert ratio < 8.0


def test_eval_plan_follows_running_stats_changed_by_train_forward():
    """the train-mode kernels update BatchNorm running statistics through raw device
    pointers, which torch's version counters never see.  eval -> train-mode forwards under no_grad
    (no optimizer step) -> eval must fold the NEW statistics: equal to a freshly built model that
    loads the state_dict, and different from the stale first result."""
    import copy
    from grl_amd import engine
    cnn, siam, _ = _fresh_models()
    clips = synth_clips(2, 2, seed=9).cuda()
    cnn.eval(); siam.eval()
    before = engine.extract_features(cnn, siam, clips).clone()
    graphed = engine.GraphedExtractor(cnn, siam)
    assert torch.equal(graphed(clips), before)
    cnn.train(); siam.train()
    with torch.no_grad():
        for s in (3, 4):
            xu, xc = cnn(synth_clips(2, 2, seed=s).cuda())
            siam(xc)
    cnn.eval(); siam.eval()
    after = engine.extract_features(cnn, siam, clips)
    cnn2, siam2, _ = _fresh_models()
    cnn2.load_state_dict(copy.deepcopy(cnn.state_dict())); siam2.load_state_dict(copy.deepcopy(siam.state_dict()))
    cnn2.eval(); siam2.eval()
    fresh = engine.extract_features(cnn2, siam2, clips)
    assert torch.equal(after, fresh)
    assert not torch.equal(after, before)
    assert torch.equal(graphed(clips), fresh)            # the captured graphs were dropped too


def test_train_step_full_size_32x4_properties():
    """BASELINE configs[1] at full size (B x T = 32 x 4) in train mode, through size-independent
    properties: finite outputs and gradients for every parameter that gets one; unit-norm output
    rows; BatchNorm bookkeeping (num_batches_tracked advances by 1 in the trunk / GCE / tail and by T
    in the TRL memo blocks, grl_model.py:153,167); the backward is linear in the upstream gradient
    (2g -> exactly 2x); and a clip's train-mode OUTPUT does not depend on its position in the batch
    (BatchNorm statistics are permutation invariant up to summation order: 1e-5)."""
    B, T = 32, 4
    cnn = _fresh_cnn_conditioned()
    cnn.train()
    from grl_amd.synthetic import synth_clips_structured
    clips = synth_clips_structured(B, T, seed=8).cuda()
    rg = torch.Generator().manual_seed(2)
    r1, r2 = torch.randn(B, 2048, generator=rg).cuda(), torch.randn(B, T, 2048, generator=rg).cuda()
    grads = []
    for scale in (1.0, 2.0):
        cnn.zero_grad(set_to_none=True)
        xu, xc = cnn(clips)
        (((xu * r1).sum() + (xc * r2).sum()) * scale).backward()
        grads.append({k: p.grad.clone() for k, p in cnn.named_parameters() if p.grad is not None})
    assert bool(torch.isfinite(xu).all()) and bool(torch.isfinite(xc).all())
    assert float((xu.norm(dim=1) - 1).abs().max()) < 1e-5 and float((xc.norm(dim=2) - 1).abs().max()) < 1e-5
    assert len(grads[0]) >= 190
    for k in grads[0]:
        assert bool(torch.isfinite(grads[0][k]).all()), k
        # exact doubling, except where a value sits in the fp32 DENORMAL range (a saturated channel
        # attention gives d*g*a*(1-a) ~ 1e-40: a denormal has fewer significand bits than its double)
        assert float((grads[1][k] - grads[0][k] * 2).abs().max()) <= 1e-36, k
    sd = cnn.state_dict()
    assert int(sd['backbone.base.1.num_batches_tracked']) == 2 and int(sd['corr_bn.num_batches_tracked']) == 2
    assert int(sd['backbone.corr_atte.6.num_batches_tracked']) == 2
    assert int(sd['temporal_learning_block.uncorr_memo_forward.bn1.num_batches_tracked']) == 2 * T
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        xu_p, xc_p = cnn(clips[perm])
    assert float((xu_p - xu.detach()[perm]).abs().max()) < 1e-5
    assert float((xc_p - xc.detach()[perm]).abs().max()) < 1e-5


@pytest.mark.parametrize('mode,out_tol,med_tol,p90_tol', [('bf16x3', 1e-3, 3e-3, 1e-2), ('bf16', 0.5, 0.5, 0.6)])
def test_train_bf16_multiplier_datapaths_against_fp32_path(mode, out_tol, med_tol, p90_tol):
    """The opt-in training datapaths (train_engine.set_math: forward + data-gradient GEMMs on the bf16
    MFMA, split 'bf16x3' or plain 'bf16' operands; weight gradients, BatchNorm and losses stay fp32)
    against the fp32 training path (itself pinned to the reference at 1e-3 above), B x T = 16 x 4:
    outputs (max norm) and every parameter gradient (relative L2).  'bf16x3' is fp32-class (measured:
    outputs 4e-4, gradients 1.6e-3 median).  Plain 'bf16' operands are only smoke-bounded here: on these
    synthetic fixtures the BatchNorm1d layers over a few near-identical rows amplify the 2^-9 operand
    rounding to ~0.3 -- measured, reported, and the reason 'bf16x3' is the recommended fast mode."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 16, 4
    clips = synth_clips_structured(B, T, seed=12).cuda()
    rg = torch.Generator().manual_seed(4)
    r1, r2 = torch.randn(B, 2048, generator=rg).cuda(), torch.randn(B, T, 2048, generator=rg).cuda()
    res = {}
    for m in ('f32', mode):
        cnn = _fresh_cnn_conditioned()
        cnn.train()
        old = TE.set_math(m)
        try:
            xu, xc = cnn(clips)
            ((xu * r1).sum() + (xc * r2).sum()).backward()
        finally:
            TE.set_math(old)
        res[m] = (xu.detach(), xc.detach(), {k: p.grad for k, p in cnn.named_parameters() if p.grad is not None})
    a, b = res[mode], res['f32']
    e_out = max(float((a[0] - b[0]).abs().max() / b[0].abs().max()), float((a[1] - b[1]).abs().max() / b[1].abs().max()))
    errs = np.sort([float((a[2][k] - b[2][k]).norm() / b[2][k].norm().clamp_min(1e-30)) for k in b[2]
                    if float(b[2][k].abs().max()) > 1e-9])
    msg = 'train math %s vs f32: outputs %.1e; gradient L2 error median %.1e p90 %.1e max %.1e' % (
        mode, e_out, np.median(errs), errs[int(.9 * len(errs))], errs[-1])
    print(msg)
    assert e_out < out_tol and np.median(errs) < med_tol and errs[int(.9 * len(errs))] < p90_tol, msg


def test_train_mixed_math_forward_is_the_fp32_forward_and_gradients_stay_fp32_class():
    """train_engine.set_math('mixed'): exact fp32 forward, split-bf16 products in the backward GEMMs.  The forward
    -- outputs, batch statistics, ReLU masks -- must be the 'f32' forward bit for bit, so no mask can flip and
    every weight gradient stays within 1e-3 (relative L2; median 1.4e-5, worst 3.4e-4) of the 'f32' gradient, the one pinned
    to the reference at 1e-3; the reference fixture itself is then checked in this mode."""
    from grl_amd import train_engine as TE
    from grl_amd.synthetic import synth_clips_structured
    B, T = 16, 4
    clips = synth_clips_structured(B, T, seed=12).cuda()
    rg = torch.Generator().manual_seed(4)
    r1, r2 = torch.randn(B, 2048, generator=rg).cuda(), torch.randn(B, T, 2048, generator=rg).cuda()
    res = {}
    for m in ('f32', 'mixed'):
        cnn = _fresh_cnn_conditioned()
        cnn.train()
        old = TE.set_math(m)
        try:
            assert TE.get_math() == m
            xu, xc = cnn(clips)
            ((xu * r1).sum() + (xc * r2).sum()).backward()
        finally:
            TE.set_math(old)
        res[m] = (xu.detach(), xc.detach(), {k: p.grad for k, p in cnn.named_parameters() if p.grad is not None},
                  {k: v.clone() for k, v in cnn.state_dict().items() if 'running' in k})
    a, b = res['mixed'], res['f32']
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(a[3][k], b[3][k]) for k in b[3])
    # (biases in front of a batch-statistics BatchNorm have a mathematically zero gradient: both runs hold rounding
    # noise there -- compared on the scale of the layer's weight gradient instead of their own)
    scale = {k: float(b[2][k].norm()) for k in b[2]}
    for k in b[2]:
        if k.endswith('.bias') and k[:-5] + '.weight' in scale:
            scale[k] = max(scale[k], 1e-3 * scale[k[:-5] + '.weight'])
    errs = sorted((float((a[2][k] - b[2][k]).norm()) / max(scale[k], 1e-30), k) for k in b[2] if scale[k] > 0)
    vals = np.array([e for e, _ in errs])
    msg = 'train math mixed vs f32: gradient L2 error median %.1e p90 %.1e max %.1e (%s)' % (
        np.median(vals), vals[int(.9 * len(vals))], vals[-1], errs[-1][1])
    print(msg)
    # weights (>= 2-D): median 5e-5, worst 1e-3 (measured 1.4e-5 / 3.4e-4); BatchNorm gains / shifts are sums over all pixels of signed terms that cancel to
    # ~1e-2 of their magnitude, which amplifies the 2^-16 product error: 1e-2 (the golden test's outlier bound)
    w_err = [e for e, k in errs if b[2][k].dim() >= 2]
    v_err = [e for e, k in errs if b[2][k].dim() < 2]
    assert max(w_err) < 1e-3 and np.median(w_err) < 5e-5 and np.median(v_err) < 1e-4 and max(v_err) < 1e-2, msg


def test_train_two_stream_trl_is_deterministic_and_matches_single_stream(monkeypatch):
    """Train mode runs the two TRL directions -- forward and, through the tape's stream tags, backward -- on two
    HIP streams, and weight gradients on a third (train_engine.WGRAD_STREAM: same launches, bit-identical gradients
    with or without it).  Run to run the step is bit-reproducible (fixed per-direction accumulation, sums at the join);
    against the single-stream order only the association of three sums changes (gradients of x_uncorr, GAP(x_corr)
    and the initial memo are accumulated per direction first): forward bit-identical, weight gradients within 1e-4."""
    from grl_amd import engine
    from grl_amd.synthetic import synth_clips_structured
    B, T = 8, 4
    clips = synth_clips_structured(B, T, seed=21).cuda()
    rg = torch.Generator().manual_seed(9)
    r1, r2 = torch.randn(B, 2048, generator=rg).cuda(), torch.randn(B, T, 2048, generator=rg).cuda()

    from grl_amd import train_engine as TE

    def run(two, wstream=True):
        monkeypatch.setattr(engine, 'TRL_STREAMS', two)
        monkeypatch.setattr(TE, 'WGRAD_STREAM', wstream)
        cnn = _fresh_cnn_conditioned()
        cnn.train()
        xu, xc = cnn(clips)
        ((xu * r1).sum() + (xc * r2).sum()).backward()
        torch.cuda.synchronize()
        return xu.detach(), xc.detach(), {k: p.grad.clone() for k, p in cnn.named_parameters() if p.grad is not None}
    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    d = run(True, wstream=False)       # weight gradients on the launch stream: the same launches, the same bits
    assert all(torch.equal(a[2][k], d[2][k]) for k in a[2])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    scale = {k: float(c[2][k].norm()) for k in c[2]}
    for k in c[2]:                     # zero-gradient biases in front of a batch-statistics BatchNorm: layer scale
        if k.endswith('.bias') and k[:-5] + '.weight' in scale:
            scale[k] = max(scale[k], 1e-3 * scale[k[:-5] + '.weight'])
    errs = sorted((float((a[2][k] - c[2][k]).norm()) / max(scale[k], 1e-30), k) for k in c[2])
    w_err = [e for e, k in errs if c[2][k].dim() >= 2]
    v_err = [e for e, k in errs if c[2][k].dim() < 2]
    msg = 'two-stream vs single-stream gradients: weights median %.1e max %.1e; vectors median %.1e max %.1e (%s)' % (
        np.median(w_err), max(w_err), np.median(v_err), max(v_err), errs[-1][1])
    print(msg)
    # (BatchNorm gains / shifts: sums over all pixels that cancel to ~1 % of their magnitude amplify the last-bit changes)
    assert max(w_err) < 1e-4 and np.median(v_err) < 1e-5 and max(v_err) < 1e-2, msg
