"""Train-mode kernels (BN forward/backward, weight/data gradients, pooling backward)
against torch autograd on CPU, through the C ABI on a real MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _cl(x):      # NCHW -> channels-last rows
    return x.permute(0, 2, 3, 1).contiguous().view(-1, x.shape[1])


@pytest.mark.parametrize('cin,cout,k,stride,H,W,n,relu,res', [
    (64, 64, 3, 1, 16, 8, 4, True, False), (128, 128, 3, 2, 16, 16, 3, True, False),
    (256, 512, 1, 2, 8, 8, 4, False, False), (64, 256, 1, 1, 8, 4, 5, True, True),
    (512, 128, 1, 1, 16, 8, 2, True, False)])
def test_conv_bn_op_forward_backward(cin, cout, k, stride, H, W, n, relu, res):
    from grl_amd import train_engine as TE
    dev = torch.device('cuda:0')
    torch.manual_seed(cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(n, cin, H, W, requires_grad=True)
    z = bn(conv(x))
    r = torch.randn_like(z) if res else None
    y = z + r if res else z
    y = F.relu(y) if relu else y
    gout = torch.randn_like(y)
    y.backward(gout)
    ref = dict(y=y.detach(), dx=x.grad, dw=conv.weight.grad, dg=bn.weight.grad, db=bn.bias.grad,
               rm=bn.running_mean.clone(), rv=bn.running_var.clone())
    conv_d, bn_d = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout)
    conv_d.load_state_dict(conv.state_dict())
    bn_d.load_state_dict({k_: (torch.zeros_like(v) if 'running_mean' in k_ else torch.ones_like(v) if 'running_var' in k_
                               else v) for k_, v in bn.state_dict().items()})
    bn_d.num_batches_tracked.zero_()
    conv_d.to(dev); bn_d.to(dev)
    tp = TE.Tape(dev)
    xd = _cl(x.detach()).to(dev)
    rd = _cl(r).to(dev) if res else None
    a, Ho, Wo, _ = TE.conv_bn(tp, xd, n, H, W, conv_d, bn_d, relu, res=rd)
    got = a.view(n, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
    assert _rel(got.numpy(), ref['y'].numpy()) < 1e-4
    assert _rel(bn_d.running_mean.cpu().numpy(), ref['rm'].numpy()) < 1e-4
    assert _rel(bn_d.running_var.cpu().numpy(), ref['rv'].numpy()) < 1e-4
    assert int(bn_d.num_batches_tracked) == 1
    tp.g[id(a)] = _cl(gout).to(dev)
    tp.backward()
    dx = tp.take(xd).view(n, H, W, cin).permute(0, 3, 1, 2).cpu()
    assert _rel(dx.numpy(), ref['dx'].numpy()) < 2e-4
    assert _rel(tp.pgrad(conv_d.weight).cpu().numpy(), ref['dw'].numpy()) < 2e-4
    assert _rel(tp.pgrad(bn_d.weight).cpu().numpy(), ref['dg'].numpy()) < 2e-4
    assert _rel(tp.pgrad(bn_d.bias).cpu().numpy(), ref['db'].numpy()) < 2e-4
    if res:
        dres = tp.take(rd).view(n, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
        assert _rel(dres.numpy(), (gout * (ref['y'] > 0)).numpy()) < 1e-6


def test_stem_pool_forward_backward():
    """Stem 7x7 conv + BN(train) + ReLU + max-pool, weight gradient through im2col."""
    from grl_amd import train_engine as TE
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    n, H, W = 2, 64, 32
    conv, bn = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
    x = torch.randn(n, 3, H, W)
    a = F.relu(bn(conv(x)))
    p = F.max_pool2d(a, 3, 2, 1)
    gout = torch.randn_like(p)
    p.backward(gout)

    class M(nn.Module):
        pass
    m = M(); m.backbone = M()
    c2, b2 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
    c2.load_state_dict(conv.state_dict())
    m.backbone.base = nn.Sequential(c2, b2, nn.ReLU(), nn.MaxPool2d(3, 2, 1), nn.Sequential(), nn.Sequential(),
                                    nn.Sequential(), nn.Sequential()).to(dev)
    tp = TE.Tape(dev)
    xd = x.to(dev)
    out = TE.trunk_train(tp, m, xd)
    got = out.view(n, H // 4, W // 4, 64).permute(0, 3, 1, 2).cpu()
    assert _rel(got.numpy(), p.detach().numpy()) < 1e-4
    tp.g[id(out)] = _cl(gout).to(dev)
    tp.backward()
    assert _rel(tp.pgrad(c2.weight).cpu().numpy(), conv.weight.grad.numpy()) < 2e-4
    assert _rel(tp.pgrad(b2.weight).cpu().numpy(), bn.weight.grad.numpy()) < 2e-4
    assert _rel(tp.pgrad(b2.bias).cpu().numpy(), bn.bias.grad.numpy()) < 2e-4


def test_maxpool_backward_ties():
    """Post-ReLU maps are full of exact ties (zeros): gradient goes to the FIRST maximum."""
    from grl_amd import engine
    from grl_amd._lib import ptr
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    n, H, W, Cc = 2, 12, 10, 8
    x = F.relu(torch.randn(n, Cc, H, W)).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    g = torch.randn_like(y)
    y.backward(g)
    xd, gd = _cl(x.detach()).to(dev), _cl(g).to(dev)
    dx = torch.empty_like(xd)
    engine._call('grl_maxpool3x3s2_bwd', ptr(xd), ptr(gd), ptr(dx), n, H, W, Cc)
    got = dx.view(n, H, W, Cc).permute(0, 3, 1, 2).cpu()
    assert _rel(got.numpy(), x.grad.numpy()) < 1e-6


def test_wgrad_dense_shapes():
    from grl_amd import train_engine as TE
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(1)
    for M, N, K, k_out in ((1000, 64, 160, 147), (4096, 128, 2048, 0), (2, 2048, 128, 0), (300, 32, 256, 0)):
        dz = rng.standard_normal((M, N)).astype(np.float32)
        x = rng.standard_normal((M, K)).astype(np.float32)
        ko = k_out or K
        dw = torch.ones(N, ko, device=dev)
        dzd, xd = torch.from_numpy(dz).to(dev), torch.from_numpy(x).to(dev)
        TE.wgrad(dzd, xd, dw, M, N, K, k_out=k_out, accumulate=1)
        ref = 1.0 + dz.astype(np.float64).T @ x.astype(np.float64)[:, :ko]
        assert _rel(dw.cpu().numpy(), ref) < 1e-5, (M, N, K)


def test_oim_matches_reference_golden(golden, synth_models):
    """OIMLoss on HIP (logits GEMM + grl_softmax_ce + grl_oim_grad + grl_oim_update) against
    outputs of the REFERENCE's OIM.forward / OIM.backward bodies and OIMLoss.forward
    (reid/loss/oim.py:14-27,46-53; tests/golden/oim.npz): zero LUT, unit-norm LUT, duplicate
    labels, and one training step's two calls on ONE LUT (trainer.py:126,138) in the autograd
    engine's order -- the clip-level update lands before the frame-level backward reads the LUT."""
    import copy
    from grl_amd.reid.loss import OIMLoss
    dev = torch.device('cuda:0')
    g = golden('oim.npz')
    for name in ('zero', 'unit', 'dup'):
        x0, y, lut0 = g[name + '.x'], g[name + '.y'], g[name + '.lut0']
        crit = OIMLoss(x0.shape[1], lut0.shape[0], scalar=30, momentum=0.5).to(dev)
        crit.lut.copy_(torch.from_numpy(lut0))
        x = torch.from_numpy(x0).to(dev).requires_grad_(True)
        loss, logits = crit(x, torch.from_numpy(y).to(dev))
        (loss * float(g[name + '.upstream'])).backward()
        assert abs(loss.item() - float(g[name + '.loss'])) < 1e-4, name
        assert _rel(logits.cpu().numpy(), g[name + '.logits']) < 1e-4, name
        assert _rel(x.grad.cpu().numpy(), g[name + '.grad_x']) < 1e-4, name
        assert _rel(crit.lut.cpu().numpy(), g[name + '.lut1']) < 1e-4, name
    siam = copy.deepcopy(synth_models[1]).to(dev).train()
    xc = torch.from_numpy(g['step.x_corr']).to(dev).requires_grad_(True)
    ids = torch.from_numpy(g['step.ids']).to(dev)
    B, T = xc.shape[:2]
    crit = OIMLoss(2048, g['step.lut0'].shape[0], scalar=30, momentum=0.5).to(dev)
    crit.lut.copy_(torch.from_numpy(g['step.lut0']))
    l_frame, _ = crit(xc.view(B * T, -1), ids.repeat_interleave(T))          # trainer.py:126
    tv = ids.view(B // 2, -1)
    _, pooled = siam(xc)                                                     # trainer.py:137
    l_vid, _ = crit(pooled, torch.cat((tv[:, 0], tv[:, 1])))                 # trainer.py:138
    (l_frame + l_vid).backward()
    assert abs(l_frame.item() - float(g['step.loss_frame'])) < 1e-4
    assert abs(l_vid.item() - float(g['step.loss_vid'])) < 1e-4
    assert _rel(pooled.detach().cpu().numpy(), g['step.pooled']) < 1e-4
    assert _rel(xc.grad.cpu().numpy(), g['step.grad_x_corr']) < 1e-4
    assert _rel(crit.lut.cpu().numpy(), g['step.lut1']) < 1e-4


def test_oim_update_kernel_matches_sequential_loop():
    """OIMLoss on HIP (logits GEMM + grl_softmax_ce + grl_oim_grad + grl_oim_update) vs the
    oracle's restatement of oim.py:14-27,46-53 with repeated labels (order-dependent updates)."""
    from grl_amd.reid.loss import OIMLoss
    from oracle import grl_oracle as O
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    x = F.normalize(torch.randn(24, 2048), dim=1)
    y = torch.tensor([5, 5, 5, 5, 9, 9, 9, 9, 5, 5, 2, 2, 7, 7, 7, 7, 9, 9, 0, 0, 5, 2, 7, 0])
    gpu = OIMLoss(2048, 12, scalar=30, momentum=0.5).to(dev)
    init = F.normalize(torch.randn(12, 2048), dim=1)
    lut_o = init.clone(); gpu.lut.copy_(init)
    xa, xb = x.clone().requires_grad_(True), x.clone().to(dev).requires_grad_(True)
    la, logits_a = O.oim_loss(xa, y, lut_o, 30.0, 0.5); (la * 0.7).backward()
    lb, logits_b = gpu(xb, y.to(dev)); (lb * 0.7).backward()
    assert not logits_b.requires_grad
    assert abs(la.item() - lb.item()) < 1e-5
    assert _rel(logits_b.cpu().numpy(), logits_a.detach().numpy()) < 1e-5
    assert _rel(xb.grad.cpu().numpy(), xa.grad.numpy()) < 1e-5
    assert _rel(gpu.lut.cpu().numpy(), lut_o.numpy()) < 1e-5


@pytest.mark.parametrize('n,c,D,weighted', [(160, 625, 2048, False), (6, 5, 32, False), (33, 625, 2048, True)])
def test_softmax_ce_and_oim_grad(n, c, D, weighted):
    """grl_softmax_ce / grl_oim_grad against F.cross_entropy autograd (mean reduction, optional
    class weights, one ignored label), and the arg-max hit count against topk."""
    from grl_amd.reid.loss import OIMLoss
    dev = torch.device('cuda:0')
    torch.manual_seed(n + c)
    x = F.normalize(torch.randn(n, D), dim=1)
    y = torch.randint(0, c, (n,))
    w = (torch.rand(c) + 0.5) if weighted else None
    lut = F.normalize(torch.randn(c, D), dim=1)
    xa = x.clone().requires_grad_(True)
    logits = xa.mm(lut.t()) * 30.0
    la = F.cross_entropy(logits, y, weight=w); la.backward()
    crit = OIMLoss(D, c, scalar=30.0, momentum=0.5, weight=None if w is None else w.to(dev)).to(dev)
    crit.lut.copy_(lut)
    xb = x.clone().to(dev).requires_grad_(True)
    lb, lg = crit(xb, y.to(dev)); lb.backward()
    assert abs(la.item() - lb.item()) < 2e-5 * max(1.0, abs(la.item()))
    assert _rel(lg.cpu().numpy(), logits.detach().numpy()) < 1e-5
    assert _rel(xb.grad.cpu().numpy(), xa.grad.numpy()) < 2e-5
    # raw kernel: hit count + an out-of-range label is ignored like torch's ignore_index
    from grl_amd import _lib
    lib = _lib.load()
    y2 = y.clone(); y2[0] = -100
    lg_d, y2_d = logits.detach().to(dev).contiguous(), y2.to(dev)
    loss, hits = torch.zeros((), device=dev), torch.zeros((), device=dev)
    ws = torch.empty(2 * n, device=dev)
    _lib.check(lib.grl_softmax_ce(lg_d.data_ptr(), c, y2_d.data_ptr(), None, n, c, loss.data_ptr(), hits.data_ptr(),
                                  None, c, ws.data_ptr(), _lib.stream()), 'grl_softmax_ce')
    ref = F.cross_entropy(logits.detach(), y2)
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    assert int(hits.item()) == int((logits.detach().argmax(1)[1:] == y2[1:]).sum())


def test_loss_block_matches_reference_golden(golden):
    """TripletLoss('soft', True) and PairLoss on HIP against the outputs of the reference's own
    modules (tests/golden/losses.npz)."""
    from grl_amd.reid.loss import TripletLoss, PairLoss
    dev = torch.device('cuda:0')
    g = golden('losses.npz')
    tri = TripletLoss('soft', True)(torch.from_numpy(g['feat']).to(dev), torch.from_numpy(g['ids']).to(dev))
    assert np.allclose(tri.cpu().numpy(), g['triplet'], rtol=1e-5, atol=1e-6)
    loss, prec = PairLoss()(torch.from_numpy(g['score']).to(dev), torch.from_numpy(g['tp']).to(dev),
                            torch.from_numpy(g['tg']).to(dev))
    assert abs(loss.item() - float(g['pair_loss'])) < 1e-6
    assert abs(float(prec) - float(g['pair_prec'])) < 1e-6


@pytest.mark.parametrize('ids,margin', [
    ([0, 0, 1, 1, 2, 2, 3, 3, 0, 1, 2, 3, 4, 4, 5, 5], 'soft'),
    ([0, 0, 1, 1, 2, 3, 3, 3], 'soft'),               # id 2 has no positive: masked maximum, no gradient
    ([7, 7, 7, 7], 'soft'),                           # no negative at all: min over dist + 1e5
    ([0, 0, 1, 1, 2, 2, 3, 3], 0.3)])
def test_triplet_forward_backward(ids, margin):
    from grl_amd.reid.loss import TripletLoss
    from oracle import grl_oracle as O
    dev = torch.device('cuda:0')
    torch.manual_seed(len(ids))
    ids = torch.tensor(ids)
    f = torch.randn(len(ids), 2048) * 0.05
    fa = f.clone().requires_grad_(True)
    if margin == 'soft':
        la = O.triplet_soft_batch_hard(fa, ids)
    else:
        diff = fa.unsqueeze(1) - fa.unsqueeze(0)
        dist = (diff.pow(2).sum(2) + 1e-12).sqrt()
        same = ids.unsqueeze(1).eq(ids.unsqueeze(0))
        pos = same & ~torch.eye(len(ids), dtype=torch.bool)
        la = torch.clamp((dist * pos.float()).max(1)[0] - (dist + 1e5 * same.float()).min(1)[0] + margin, min=0)
    r = torch.rand(len(ids)) + 0.5
    (la * r).sum().backward()
    fb = f.clone().to(dev).requires_grad_(True)
    lb = TripletLoss(margin, True)(fb, ids.to(dev))
    (lb * r.to(dev)).sum().backward()
    assert np.allclose(lb.detach().cpu().numpy(), la.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert _rel(fb.grad.cpu().numpy(), fa.grad.numpy()) < 2e-5


def test_pair_prob_and_bce_forward_backward():
    from grl_amd.reid.loss import PairLoss
    from grl_amd.reid.loss.pairloss import pair_prob
    from oracle import grl_oracle as O
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    n = 16
    scores = torch.randn(n, n, 2) * 3
    scores[0, 0] = torch.tensor([120.0, -120.0])       # saturated: exercises the log / denominator clamps
    tp = torch.arange(n) // 2
    tg = torch.arange(n) // 2
    tg[3] = 99
    sa = scores.clone().requires_grad_(True)
    prob_a = F.softmax(sa.view(-1, 2), dim=-1).view(n, n, 2)[:, :, 1]
    la, pa = O.pair_loss(prob_a, tp, tg); (la * 20).backward()
    sb = scores.clone().to(dev).requires_grad_(True)
    prob_b = pair_prob(sb)
    lb, pb = PairLoss()(prob_b, tp.to(dev), tg.to(dev)); (lb * 20).backward()
    assert _rel(prob_b.detach().cpu().numpy(), prob_a.detach().numpy()) < 1e-6
    assert abs(la.item() - lb.item()) < 1e-5 * max(1.0, abs(la.item()))
    assert abs(float(pa) - float(pb)) < 1e-6
    assert _rel(sb.grad.cpu().numpy(), sa.grad.numpy()) < 2e-5


def test_oim_out_of_range_labels_do_not_touch_memory():
    """A label outside [0, num_classes) is ignored by the cross entropy (torch's ignore_index
    convention) and by the LUT update: nothing is read or written out of bounds."""
    from grl_amd.reid.loss import OIMLoss
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    crit = OIMLoss(64, 6, scalar=10, momentum=0.5).to(dev)
    crit.lut.copy_(F.normalize(torch.randn(6, 64), dim=1))
    before = crit.lut.clone()
    x = F.normalize(torch.randn(5, 64), dim=1).to(dev).requires_grad_(True)
    y = torch.tensor([2, 10 ** 6, -100, 2, 5], device=dev)
    loss, _ = crit(x, y)
    loss.backward()
    ref = F.cross_entropy((x.detach().cpu().mm(before.cpu().t()) * 10)[[0, 3, 4]], torch.tensor([2, 2, 5]))
    assert abs(loss.item() - ref.item()) < 1e-5
    assert torch.isfinite(x.grad).all() and float(x.grad[1].abs().max()) == 0 and float(x.grad[2].abs().max()) == 0
    changed = (crit.lut != before).any(1).cpu().tolist()
    assert changed == [False, False, True, False, False, True]


@pytest.mark.parametrize('cin,cout,kh,kw,stride,pad,H,W,n', [
    (64, 64, 3, 3, 1, 1, 16, 8, 5),       # same-size window, Wo divides 32: the linear-source fast path; 640 pixels
    (128, 96, 3, 3, 1, 1, 32, 16, 3),     # ... Wo = 16, N ends inside a tile
    (64, 40, 3, 3, 1, 1, 64, 32, 1),      # ... Wo = 32
    (64, 64, 3, 3, 1, 1, 10, 6, 7),       # same size but Wo = 6: the generic pixel walk
    (128, 72, 3, 3, 2, 1, 16, 16, 3),     # stride 2
    (256, 64, 1, 1, 2, 0, 8, 8, 9),       # 1x1 / s2 downsample window
    (64, 64, 3, 3, 1, 1, 4, 8, 9),        # 32-pixel images: a stage spans a whole image (C must be a multiple of 64)
])
def test_wgrad_conv_windows_fp32(cin, cout, kh, kw, stride, pad, H, W, n):
    """fp32 weight gradient of an implicit-GEMM convolution, dW[o][c][tap] = sum_m dz[m][o] * x[window(m, tap)][c]: the
    kernel's stage items come from per-item pointers (same-size stride-1 windows: a source that is linear in the pixel
    index, with the column test fixed per item and the row test following the pixel) or from the generic pixel walk;
    out-of-image taps, rows past the pixel range and columns past N / K read a zero chunk.  Operands sit inside
    NaN-poisoned allocations: any read outside the window or past the range shows."""
    from grl_amd import train_engine as TE
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(cin + 7 * H + W)
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    M, K = n * Ho * Wo, kh * kw * cin
    x = rng.standard_normal((n, H, W, cin)).astype(np.float32)
    dz = rng.standard_normal((M, cout)).astype(np.float32)
    xp = np.zeros((n, H + 2 * pad, W + 2 * pad, cin), np.float64)
    xp[:, pad:pad + H, pad:pad + W] = x
    cols = np.empty((n, Ho, Wo, kh * kw, cin), np.float64)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, :, ky * kw + kx] = xp[:, ky:ky + stride * Ho:stride, kx:kx + stride * Wo:stride]
    ref = dz.astype(np.float64).T @ cols.reshape(M, K)                     # [cout][tap][c]
    ref = ref.reshape(cout, kh * kw, cin).transpose(0, 2, 1).reshape(cout, K)       # dW leaves in torch layout [cout][c][kh][kw]
    guard = 4096
    bx = torch.full((n * H * W * cin + 2 * guard,), float('nan'), device=dev)
    bz = torch.full((M * cout + 2 * guard,), float('nan'), device=dev)
    bx[guard:guard + x.size] = torch.from_numpy(x.reshape(-1)).to(dev)
    bz[guard:guard + dz.size] = torch.from_numpy(dz.reshape(-1)).to(dev)
    xd = bx[guard:guard + x.size].view(n * H * W, cin)
    dzd = bz[guard:guard + dz.size].view(M, cout)
    dw = torch.zeros(cout, K, device=dev)
    TE.wgrad(dzd, xd, dw, M, cout, K, conv=(H, W, cin, Ho, Wo, kh, kw, stride, pad), accumulate=0)
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    assert _rel(got, ref) < 2e-6, _rel(got, ref)
    dw2 = torch.zeros(cout, K, device=dev)
    TE.wgrad(dzd, xd, dw2, M, cout, K, conv=(H, W, cin, Ho, Wo, kh, kw, stride, pad), accumulate=0)
    assert torch.equal(dw, dw2)                           # private slabs, fixed reduction order: run-to-run identical


@pytest.mark.parametrize('M,N,K', [(4, 2048, 128), (4, 128, 2048), (2, 32, 2048), (7, 64, 64), (33, 96, 160), (4, 1024, 512)])
def test_wgrad_tiny_M_ignores_memory_past_the_operands(M, N, K):
    """dW = dz^T X with a handful of rows (TRL channel MLP: M = clips; verification head: M = pairs):
    the operands sit inside poisoned allocations, so a read past row M-1 (stage padding) shows."""
    from grl_amd import train_engine as TE
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(M * 131 + N + K)
    dz = rng.standard_normal((M, N)).astype(np.float32)
    x = rng.standard_normal((M, K)).astype(np.float32)
    ref = dz.astype(np.float64).T @ x.astype(np.float64)
    outs = []
    for poison in (1e30, float('nan'), 0.0):
        bz = torch.full((M + 64, N), poison, device=dev); bx = torch.full((M + 64, K), poison, device=dev)
        bz[:M] = torch.from_numpy(dz).to(dev); bx[:M] = torch.from_numpy(x).to(dev)
        dw = torch.zeros(N, K, device=dev)
        TE.wgrad(bz[:M], bx[:M], dw, M, N, K, accumulate=0)
        outs.append(dw.cpu().numpy())
        assert np.isfinite(outs[-1]).all(), poison
        assert _rel(outs[-1], ref) < 1e-5, poison
    assert np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize('M,N,K,conv', [(4096, 128, 2048, None), (1000, 256, 128, None), (40, 2048, 128, None),
                                        (2 * 16 * 8, 128, 9 * 128, (16, 8, 128, 16, 8, 3, 3, 1, 1)),
                                        (2 * 8 * 8, 256, 9 * 128, (16, 16, 128, 8, 8, 3, 3, 2, 1))])
def test_wgrad_bf16_datapaths(M, N, K, conv):
    """Weight gradient on the bf16 MFMA (GrlWgrad.math): transpose reads of [m][out] bf16 planes.
    On bf16-representable operands 'bf16x3' and 'bf16' reproduce the exact fp32 kernel up to fp32
    summation order; on general fp32 operands 'bf16x3' stays at ~1e-5 of the float64 product and
    'bf16' at the 2^-8 operand rounding."""
    from grl_amd import train_engine as TE
    from grl_amd._lib import MATH_F32, MATH_BF16, MATH_BF16X3
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(M + N + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]

    def run(dz, x, math):
        dw = torch.zeros((N, cin, 3, 3) if conv else (N, K), device=dev)
        TE.wgrad(dz, x, dw, M, N, K, conv=conv, accumulate=0, math=math)
        return dw
    for representable in (True, False):
        dz = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
        x = torch.from_numpy(np.maximum(rng.standard_normal((rows_in, cin)), -0.5).astype(np.float32))
        if representable:
            dz, x = dz.bfloat16().float(), x.bfloat16().float()
        dzd, xd = dz.to(dev), x.to(dev)
        ref32 = run(dzd, xd, MATH_F32).cpu().double()
        for math, tol in ((MATH_BF16X3, 2e-6 if representable else 3e-5), (MATH_BF16, 2e-6 if representable else 2e-2)):
            got = run(dzd, xd, math).cpu().double()
            err = float((got - ref32).abs().max() / ref32.abs().max())
            assert err < tol, (representable, math, err)


@pytest.mark.parametrize('M,Cc', [(4096, 64), (1024, 128), (777 * 4, 512)])
def test_bn_backward_mask_recomputed_from_z_is_the_activation_mask(M, Cc):
    """y = relu(bn(z)) without a residual: grl_bn_bwd recomputes the ReLU mask from z with the forward's own
    (z - mean) * scale + beta instead of reading the activation -- dz, dgamma, dbeta must be the ones the activation
    mask gives, bit for bit (values straddling zero included: beta shifts half of them negative)."""
    from grl_amd import train_engine as TE
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(M + Cc)
    bn = nn.BatchNorm1d(Cc).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5); bn.bias.copy_(torch.randn(Cc, generator=g) * 0.3)
    z = (torch.randn(M, Cc, generator=g) * 2 + 0.7).to(dev)
    dy = torch.randn(M, Cc, generator=g).to(dev)
    rows = TE._lib.load().grl_col_stats_rows(M)
    slab = torch.empty(rows, 2, Cc, device=dev)
    TE._call('grl_col_stats', TE.ptr(z), TE.ptr(slab), M, Cc, Cc, TE.ptr(z))
    st = TE.bn_finalize(slab, rows, Cc, M, bn, dev, pivot=z)
    a = torch.empty_like(z)
    TE.bn_apply(z, st, None, a, M, Cc, True)
    assert 0.2 < float((a > 0).float().mean()) < 0.8
    out = []
    for from_z in (False, True):
        dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        dz = TE.bn_backward(dy, z, a, st, bn.weight, dg, db, M, Cc, mask_from_z=from_z)
        out.append((dz, dg, db))
    assert all(torch.equal(x, y) for x, y in zip(*out))


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W", [(3, 64, 32), (2, 256, 128), (2, 36, 20), (1, 16, 32)])
@pytest.mark.parametrize("dz_bf16", [0, 1])
def test_stem_weight_gradient_without_im2col_matches_autograd(n, H, W, dz_bf16):
    """grl_stem_wgrad: dW of the 7x7/s2/p3 stem conv (resnets1.py:106-107) straight from the NCHW clip -- persistent
    workgroups, patch + dz tile in LDS, B operand through the k -> offset table -- against torch's float64 autograd:
    exact-fp32 products, so 2e-6; ragged tiles (36 x 20: partial 8 x 16 tiles), accumulation into a non-zero dW, bf16 dz
    converted exactly; and the same result twice (fixed slab order)."""
    from grl_amd import _lib
    from grl_amd._lib import ptr
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n * 1000 + H)
    x = torch.randn(n, 3, H, W, generator=g).to(dev)
    Ho, Wo = H // 2, W // 2
    dz = torch.randn(n * Ho * Wo, 64, generator=g).to(dev)
    if dz_bf16:
        dz = dz.bfloat16()
    w = torch.zeros(64, 3, 7, 7, dtype=torch.float64, device=dev, requires_grad=True)
    out = torch.nn.functional.conv2d(x.double(), w, stride=2, padding=3)
    ref = torch.autograd.grad(out, w, dz.double().view(n, Ho, Wo, 64).permute(0, 3, 1, 2))[0]
    outs = []
    for _ in range(2):
        dw = torch.full((64, 3, 7, 7), 0.25, device=dev)
        ws = torch.empty(lib.grl_stem_wgrad_workspace_floats(n, H, W), device=dev)
        _lib.check(lib.grl_stem_wgrad(ptr(x), ptr(dz), dz_bf16, ptr(dw), ptr(ws), n, H, W, 1, _lib.stream()))
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    err = ((outs[0].double() - 0.25 - ref).norm() / ref.norm()).item()
    assert err < 2e-6, err
    dw = torch.full((64, 3, 7, 7), 7.0, device=dev)                        # accumulate = 0 overwrites
    _lib.check(lib.grl_stem_wgrad(ptr(x), ptr(dz), dz_bf16, ptr(dw), ptr(ws), n, H, W, 0, _lib.stream()))
    assert ((dw.double() - ref).norm() / ref.norm()).item() < 2e-6


@pytest.mark.gpu
def test_oim_top1_read_out_equals_reference_accuracy():
    """SEQTrainer._top1: the precision taken from the cross-entropy launch (rows whose arg-max is the label) equals
    the reference's accuracy(output.data, target.data)[0] (eva_functions.py:118-131: topk / eq / sum), ties included."""
    from grl_amd.reid.loss import OIMLoss
    from grl_amd.reid.evaluator import accuracy
    from grl_amd.reid.train import SEQTrainer
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    crit = OIMLoss(64, 37, scalar=30, momentum=0.5).to(dev)
    crit.lut.copy_(torch.nn.functional.normalize(torch.randn(37, 64, generator=g), dim=1))
    crit.lut[5] = crit.lut[4]                                   # a tie between classes 4 and 5 for every row
    x = torch.nn.functional.normalize(torch.randn(48, 64, generator=g), dim=1).to(dev)
    y = torch.randint(0, 37, (48,), generator=g).to(dev)
    y[:6] = torch.tensor([4, 5, 4, 5, 4, 5])
    x[:6] = crit.lut[4]                                         # rows whose two best classes tie exactly
    loss, logits = crit(x, y)
    want = accuracy(logits.data, y.data)[0]
    got = SEQTrainer._top1(logits, y)
    assert float(got) == float(want), (float(got), float(want))
    assert 0.0 < float(got) < 1.0


@pytest.mark.parametrize('M,N,K,conv', [(262144, 256, 64, None), (65536, 512, 128, None), (65536 + 64, 256, 64, None),
                                        (65536 + 40, 256, 128, None), (65536, 1024, 256, None), (65536, 512, 512, None),
                                        (65536 + 72, 512, 256, None),
                                        (8 * 64 * 32, 256, 9 * 32, (64, 32, 32, 64, 32, 3, 3, 1, 1))])
@pytest.mark.parametrize('math', ['f32', 'bf16x3', 'bf16'])
def test_statistics_gemm_on_the_wide_tile_keeps_the_small_tiles_bits(M, N, K, conv, math):
    """The train-forward statistics GEMMs run on the 128 x 128 tile with their partial sums formed in the order of the
    tile the rule used before (64 x 64: one slab row per 64 rows; 128 x 64): outputs AND statistics slabs bit for bit
    those of GRL_GEMM_WIDE_STATS=0, including row counts that are not a multiple of 64 / 128 -- so no parity fixture
    can move (any other summation order re-rolls the ReLU flips at B = 4)."""
    import os
    from grl_amd import engine
    from grl_amd._lib import MATH_F32, MATH_BF16X3, MATH_BF16
    dev = torch.device('cuda:0')
    g = torch.Generator(dev).manual_seed(M + N + K)
    cin = K if conv is None else conv[2]
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    a = torch.randn(rows_in, cin, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.1
    res = {}
    was = os.environ.get('GRL_GEMM_WIDE_STATS')
    try:
        for mode in ('0', '1'):
            os.environ['GRL_GEMM_WIDE_STATS'] = mode
            y = torch.empty(M, N, device=dev)
            _, slab = engine.gemm(a, w, y, M, N, K, stats=True, kblock=(math == 'f32'), conv=conv,
                                  math={'f32': MATH_F32, 'bf16x3': MATH_BF16X3, 'bf16': MATH_BF16}[math])
            torch.cuda.synchronize()
            res[mode] = (y, slab.clone())
    finally:
        if was is None:
            os.environ.pop('GRL_GEMM_WIDE_STATS', None)
        else:
            os.environ['GRL_GEMM_WIDE_STATS'] = was
    assert res['0'][1].shape == res['1'][1].shape
    assert torch.equal(res['0'][0], res['1'][0])
    assert torch.equal(res['0'][1], res['1'][1])
    ref = a.double() @ w.double().t() if conv is None else None
    if ref is not None:
        s = res['1'][1].double().sum(0)
        tol = {'f32': 1e-5, 'bf16x3': 1e-4, 'bf16': 2e-2}[math]
        assert float((s[0] - ref.sum(0)).norm() / ref.sum(0).norm()) < tol
        assert float((s[1] - (ref * ref).sum(0)).norm() / (ref * ref).sum(0).norm()) < tol


@pytest.mark.parametrize('math', ['f32', 'bf16s'])
def test_bn_finalize_inside_the_apply_pass_is_bit_identical(math):
    """Round 6 (train_bnfuse.hip): for BatchNorms with <= 64 statistics-slab rows the finalize runs inside the apply launch
    (forward: grl_bn_finalize_apply; backward: inside grl_bn_bwd* / grl_bn_bwd_finish*), every workgroup reducing the slab
    columns of its own 64 channels in slab_totals' order.  A whole train step (4 x 4 clips: layers 2-4, GCE and TRL are all
    eligible, with residuals, mask bits, masks recomputed from z, fused GEMM-epilogue reduces) must give the SAME BITS with
    the form on and off: outputs, BatchNorm running statistics, every parameter gradient."""
    from grl_amd import train_engine as TE
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_clips_structured, synth_state_dict
    import contextlib, io

    def run(on):
        with contextlib.redirect_stdout(io.StringIO()):
            cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
        cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'))
        cnn = cnn.cuda().train()
        was = TE.set_bn_finapply(on)
        old = TE.set_math(math)
        try:
            xu, xc = cnn(synth_clips_structured(4, 4, seed=3).cuda())
            g = torch.Generator(device='cpu').manual_seed(1)
            (xu * torch.randn(xu.shape, generator=g).cuda()).sum().add((xc * torch.randn(xc.shape, generator=g).cuda()).sum()).backward()
            torch.cuda.synchronize()
        finally:
            TE.set_math(old)
            TE.set_bn_finapply(was)
        grads = {n: p.grad.clone() for n, p in cnn.named_parameters() if p.grad is not None}
        stats = {n: b.clone() for n, b in cnn.named_buffers()}
        return xu.detach().clone(), xc.detach().clone(), grads, stats

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[2].keys() == b[2].keys() and len(a[2]) > 150
    bad = [n for n in a[2] if not torch.equal(a[2][n], b[2][n])]
    assert not bad, bad[:5]
    bad = [n for n in a[3] if not torch.equal(a[3][n], b[3][n])]
    assert not bad, bad[:5]
