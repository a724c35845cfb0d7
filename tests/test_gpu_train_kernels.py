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


def test_oim_update_kernel_matches_sequential_loop():
    from grl_amd.reid.loss import OIMLoss
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    x = F.normalize(torch.randn(24, 2048), dim=1)
    y = torch.tensor([5, 5, 5, 5, 9, 9, 9, 9, 5, 5, 2, 2, 7, 7, 7, 7, 9, 9, 0, 0, 5, 2, 7, 0])
    cpu, gpu = OIMLoss(2048, 12, scalar=30, momentum=0.5), OIMLoss(2048, 12, scalar=30, momentum=0.5).to(dev)
    init = F.normalize(torch.randn(12, 2048), dim=1)
    cpu.lut.copy_(init); gpu.lut.copy_(init)
    xa, xb = x.clone().requires_grad_(True), x.clone().to(dev).requires_grad_(True)
    la, _ = cpu(xa, y); la.backward()
    lb, _ = gpu(xb, y.to(dev)); lb.backward()
    assert abs(la.item() - lb.item()) < 1e-5
    assert _rel(xb.grad.cpu().numpy(), xa.grad.numpy()) < 1e-5
    assert _rel(gpu.lut.cpu().numpy(), cpu.lut.numpy()) < 1e-5
