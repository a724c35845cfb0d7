import sys, torch, time
sys.path.insert(0, '/root/repo')
from grl_amd import _lib
from grl_amd._lib import ptr
lib = _lib.load()
dev = torch.device('cuda:0')
for (n, H, W) in ((3, 64, 32), (5, 256, 128), (2, 36, 20)):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 3, H, W, generator=g).to(dev)
    Ho, Wo = H // 2, W // 2
    dz = torch.randn(n * Ho * Wo, 64, generator=g).to(dev)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).double()
    xd = x.double().requires_grad_(False)
    out = torch.nn.functional.conv2d(xd, conv.weight, stride=2, padding=3)
    ref = torch.autograd.grad(out, conv.weight, dz.double().view(n, Ho, Wo, 64).permute(0, 3, 1, 2))[0]
    for b16 in (0, 1):
        dzz = dz.bfloat16() if b16 else dz
        refb = ref if not b16 else torch.autograd.grad(torch.nn.functional.conv2d(xd, conv.weight, stride=2, padding=3), conv.weight, dzz.double().view(n, Ho, Wo, 64).permute(0, 3, 1, 2))[0]
        dw = torch.full((64, 3, 7, 7), 0.5, device=dev)
        ws = torch.empty(lib.grl_stem_wgrad_workspace_floats(n, H, W), device=dev)
        _lib.check(lib.grl_stem_wgrad(ptr(x), ptr(dzz), b16, ptr(dw), ptr(ws), n, H, W, 1, _lib.stream()))
        torch.cuda.synchronize()
        err = ((dw.double() - 0.5 - refb).norm() / refb.norm()).item()
        print(n, H, W, 'bf16' if b16 else 'f32', 'rel err %.2e' % err)
# timing at 32x4 and 64x8
for n in (128, 512):
    x = torch.randn(n, 3, 256, 128, device=dev); dz = torch.randn(n * 128 * 64, 64, device=dev)
    dw = torch.zeros(64, 3, 7, 7, device=dev); ws = torch.empty(lib.grl_stem_wgrad_workspace_floats(n, 256, 128), device=dev)
    for b16 in (0, 1):
        dzz = dz.bfloat16() if b16 else dz
        for _ in range(3): lib.grl_stem_wgrad(ptr(x), ptr(dzz), b16, ptr(dw), ptr(ws), n, 256, 128, 1, _lib.stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.grl_stem_wgrad(ptr(x), ptr(dzz), b16, ptr(dw), ptr(ws), n, 256, 128, 1, _lib.stream())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('n=%d %s: %.3f ms  (%.1f TFLOP/s)' % (n, 'bf16 dz' if b16 else 'f32 dz', ms, 2.0 * n * 8192 * 64 * 147 / ms / 1e9))
