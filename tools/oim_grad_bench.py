import sys, torch, ctypes as C
sys.path.insert(0, '/root/repo')
from grl_amd import _lib
from grl_amd._lib import ptr
lib = _lib.load(); dev = torch.device('cuda:0')
for n in (128, 32):
    dl = torch.randn(n, 625, device=dev); lut = torch.randn(625, 2048, device=dev); g = torch.ones((), device=dev); dx = torch.empty(n, 2048, device=dev)
    for _ in range(5): lib.grl_oim_grad(ptr(dl), 625, ptr(lut), ptr(g), C.c_float(30.0), ptr(dx), n, 625, 2048, _lib.stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.grl_oim_grad(ptr(dl), 625, ptr(lut), ptr(g), C.c_float(30.0), ptr(dx), n, 625, 2048, _lib.stream())
    e1.record(); torch.cuda.synchronize()
    ref = 30.0 * dl.double() @ lut.double()
    print('oim_grad n=%d: %.1f us, rel err %.1e' % (n, e0.elapsed_time(e1) * 20, ((dx.double() - ref).norm() / ref.norm()).item()))
