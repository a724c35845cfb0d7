import sys, torch
sys.path.insert(0, '/root/repo')
from grl_amd import engine
dev = torch.device('cuda:0')
for (M, N, K, kb) in ((65536, 256, 64, True), (65536, 256, 64, False), (16384, 512, 128, True), (4096, 1024, 256, True), (65536, 128, 512, True), (4096, 2048, 512, True), (65536, 256, 1024, True)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; y = torch.empty(M, N, device=dev)
    _, slab = engine.gemm(a, w, y, M, N, K, stats=True, kblock=kb)
    ref = a.double() @ w.double().t()
    e = ((y.double() - ref).norm() / ref.norm()).item()
    s = slab.double().sum(0)
    es = ((s[0] - ref.sum(0)).norm() / ref.sum(0).norm()).item(); eq = ((s[1] - (ref * ref).sum(0)).norm() / (ref * ref).sum(0).norm()).item()
    print((M, N, K, kb), 'rows', slab.shape[0], 'y err %.1e sum err %.1e sq err %.1e' % (e, es, eq))
