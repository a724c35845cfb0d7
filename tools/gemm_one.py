import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine
M, N, K, iters = [int(v) for v in sys.argv[1:5]]
dev = torch.device('cuda:0')
a, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.empty(M, N, device=dev)
sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
for _ in range(iters):
    engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, relu=True)
torch.cuda.synchronize()
