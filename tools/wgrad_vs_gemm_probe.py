#!/usr/bin/env python
"""Counter workload: the same 16384 x 2048 x 2048 product as a forward GEMM (K-contiguous operands) and as a weight
gradient (reduction index = the operands' ROW), 10 launches each, for `rocprofv3 --pmc ...` passes (tools/README.md).
   rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum -d out -- python3 tools/wgrad_vs_gemm_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine, train_engine as TE

dev = torch.device('cuda:0')
M, N, K = 16384, 2048, 2048
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
y = torch.empty(M, N, device=dev)
dz = torch.randn(M, N, device=dev)
dw = torch.zeros(N, K, device=dev)
for _ in range(10):
    engine.gemm(a, w, y, M, N, K)
for _ in range(10):
    TE.wgrad(dz, a, dw, M, N, K)
torch.cuda.synchronize()
