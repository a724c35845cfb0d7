#!/usr/bin/env python
"""kblock (K-blocked accumulation, the train-mode forward / data-gradient GEMM form) vs the single chain, per shape,
with and without the per-channel statistics epilogue.   python tools/kblock_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine
dev = torch.device('cuda:0')
SHAPES = [(4096, 512, 2048), (4096, 2048, 512), (4096, 512, 512), (4096, 2048, 2048), (16384, 2048, 2048), (16384, 512, 2048),
          (16384, 1024, 256), (262144, 256, 64), (65536, 512, 128), (16384, 256, 1024)]
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
    row = []
    for kb, st in ((False, False), (True, False), (True, True), (False, True)):
        def run():
            return engine.gemm(a, w, y, M, N, K, kblock=kb, stats=True if st else None)
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 6
        row.append('%s%s %6.1f us %5.1f TF' % ('kblock' if kb else 'chain ', '+stats' if st else '      ', ms * 1e3, 2.0 * M * N * K / ms / 1e9))
    print('%-22s' % str((M, N, K)), ' | '.join(row), flush=True)
