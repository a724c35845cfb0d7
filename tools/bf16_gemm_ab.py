#!/usr/bin/env python
"""A/B of the two GRL_MATH_BF16S GEMM kernels (128x128 register-staged vs 256x256 LDS-DMA) in ONE
process, interleaved rounds, on the shapes of BASELINE configs[2] (64 clips x 8 frames), plus a
bit-level comparison of their outputs.   python tools/bf16_gemm_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine, _lib
SHAPES_ALL = [  # (M, N, K, conv, calls/step, residual)
    (8192, 2048, 2048, None, 16, False), (65536, 2048, 2048, None, 2, False),
    (65536, 512, 4608, (16, 8, 512, 16, 8, 3, 3, 1, 1), 3, False),
    (65536, 256, 2304, (16, 8, 256, 16, 8, 3, 3, 1, 1), 5, False),
    (65536, 2048, 512, None, 3, True), (8192, 512, 2048, None, 16, False), (8192, 2048, 512, None, 16, True),
    (65536, 1024, 256, None, 6, True), (65536, 2048, 1024, None, 1, False), (65536, 1024, 2048, None, 1, False),
    (65536, 512, 2048, None, 2, False), (65536, 256, 1024, None, 6, False), (8192, 512, 512, None, 16, False),
    (262144, 512, 128, None, 4, True), (262144, 128, 1152, (32, 16, 128, 32, 16, 3, 3, 1, 1), 3, False),
    (262144, 128, 512, None, 3, False), (65536, 512, 1024, None, 1, False),
    (65536, 256, 2304, (32, 16, 256, 16, 8, 3, 3, 2, 1), 1, False),
    (65536, 1024, 512, (32, 16, 512, 16, 8, 1, 1, 2, 0), 1, False),
]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = SHAPES_ALL[:int(sys.argv[2])] if len(sys.argv) > 2 else SHAPES_ALL
lib = _lib.load()
dev = torch.device('cuda:0')
tot = {0: 0.0, 1: 0.0}
print('%-58s %22s %22s  equal' % ('shape (M,N,K,conv) x calls', '128x128 family', '256x256 LDS-DMA'))
for (M, N, K, conv, calls, has_res) in SHAPES:
    rows_in = M if conv is None else (M // (conv[3] * conv[4])) * conv[0] * conv[1]
    cin = K if conv is None else conv[2]
    a = torch.randn(rows_in, cin, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).bfloat16() if has_res else None
    ys = {m: torch.empty(M, N, device=dev, dtype=torch.bfloat16) for m in (0, 1)}
    times = {0: [], 1: []}
    for rd in range(rounds + 1):
        for m in (0, 1):
            lib.grl_gemm_bf16_tile_mode(m)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                engine.gemm(a, w, ys[m], M, N, K, scale=sc, shift=sh, res=r, relu=True, conv=conv, math=2)
            e1.record(); torch.cuda.synchronize()
            if rd:
                times[m].append(e0.elapsed_time(e1) / 3)
    lib.grl_gemm_bf16_tile_mode(-1)
    fl = 2.0 * M * N * K
    med = {m: sorted(times[m])[len(times[m]) // 2] for m in (0, 1)}
    for m in (0, 1):
        tot[m] += med[m] * calls
    print('%-58s %9.3f ms %7.0f TF %9.3f ms %7.0f TF  %s' % (str((M, N, K, conv)) + ' x%d' % calls, med[0], fl / med[0] / 1e9,
                                                           med[1], fl / med[1] / 1e9, bool(torch.equal(ys[0], ys[1]))))
print('per-step total of these shapes: %.3f ms vs %.3f ms' % (tot[0], tot[1]))
