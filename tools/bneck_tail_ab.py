#!/usr/bin/env python
"""A/B of the fused bottleneck tail (fuse_bf16.hip) against the two launches it replaces, at the pixel counts of
BASELINE configs[2] (64 clips x 8 frames).   python tools/bneck_tail_ab.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_fuse_bf16 import _C

dev = torch.device('cuda:0')
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
BF = torch.bfloat16
F32 = len(sys.argv) > 2 and sys.argv[2] == 'f32'      # python tools/bneck_tail_ab.py 128 f32: the exact-fp32 twin at 32 x 4


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (P, C4, Pn, px) in ((64, 256, 64, 64 * 32), (64, 256, 128, 64 * 32), (128, 512, 128, 32 * 16), (128, 512, 256, 32 * 16)):
    M = frames * px
    g = torch.Generator().manual_seed(1)
    c3, c1 = _C(C4, P, g, dev), _C(Pn, C4, g, dev)
    if F32 and Pn == 256:
        continue
    dt = torch.float32 if F32 else BF
    t2 = torch.randn(M, P, generator=g).clamp_min(0).to(dev).to(dt)
    res = torch.randn(M, C4, generator=g).to(dev).to(dt)
    y0 = torch.empty(M, C4, dtype=dt, device=dev)
    u0 = torch.empty(M, Pn, dtype=dt, device=dev)
    if F32:
        def unfused():
            engine.gemm(t2, c3.w, y0, M, C4, P, scale=c3.scale, shift=c3.shift, res=res, relu=True)
            engine.gemm(y0, c1.w, u0, M, Pn, C4, scale=c1.scale, shift=c1.shift, relu=True)

        def conv3_only():
            engine.gemm(t2, c3.w, y0, M, C4, P, scale=c3.scale, shift=c3.shift, res=res, relu=True)

        def fused():
            engine.bneck_tail_f32(t2, c3, res, c1, M)
        a, b, c = timeit(unfused), timeit(conv3_only), timeit(fused)
        fl = 2.0 * M * C4 * (P + Pn)
        print('f32 P %3d C4 %3d Pn %3d M %8d: unfused %7.1f us (conv3 alone %7.1f)  fused %7.1f us = %.1f TFLOP/s, %.2f TB/s compulsory'
              % (P, C4, Pn, M, a, b, c, fl / c / 1e6, M * (P + 2 * C4 + Pn) * 4 / c / 1e6), flush=True)
        continue

    def unfused():
        engine.gemm(t2, c3.wb(), y0, M, C4, P, scale=c3.scale, shift=c3.shift, res=res, relu=True, math=engine.MATH_BF16S)
        engine.gemm(y0, c1.wb(), u0, M, Pn, C4, scale=c1.scale, shift=c1.shift, relu=True, math=engine.MATH_BF16S)

    def conv3_only():
        engine.gemm(t2, c3.wb(), y0, M, C4, P, scale=c3.scale, shift=c3.shift, res=res, relu=True, math=engine.MATH_BF16S)

    def fused():
        engine.bneck_tail_bf16(t2, c3, res, c1, M)

    a, b, c = timeit(unfused), timeit(conv3_only), timeit(fused)
    byts = M * (P + 2 * C4 + Pn) * 2
    print('P %3d C4 %3d Pn %3d M %8d: unfused %7.1f us (conv3 alone %7.1f)  fused %7.1f us = %.2f TB/s of compulsory traffic'
          % (P, C4, Pn, M, a, b, c, byts / c / 1e6), flush=True)
