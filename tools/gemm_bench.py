#!/usr/bin/env python
"""Per-shape timing of grl_conv_gemm_f32 on the shapes of one eval step (B x T = 32 x 4).
  python tools/gemm_bench.py [tile ...]     e.g.  128x128 128x64 64x64
Each tile is forced through grl_gemm_force_tile; 'auto' uses the library heuristic.  The first rows of a cold
process run on ramping clocks: warm the box with one throw-away run before comparing columns."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import engine

SHAPES = [  # (M, N, K, conv, calls per step, epilogue: res?)
    (4096, 2048, 2048, None, 8, False), (16384, 2048, 2048, None, 2, False),
    (16384, 512, 4608, (16, 8, 512, 16, 8, 3, 3, 1, 1), 3, False),
    (16384, 2048, 512, None, 3, True), (16384, 256, 2304, (16, 8, 256, 16, 8, 3, 3, 1, 1), 5, False),
    (262144, 256, 64, None, 4, True), (4096, 512, 2048, None, 8, False), (4096, 2048, 512, None, 8, True),
    (16384, 1024, 256, None, 6, True), (262144, 64, 576, (64, 32, 64, 64, 32, 3, 3, 1, 1), 3, False),
    (65536, 512, 128, None, 4, True), (16384, 2048, 1024, None, 1, False),
    (65536, 128, 1152, (32, 16, 128, 32, 16, 3, 3, 1, 1), 3, False), (16384, 1024, 2048, None, 1, False),
    (16384, 512, 2048, None, 2, False), (16384, 256, 1024, None, 6, False), (65536, 128, 512, None, 3, False),
    (4096, 512, 512, None, 8, False), (262144, 64, 256, None, 2, False), (262144, 128, 256, None, 1, False),
    (65536, 256, 512, None, 1, False), (16384, 512, 1024, None, 1, False), (262144, 64, 64, None, 1, False),
]


def bench(tile, iters=20):
    math = 0
    if ':' in tile:                       # e.g. bf16x3:256x128
        m, tile = tile.split(':')
        math = {'f32': 0, 'bf16': 1, 'bf16x3': 3, 'bf16s': 2}[m]
    elif tile in ('bf16x3', 'bf16', 'f32', 'bf16s'):
        math = {'f32': 0, 'bf16': 1, 'bf16x3': 3, 'bf16s': 2}[tile]
        tile = 'auto'
    from grl_amd import _lib
    if tile == 'auto':
        _lib.load().grl_gemm_force_tile(0, 0)
    else:
        _lib.load().grl_gemm_force_tile(*(int(v) for v in tile.split('x')))      # (GRL_GEMM_TILE is read once per process)
    dev = torch.device('cuda:0')
    res = {}
    for (M, N, K, conv, calls, has_res) in SHAPES:
        if conv is None:
            a = torch.randn(M, K, device=dev)
        else:
            H, W, Cc = conv[0], conv[1], conv[2]
            a = torch.randn(M // (conv[3] * conv[4]) * H * W, Cc, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        y = torch.empty(M, N, device=dev)
        sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if has_res else None
        if math == 2:
            if K % 64:
                continue
            a, w, y = a.bfloat16(), w.bfloat16(), y.bfloat16()
            r = r.bfloat16() if r is not None else None
        for _ in range(2):
            engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, res=r, relu=True, conv=conv, math=math)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            engine.gemm(a, w, y, M, N, K, scale=sc, shift=sh, res=r, relu=True, conv=conv, math=math)
        e1.record()
        torch.cuda.synchronize()
        res[(M, N, K, conv is not None)] = (e0.elapsed_time(e1) / iters, calls, 2.0 * M * N * K)
    return res


if __name__ == '__main__':
    tiles = sys.argv[1:] or ['auto']
    allres = {t: bench(t) for t in tiles}
    print('%-34s' % 'shape (M,N,K,conv)' + ''.join('%22s' % t for t in tiles))
    tot = {t: 0.0 for t in tiles}
    for key in allres[tiles[0]]:
        line = '%-34s' % str(key)
        for t in tiles:
            if key not in allres[t]:
                line += '%22s' % '-'
                continue
            ms, calls, fl = allres[t][key]
            tot[t] += ms * calls
            line += '   %7.3f ms %6.1f TF' % (ms, fl / ms / 1e9)
        print(line)
    print('%-34s' % 'per-step total (ms)' + ''.join('%22.3f' % tot[t] for t in tiles))
