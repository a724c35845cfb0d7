#!/usr/bin/env python
"""One GEMM shape under each forced tile (child processes: GRL_GEMM_TILE is read once per process), fp32 or bf16
storage (`f32+stats` / `bf16s+stats`: the train-mode forward form), alone and with a twin of itself on a second stream (the two TRL directions run their memo-block GEMMs
concurrently).   python tools/gemm_tile_ab.py bf16s 8192 512 2048 [8192 2048 512 ...]"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('GRL_TILE_AB_CHILD'):
    import torch
    from grl_amd import engine
    math = sys.argv[1]
    M, N, K = (int(v) for v in sys.argv[2:5])
    dev = torch.device('cuda:0')
    with_res = math.endswith('+res')                      # the eval form of the expansion layers: + residual, ReLU
    math = math.replace('+res', '')
    stats = math.endswith('+stats')                       # the train-mode forward form: raw output + statistics slab
    math = math.replace('+stats', '')
    dt = torch.bfloat16 if math == 'bf16s' else torch.float32
    mk = lambda: (torch.randn(M, K, device=dev).to(dt), (torch.randn(N, K, device=dev) * 0.05).to(dt), torch.empty(M, N, device=dev, dtype=dt))
    a, w, y = mk(); a2, w2, y2 = mk()
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    m = {'bf16s': engine.MATH_BF16S, 'bf16x3': engine.MATH_BF16X3}.get(math, engine.MATH_F32)     # (bf16x3: the `mixed` backward's datapath)
    s2 = torch.cuda.Stream()
    kw = dict(stats=True) if stats else dict(scale=sc, shift=sh, relu=True)
    if with_res:
        kw['res'] = torch.randn(M, N, device=dev).to(dt)
    def one(): engine.gemm(a, w, y, M, N, K, math=m, **kw)
    def two():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2): engine.gemm(a2, w2, y2, M, N, K, math=m, **kw)
        engine.gemm(a, w, y, M, N, K, math=m, **kw)
        torch.cuda.current_stream().wait_stream(s2)
    out = []
    for fn, mult in ((one, 1), (two, 2)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        out.append('%.1f us (%.0f TF)' % (us, mult * 2.0 * M * N * K / us / 1e6))
    print('alone %s | two streams %s' % tuple(out))
    sys.exit(0)
math = sys.argv[1]
shapes = [tuple(sys.argv[i:i + 3]) for i in range(2, len(sys.argv), 3)]
for shp in shapes:
    for tile in ('', '128x128', '128x64', '64x64') + (('256x256',) if math.startswith('bf16s') else ()):
        env = dict(os.environ, GRL_TILE_AB_CHILD='1', GRL_GEMM_TILE='' if tile == '256x256' else tile,
                   GRL_GEMM_BF16_256='1' if tile == '256x256' else ('0' if tile else os.environ.get('GRL_GEMM_BF16_256', '-1')))
        r = subprocess.run([sys.executable, __file__, math] + list(shp), env=env, capture_output=True, text=True)
        print('%s %-18s tile %-8s %s' % (math, 'x'.join(shp), tile or 'auto', r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]))
