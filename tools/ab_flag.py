#!/usr/bin/env python
"""Same-process A/B of a module-level engine flag on the headline step (interleaved rounds).
   python tools/ab_flag.py TRL_STREAMS [math] [B] [T]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from grl_amd import engine
from grl_amd.synthetic import synth_clips
flag = sys.argv[1]
math = sys.argv[2] if len(sys.argv) > 2 else 'f32'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
T = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device('cuda:0')
cnn, siam, _, _ = bench.build_models(dev)
clips = synth_clips(B, T, seed=0).to(dev)
def run(v, n):
    setattr(engine, flag, v)
    with engine.math_mode(math):
        for _ in range(3): f = engine.extract_features(cnn, siam, clips)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f = engine.extract_features(cnn, siam, clips)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(f.double().abs().sum())
res = {False: [], True: []}
for _ in range(5):
    for v in (False, True):
        res[v].append(run(v, 20))
for v in (False, True):
    ms = sorted(r[0] for r in res[v])
    print('%s=%s %s B=%d T=%d: median %.3f ms  min %.3f  checksum %.10f' % (flag, v, math, B, T, ms[2], ms[0], res[v][0][1]))
