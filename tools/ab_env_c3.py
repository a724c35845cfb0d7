#!/usr/bin/env python
"""tools/ab_env.py for BASELINE configs[2] (bf16 storage, 64 clips x 8 frames): A/B of one env knob read at load time.
   python tools/ab_env_c3.py GRL_GEMM_DMA_CONV 0 1 [rounds]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
from grl_amd import engine
from grl_amd.synthetic import synth_clips
dev = torch.device('cuda:0')
cnn, siam, _, _ = bench.build_models(dev)
clips = synth_clips(64, 8, seed=0).to(dev)
with engine.math_mode('bf16s'):
    for _ in range(5): f = engine.extract_features(cnn, siam, clips)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): f = engine.extract_features(cnn, siam, clips)
    torch.cuda.synchronize()
print('MS %%.4f %%.10f' %% ((time.perf_counter() - t0) / 30 * 1e3, float(f.double().abs().sum())))
''' % R
key, a, b = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
res = {a: [], b: []}
for _ in range(rounds):
    for v in (a, b):
        out = subprocess.run([sys.executable, '-c', CODE], env=dict(os.environ, **{key: v}), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('MS')]
        if not line:
            print(out.stderr[-2000:]); sys.exit(1)
        res[v].append(line[0].split()[1:])
for v in (a, b):
    ms = sorted(float(x[0]) for x in res[v])
    print('%s=%s: median %.3f ms min %.3f  checksum %s' % (key, v, ms[len(ms) // 2], ms[0], res[v][0][1]))
