#!/usr/bin/env python
"""Small-batch latency of the eval path: eager launches vs HIP-graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_models
from grl_amd import engine
from grl_amd.synthetic import synth_clips
dev = torch.device('cuda:0')
cnn, siam, _, _ = build_models(dev)
gx = engine.GraphedExtractor(cnn, siam)
import sys as _s
if len(_s.argv) > 1:
    engine.set_math(_s.argv[1])
for b in (1, 8, 32):
    clips = synth_clips(b, 4, seed=b).to(dev)
    for fn, name in ((lambda: engine.extract_features(cnn, siam, clips), 'eager'), (lambda: gx(clips), 'graph')):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print('B=%2d T=4 %-5s %7.3f ms/step  %8.1f clip-features/s' % (b, name, dt * 1e3, b / dt))
