#!/usr/bin/env python
"""Same-process A/B of engine.FUSE_TRL_SQDIFF on the headline step (B x T = 32 x 4, fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grl_amd import engine
from grl_amd.synthetic import synth_clips
dev = torch.device('cuda:0')
cnn, siam, _, _ = bench.build_models(dev)
clips = synth_clips(32, 4, seed=0).to(dev)
res = {True: [], False: []}
for rnd in range(7):
    for flag in (True, False):
        engine.FUSE_TRL_SQDIFF = flag
        for _ in range(3):
            engine.extract_features(cnn, siam, clips)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            engine.extract_features(cnn, siam, clips)
        torch.cuda.synchronize()
        res[flag].append((time.perf_counter() - t0) / 20 * 1e3)
for flag in (True, False):
    v = sorted(res[flag])
    print('fused' if flag else 'unfused', 'median %.3f ms min %.3f' % (v[len(v) // 2], v[0]))
