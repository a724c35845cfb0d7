import sys, time, os, io, contextlib
sys.path.insert(0, '.')
import torch
from bench import build_models
from oracle import grl_oracle as O
from grl_amd.synthetic import synth_clips
cnn, siam, sd, ssd = build_models('cpu')
clips = synth_clips(8, 4, seed=0)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    O.extract_features(sd, ssd, clips[:2])
    t0 = time.time(); O.extract_features(sd, ssd, clips); dt = time.time() - t0
    print('threads', th, 'clips/s %.2f' % (8 / dt), flush=True)
