#!/usr/bin/env python
"""Informational: the oracle's plain torch ops (= what the reference would execute)
run on the SAME MI355X through stock PyTorch-ROCm (MIOpen / rocBLAS), eval path,
B x T = 32 x 4.  Not part of the product or of bench.py's JSON line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_models, B, T
from oracle import grl_oracle as O
from grl_amd.synthetic import synth_clips

dev = torch.device('cuda:0')
cnn, siam, sd, ssd = build_models(dev)
sdd = {k: v.to(dev) for k, v in sd.items()}
ssdd = {k: v.to(dev) for k, v in ssd.items()}
clips = synth_clips(B, T, seed=0).to(dev)
for tf32 in (False,):
    for _ in range(3):
        O.extract_features(sdd, ssdd, clips)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        f = O.extract_features(sdd, ssdd, clips)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('stock PyTorch-ROCm fp32 (NCHW, MIOpen): %.2f ms/step = %.1f clip-features/s' % (dt * 1e3, B / dt))
cl = clips.view(B * T, 3, 256, 128)

# train-mode forward + backward of the CNN only (batch-stat BN, autograd), same GPU
sdt = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone())
       for k, v in sdd.items()}
def train_step():
    xu, xc = O.grl_forward(sdt, clips, train=True)
    (xu.sum() + xc.sum()).backward()
for _ in range(2):
    train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    train_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('stock PyTorch-ROCm fp32 train fwd+bwd (CNN only): %.2f ms/step = %.1f clips/s' % (dt * 1e3, B / dt))

# the same on this build (CNN forward + HIP backward only)
cnn.train()
def ours():
    xu, xc = cnn(clips)
    (xu.sum() + xc.sum()).backward()
for _ in range(2):
    ours()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ours()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('this build        fp32 train fwd+bwd (CNN only): %.2f ms/step = %.1f clips/s' % (dt * 1e3, B / dt))
