#!/usr/bin/env python
"""GPU time of the fwd -> bwd turnaround of one training step (NOT under a profiler): everything between the last
launch of the CNN forward and the first launch of the CNN backward -- the two Siamese heads, the five loss terms, their
backward and autograd's gradient sums.    python tools/turnaround_time.py [math] [BxT]"""
import os, sys, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.reid.train import SEQTrainer
from grl_amd.reid.loss import OIMLoss, PairLoss
from grl_amd.synthetic import synth_clips, synth_state_dict
math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
b, t = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '32x4').split('x'))
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
cnn.load_state_dict(synth_state_dict(cnn, seed=0))
cnn, siam, siamv = cnn.to(dev).train(), siam.to(dev).train(), siamv.to(dev).train()
tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
try:
    opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
except Exception:
    opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
clips = synth_clips(b, t, seed=0).to(dev)
pids = (torch.arange(b, device=dev) // 2 * 7) % 625
TE.set_math(math)
ev = {k: torch.cuda.Event(enable_timing=True) for k in ('s', 'a', 'b', 'e')}
host = {}
orig_fwd = cnn.forward
def fwd(x):
    out = orig_fwd(x)
    ev['a'].record(); host['a'] = time.perf_counter()
    return out
cnn.forward = fwd
orig_bwd = TE._GrlTrainFn.backward
def bwd(ctx, *g):
    ev['b'].record(); host['b'] = time.perf_counter()
    return orig_bwd(ctx, *g)
TE._GrlTrainFn.backward = staticmethod(bwd)
N, acc, hacc = 8, [0.0] * 4, [0.0] * 3
for it in range(3 + N):
    ev['s'].record(); host['s'] = time.perf_counter()
    loss, _, _, _ = tr._forward([clips], pids, 0, 0)
    opt.zero_grad(); loss.backward(); opt.step()
    ev['e'].record(); host['e'] = time.perf_counter()
    torch.cuda.synchronize()
    if it >= 3:
        for i, (p, q) in enumerate((('s', 'a'), ('a', 'b'), ('b', 'e'), ('s', 'e'))):
            acc[i] += ev[p].elapsed_time(ev[q]) / N
        for i, (p, q) in enumerate((('s', 'a'), ('a', 'b'), ('b', 'e'))):
            hacc[i] += (host[q] - host[p]) * 1e3 / N
print('%s %dx%d GPU: cnn forward %.2f ms | heads + losses fwd/bwd (turnaround) %.2f ms | cnn backward + SGD %.2f ms | step %.2f ms'
      % ((math, b, t) + tuple(acc)))
print('        host issue: %.2f | %.2f | %.2f ms (each step synchronised: the host never runs ahead of the step)' % tuple(hacc))
