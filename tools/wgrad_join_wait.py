#!/usr/bin/env python
"""How long does the launch stream WAIT for the weight-gradient stream at the end of the backward?  (If ~0 the side
stream is not on the critical path and shortening it -- e.g. reducing the slabs in-kernel -- would not shorten the
step.)   python tools/wgrad_join_wait.py [math]"""
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips
math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
cnn.load_state_dict(synth_state_dict(cnn, seed=0)); cnn = cnn.to(dev).train()
cl = synth_clips(32, 4, seed=0).to(dev)
TE.set_math(math)
waits = []
orig = TE.Tape.wgrad_join
def timed(self):
    if self.wheld:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                 # the launch stream's own work is done here
        orig(self)
        e1.record()                 # ... and here the side stream's is too
        waits.append((e0, e1))
    else:
        orig(self)
TE.Tape.wgrad_join = timed
for it in range(8):
    xu, xc = cnn(cl); cnn.zero_grad(set_to_none=True); (xu.sum() + xc.sum()).backward()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in waits]
per = len(ms) // 8
print('%s: %d joins per step; launch stream waited (ms, last 3 steps):' % (math, per), [round(v, 3) for v in ms[-3 * per:]])
