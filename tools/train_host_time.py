#!/usr/bin/env python
"""Is the training step host-bound?  CPU time spent ISSUING forward / backward (no sync) next to the GPU time of the
step.   python tools/train_host_time.py [math]"""
import os, sys, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.reid.train import SEQTrainer
from grl_amd.reid.loss import OIMLoss, PairLoss
from grl_amd.synthetic import synth_clips, synth_state_dict
math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
cnn.load_state_dict(synth_state_dict(cnn, seed=0))
cnn, siam, siamv = cnn.to(dev).train(), siam.to(dev).train(), siamv.to(dev).train()
tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
clips = synth_clips(32, 4, seed=0).to(dev)
pids = (torch.arange(32, device=dev) // 2 * 7) % 625
TE.set_math(math)
acc = [0.0] * 4
N = 6
for it in range(3 + N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _, _, _ = tr._forward([clips], pids, 0, 0)
    t1 = time.perf_counter()
    opt.zero_grad(); loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if it >= 3:
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t0)): acc[i] += v * 1e3 / N
print('%s: host issue time forward %.1f ms, backward %.1f ms, optimizer %.1f ms; step (synchronised) %.1f ms' % ((math,) + tuple(acc)))
