#!/usr/bin/env python
"""cProfile of the host side of one training step (which Python functions the launch overhead sits in).
   python tools/train_host_profile.py [math]"""
import os, sys, contextlib, io, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.reid.train import SEQTrainer
from grl_amd.reid.loss import OIMLoss, PairLoss
from grl_amd.synthetic import synth_clips, synth_state_dict
math = sys.argv[1] if len(sys.argv) > 1 else 'bf16s'
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
cnn.load_state_dict(synth_state_dict(cnn, seed=0))
cnn, siam, siamv = cnn.to(dev).train(), siam.to(dev).train(), siamv.to(dev).train()
tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
clips = synth_clips(32, 4, seed=0).to(dev)
pids = (torch.arange(32, device=dev) // 2 * 7) % 625
TE.set_math(math)


def step():
    loss, _, _, _ = tr._forward([clips], pids, 0, 0)
    opt.zero_grad(); loss.backward(); opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
