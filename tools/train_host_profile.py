#!/usr/bin/env python
"""cProfile of the host side of the CNN's training forward AND backward (the backward closures normally run on
autograd's worker thread, out of cProfile's sight: here the tape is replayed from the main thread).
   python tools/train_host_profile.py [math] [B] [T]"""
import os, sys, contextlib, io, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import train_engine as TE, engine
from grl_amd.reid import models
from grl_amd.synthetic import synth_clips, synth_state_dict
math = sys.argv[1] if len(sys.argv) > 1 else 'bf16s'
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 4)
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
cnn.load_state_dict(synth_state_dict(cnn, seed=0))
cnn = cnn.to(dev).train()
clips = synth_clips(B, T, seed=0).to(dev)
r1, r2 = torch.randn(B, 2048, device=dev), torch.randn(B, T, 2048, device=dev)
TE.set_math(math)
params = tuple(cnn.parameters())


def fwd_bwd():
    """what _GrlTrainFn.forward / backward do, on this thread"""
    tp = TE.Tape(dev)
    tp.b16 = math == 'bf16s'
    tp.reserve_param_grads(params, cuts=TE._grl_cuts(cnn))
    x = clips.view(B * T, 3, 256, 128)
    x4 = TE.trunk_train(tp, cnn, x)
    xu, xc, _ = TE.gce_train(tp, cnn, x4, B, T)
    fu, fc = TE.trl_train(tp, cnn, xu, xc, B, T)
    fc2d = fc.view(B * T, 2048)
    xcn = TE.bn1d_l2norm(tp, fc2d, B * T, 2048, cnn.corr_bn)
    xun = TE.bn1d_l2norm(tp, fu, B, 2048, cnn.uncorr_bn)
    tp.g[id(xun)] = r1
    tp.g[id(xcn)] = r2.view(B * T, 2048)
    tp.backward()


for _ in range(3):
    fwd_bwd()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    fwd_bwd()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(32)
