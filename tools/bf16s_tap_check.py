#!/usr/bin/env python
"""Where does the bf16-storage training forward leave the fp32 one?  Relative L2 error of every tap."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips_structured
dev = torch.device('cuda:0')
B, T = 8, 4
out = {}
for math in ('f32', 'bf16s'):
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile=sys.argv[1] if len(sys.argv) > 1 else 'conditioned'))
    cnn = cnn.to(dev).train()
    cnn._grl_taps = {}
    old = TE.set_math(math)
    try:
        with torch.no_grad():
            xu, xc = cnn(synth_clips_structured(B, T, seed=3).to(dev))
    finally:
        TE.set_math(old)
    taps = dict(cnn._grl_taps)
    taps['x_uncorr'], taps['x_corr'] = xu, xc
    out[math] = {k: (v.float() if torch.is_tensor(v) else torch.stack([t.float() for t in v])) for k, v in taps.items()}
for k in out['f32']:
    a, b = out['f32'][k].double(), out['bf16s'][k].double()
    print('%-14s rel L2 %.2e   max-norm rel %.2e   |ref| rms %.3g' % (k, float((a - b).norm() / a.norm()),
          float((a - b).abs().max() / a.abs().max()), float(a.pow(2).mean().sqrt())))
