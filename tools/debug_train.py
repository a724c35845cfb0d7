import sys, io, contextlib
sys.path.insert(0, '.')
import numpy as np, torch
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips
from oracle import grl_oracle as O
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', pretrained=False)
sd = synth_state_dict(cnn, seed=0)
cnn.load_state_dict(sd)
clips = synth_clips(2, 4, seed=0)
ot = {}
sdc = {k: v.clone() for k, v in sd.items()}
with torch.no_grad():
    xu_o, xc_o = O.grl_forward(sdc, clips, train=True, taps=ot)
cnn.cuda().train()
cnn._grl_taps = {}
xu, xc = cnn(clips.cuda())
gt = cnn._grl_taps
def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max()).item()
for k in ['stem', 'pool', 'layer1', 'layer2', 'layer3', 'layer4', 'x_glo', 'glo', 'corr_map', 'f_uncorr', 'f_corr']:
    print(k, rel(gt[k].reshape(ot[k].shape), ot[k]))
for d in ('fwd', 'bwd'):
    for i in range(4):
        print(d, i, 'catte', rel(gt[d + '_catte'][i], ot[d + '_catte'][i]), 'memo', )
print('xu', rel(xu, xu_o), 'xc', rel(xc, xc_o))
st = cnn.state_dict()
for k in ['backbone.base.1.running_var', 'backbone.base.7.2.bn3.running_var', 'backbone.corr_atte.6.running_var', 'backbone.corr_atte.6.running_mean',
          'temporal_learning_block.uncorr_memo_forward.bn1.running_var', 'uncorr_bn.running_var', 'uncorr_bn.running_mean']:
    print(k, rel(st[k], sdc[k]))
