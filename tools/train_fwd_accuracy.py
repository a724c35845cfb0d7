"""GPU box: per-stage accuracy of the HIP train-mode forward against the float64 oracle, next to
the fp32 oracle's (= what the reference computes).  Conditioned weights, structured clips."""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips_structured
from oracle import grl_oracle as O
B, T = 8, 4
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
sd0 = synth_state_dict(cnn, seed=0, profile='conditioned')
cnn.load_state_dict(sd0)
clips = synth_clips_structured(B, T, seed=3)
torch.set_num_threads(16)
def oracle(dt):
    sd = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items()}
    taps = {}
    with torch.no_grad():
        xu, xc = O.grl_forward(sd, clips.to(dt), train=True, taps=taps)
    taps['x_uncorr'], taps['x_corr'] = xu, xc
    return taps
t64, t32 = oracle(torch.float64), oracle(torch.float32)
cnn.cuda().train()
cnn._grl_taps = {}
with torch.no_grad():
    xu, xc = cnn(clips.cuda())
th = dict(cnn._grl_taps); th['x_uncorr'], th['x_corr'] = xu, xc
def rel2(a, b):
    a = a.double().cpu().reshape(-1); b = b.double().reshape(-1)
    return float((a - b).norm() / b.norm())
for k in ('stem', 'pool', 'layer1', 'layer2', 'layer3', 'layer4', 'x_glo', 'glo', 'corr_map', 'f_uncorr', 'f_corr', 'x_uncorr', 'x_corr'):
    if k in th and k in t64:
        a = th[k].contiguous()
        print('%-10s L2-rel vs f64: HIP %.2e   torch-fp32 %.2e' % (k, rel2(a.reshape(t64[k].shape) if a.numel() == t64[k].numel() else a, t64[k]), rel2(t32[k], t64[k])))
