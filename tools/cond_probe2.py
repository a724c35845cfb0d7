"""Where does the reference's fp32 backward leave its float64 backward?  Hooks the gradient at the
input of every leaf module.   python tools/cond_probe2.py B T profile"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as MG
from grl_amd.synthetic import synth_state_dict, synth_clips, synth_clips_structured
B, T = int(sys.argv[1]), int(sys.argv[2])
profile = sys.argv[3] if len(sys.argv) > 3 else 'conditioned'
torch.set_num_threads(8)
ref_models = MG.import_reference()[0]
cnn = ref_models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625)
sd = synth_state_dict(cnn, seed=0, profile=profile)
clips = (synth_clips_structured if os.environ.get('STRUCT', '1') == '1' else synth_clips)(B, T, seed=3)
g = np.random.Generator(np.random.PCG64(7))
r1 = torch.from_numpy(g.standard_normal((B, 2048)).astype(np.float32))
r2 = torch.from_numpy(g.standard_normal((B, T, 2048)).astype(np.float32))
rec = {}
def run(dt):
    out = {}
    cnn.load_state_dict(sd, strict=True); cnn.zero_grad(set_to_none=True); cnn.to(dt).train()
    hs = []
    cnt = {}
    for name, m in cnn.named_modules():
        if len(list(m.children())) == 0:
            def fh(mod, inp, o, name=name):
                i = cnt.get(name, 0); cnt[name] = i + 1
                if torch.is_tensor(o) and o.requires_grad:
                    out[('act', name, i)] = o.detach().double()
                    o.register_hook(lambda gr, name=name, i=i: out.__setitem__(('grad', name, i), gr.detach().double()))
            hs.append(m.register_forward_hook(fh))
    xu, xc = cnn(clips.to(dt))
    ((xu * r1.to(dt)).sum() + (xc * r2.to(dt)).sum()).backward()
    for h in hs: h.remove()
    return out
a = run(torch.float32); b = run(torch.float64)
rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-300))
rel2 = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-300))
for k in b:
    if k[0] == 'grad':
        ka = ('act',) + k[1:]
        print('%-62s #%d act max %.1e l2 %.1e | grad max %.1e l2 %.1e' % (k[1], k[2], rel(a[ka], b[ka]), rel2(a[ka], b[ka]), rel(a[k], b[k]), rel2(a[k], b[k])))
