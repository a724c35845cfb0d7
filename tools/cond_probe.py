"""Build-container probe: how far is the reference's own fp32 train forward+backward from its
float64 run, for a synthetic weight profile?  (Chooses the 'conditioned' profile of
grl_amd/synthetic.py.)   python tools/cond_probe.py B T [profile]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
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
res = {}
for dt in (torch.float32, torch.float64):
    cnn.load_state_dict(sd, strict=True); cnn.zero_grad(set_to_none=True); cnn.to(dt).train()
    t0 = time.time()
    xu, xc = cnn(clips.to(dt))
    ((xu * r1.to(dt)).sum() + (xc * r2.to(dt)).sum()).backward()
    res[dt] = (xu.detach().double(), xc.detach().double(), {k: p.grad.detach().double() for k, p in cnn.named_parameters() if p.grad is not None})
    print(dt, 'took %.1fs' % (time.time() - t0), flush=True)
a, b = res[torch.float32], res[torch.float64]
rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-300))
print('x_uncorr %.2e  x_corr %.2e' % (rel(a[0], b[0]), rel(a[1], b[1])))
rel2 = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-300))
e2 = {k: rel2(a[2][k], b[2][k]) for k in b[2] if float(b[2][k].abs().max()) > 1e-9}
v2 = np.array(sorted(e2.values()))
print('grads L2-rel: n %d median %.2e p90 %.2e max %.2e' % (len(v2), np.median(v2), v2[int(.9 * len(v2))], v2[-1]))
for k in sorted(e2, key=e2.get)[-5:]:
    print('  L2 %-60s %.2e' % (k, e2[k]))
errs = {k: rel(a[2][k], b[2][k]) for k in b[2]}
v = np.array(sorted(errs.values()))
print('grads: n %d median %.2e p90 %.2e max %.2e' % (len(v), np.median(v), v[int(.9 * len(v))], v[-1]))
for k in sorted(errs, key=errs.get)[-8:]:
    print('  %-60s %.2e  |g|max %.2e' % (k, errs[k], float(b[2][k].abs().max())))

if os.environ.get('ALL'):
    for k in errs:
        print('%-70s %.2e' % (k, errs[k]))
