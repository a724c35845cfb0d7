#!/usr/bin/env python
"""bf16-storage training ('bf16s') against the exact-fp32 step on the same inputs: outputs, parameter gradients
(relative L2 / cosine per tensor), then the step time at B x T = 32 x 4 and 64 x 8.
   python tools/bf16s_train_check.py [B T]"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from grl_amd import train_engine as TE
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict, synth_clips_structured, synth_clips

dev = torch.device('cuda:0')
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 4)


def fresh():
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'))
    return cnn.to(dev).train()


def run(math, clips, r1, r2):
    cnn = fresh()
    old = TE.set_math(math)
    try:
        xu, xc = cnn(clips)
        ((xu * r1).sum() + (xc * r2).sum()).backward()
    finally:
        TE.set_math(old)
    torch.cuda.synchronize()
    return xu.detach(), xc.detach(), {k: p.grad.detach().clone() for k, p in cnn.named_parameters() if p.grad is not None}


clips = synth_clips_structured(B, T, seed=3).to(dev)
g = torch.Generator().manual_seed(7)
r1, r2 = torch.randn(B, 2048, generator=g).to(dev), torch.randn(B, T, 2048, generator=g).to(dev)
a = run('f32', clips, r1, r2)
b = run('bf16s', clips, r1, r2)
rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
print('outputs: x_uncorr %.2e x_corr %.2e (max-norm relative)' % (rel(b[0], a[0]), rel(b[1], a[1])))
errs = {}
for k in a[2]:
    ga, gb = a[2][k].double().reshape(-1), b[2][k].double().reshape(-1)
    if float(ga.norm()) < 1e-12:
        continue
    errs[k] = (float((ga - gb).norm() / ga.norm()), float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-300)))
v = np.array(sorted(e[0] for e in errs.values()))
c = np.array(sorted(e[1] for e in errs.values()))
print('%d gradient tensors: relative L2 median %.2e p90 %.2e max %.2e; cosine min %.5f median %.5f' % (
    len(v), np.median(v), v[int(0.9 * len(v))], v[-1], c[0], np.median(c)))
for k, e in sorted(errs.items(), key=lambda kv: -kv[1][0])[:8]:
    print('   %-60s L2 %.2e cos %.5f' % (k, e[0], e[1]))
groups = {}
for k, e in errs.items():
    gk = k.split('.')[0] + '.' + k.split('.')[1] + ('.' + k.split('.')[2] if k.startswith('backbone.base') else '')
    groups.setdefault(gk, []).append(e[0])
for gk, v in groups.items():
    print('   group %-50s n %3d  median L2 %.2e' % (gk, len(v), float(np.median(v))))
assert all(torch.isfinite(x).all() for x in b[2].values())
if os.environ.get('GRL_CHECK_ONLY'):
    sys.exit(0)

for (bb, tt) in ((32, 4), (64, 8)):
    for math in ('f32', 'bf16s'):
        cnn = fresh()
        cl = synth_clips(bb, tt, seed=0).to(dev)
        q1, q2 = torch.randn(bb, 2048, device=dev), torch.randn(bb, tt, 2048, device=dev)
        old = TE.set_math(math)
        try:
            def step():
                xu, xc = cnn(cl)
                cnn.zero_grad(set_to_none=True)
                ((xu * q1).sum() + (xc * q2).sum()).backward()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            print('CNN forward + backward %d x %d %s: %.2f ms' % (bb, tt, math, (time.perf_counter() - t0) / 4 * 1e3))
        finally:
            TE.set_math(old)
        del cnn, cl
        torch.cuda.empty_cache()
