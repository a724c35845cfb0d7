#!/usr/bin/env python
"""Per-shape timing of every GEMM / weight-gradient launch of one training step (HIP events around each launch).
   python tools/train_gemm_report.py [math] > report.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from grl_amd import engine, train_engine as TE
from grl_amd.synthetic import synth_clips

math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
dev = torch.device('cuda:0')
recs = []
og, ow = engine.gemm, TE.wgrad
phase = ['fwd']


def tg(a, w, y, M, N, K, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = og(a, w, y, M, N, K, *args, **kw); e1.record()
    recs.append((phase[0] + ' gemm', (M, N, K, kw.get('conv'), 'stats' if kw.get('stats') else ''), 2.0 * M * N * K, e0, e1))
    return r


def tw(dz, x, dw, M, N, K, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ow(dz, x, dw, M, N, K, *args, **kw); e1.record()
    recs.append(('wgrad', (M, N, K, kw.get('conv')), 2.0 * M * N * K * (kw['conv'][5] * kw['conv'][6] if kw.get('conv') and False else 1), e0, e1))
    return r


from grl_amd.reid.train.trainer import SEQTrainer
from grl_amd.reid.loss import PairLoss, OIMLoss
from grl_amd.reid import models
from grl_amd.synthetic import synth_state_dict
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
cnn.load_state_dict(synth_state_dict(cnn, seed=0)); cnn.to(dev); siam.to(dev); siamv.to(dev)
crit_c = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev); crit_u = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev)
trainer = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), crit_c, crit_u, None)
cnn.train(); siam.train(); siamv.train()
TE.set_math(math)
clips = synth_clips(32, 4, seed=0).to(dev)
pids = (torch.arange(32, device=dev) // 2 * 7) % 625
obk = TE.Tape.backward


def bk(self):
    phase[0] = 'bwd'
    try:
        return obk(self)
    finally:
        phase[0] = 'fwd'


for it in range(3):
    if it == 2:
        engine.gemm, TE.wgrad, TE.Tape.backward = tg, tw, bk
    loss, _, _, _ = trainer._forward([clips], pids, 0, 0)
    for p in trainer._all_params():
        p.grad = None
    loss.backward()
torch.cuda.synchronize()
agg = {}
for kind, shape, fl, a, b in recs:
    e = agg.setdefault((kind, str(shape)), [0, 0.0, fl])
    e[0] += 1; e[1] += a.elapsed_time(b)
tot = {}
for (kind, shape), (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-9s %-62s calls %3d  total %7.3f ms  %6.1f TF/s' % (kind, shape, cnt, ms, fl * cnt / (ms * 1e-3) / 1e12))
    tot[kind] = tot.get(kind, 0) + ms
print(tot)
