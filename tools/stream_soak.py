#!/usr/bin/env python
"""Soak test of the multi-stream paths: many repetitions of (a) the two-stream eval extraction against the
single-stream result, fp32 and bf16-storage, several batch shapes, (b) the training step (TRL directions + weight
gradients on side streams) against its own first run.  Any mismatch is a race.   python tools/stream_soak.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from grl_amd import engine
from grl_amd.synthetic import synth_clips, synth_clips_structured
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')
cnn, siam, _, _ = bench.build_models(dev)
bad = 0
for mode in ('f32', 'bf16s'):
    for (b, t) in ((32, 4), (8, 4), (1, 4), (5, 3), (64, 8) if mode == 'bf16s' else (16, 2)):
        c = synth_clips(b, t, seed=b + t).to(dev)
        with engine.math_mode(mode):
            engine.TRL_STREAMS = False
            want = engine.extract_features(cnn, siam, c)
            engine.TRL_STREAMS = True
            n = sum(0 if torch.equal(engine.extract_features(cnn, siam, c), want) else 1 for _ in range(reps))
        print('eval %-5s B=%2d T=%d: %d/%d mismatches' % (mode, b, t, n, reps), flush=True)
        bad += n
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_parity as TP
clips = synth_clips_structured(8, 4, seed=21).to(dev)
rg = torch.Generator().manual_seed(9)
r1, r2 = torch.randn(8, 2048, generator=rg).to(dev), torch.randn(8, 4, 2048, generator=rg).to(dev)
first = None
n = 0
for it in range(max(10, reps // 5)):
    m = TP._fresh_cnn_conditioned(); m.train()
    xu, xc = m(clips)
    ((xu * r1).sum() + (xc * r2).sum()).backward()
    torch.cuda.synchronize()
    g = [p.grad.clone() for p in m.parameters() if p.grad is not None] + [xu.detach(), xc.detach()]
    if first is None:
        first = g
    elif not all(torch.equal(a, b) for a, b in zip(g, first)):
        n += 1
print('train step: %d/%d runs differ from the first' % (n, max(10, reps // 5) - 1))
bad += n
# (c) the whole SEQTrainer step -- heads on two streams, weight table on the side stream, fused BatchNorm reduce,
# gradient buffers adopted by autograd, fused SGD -- twice from the same start: every parameter, BatchNorm statistic and
# OIM table equal after `nst` steps
import contextlib, io
from grl_amd.reid import models
from grl_amd.reid.train import SEQTrainer
from grl_amd.reid.loss import OIMLoss, PairLoss
from grl_amd.synthetic import synth_state_dict
nst = max(8, reps // 4)
def run_trainer():
    with contextlib.redirect_stdout(io.StringIO()):
        c = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    sm = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    sv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    c.load_state_dict(synth_state_dict(c, seed=0, profile='conditioned'))
    sm.load_state_dict(synth_state_dict(sm, seed=0, prefix='siamese.'))
    sv.load_state_dict(synth_state_dict(sv, seed=0, prefix='siamese_video.'))
    mods = [m.to(dev).train() for m in (c, sm, sv)]
    crits = [OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev) for _ in range(2)]
    tr = SEQTrainer(mods[0], mods[1], mods[2], PairLoss().to(dev), crits[0], crits[1], None)
    opt = torch.optim.SGD(tr._all_params(), lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True, fused=True)
    for i in range(nst):
        cl = synth_clips_structured(8, 4, seed=100 + i).to(dev)
        pid = ((torch.arange(8) // 2) * 7 + i).to(dev) % 625
        loss = tr._forward([cl], pid, 0, 0)[0]
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
    return [v.detach().clone() for m in mods for v in m.state_dict().values()] + [cr.lut.clone() for cr in crits]
a, b = run_trainer(), run_trainer()
n = sum(0 if torch.equal(x, y) else 1 for x, y in zip(a, b))
print('trainer, %d steps twice: %d/%d state tensors differ' % (nst, n, len(a)))
bad += n
# (d) round 6: the compressed-input pipeline -- engine.DevicePrefetcher decodes JPEG batches two ahead on its own
# high-priority streams (pinned ring, H2D, unstuff / entropy / IDCT / colour kernels) while the previous batches' eval
# steps run on the main and TRL streams.  DIFFERENT batches in flight at once (a race between a decode and a consumer
# would mix them up): every batch's features must equal those of the same frames decoded serially and fed as a tensor.
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import decode_rate
from grl_amd.reid.data.jpeg import JpegBatch, decode_jpeg_batch
cnn.eval(); siam.eval()
nb = 6
frames = decode_rate.make_frames(nb * 8 * 4)
batches = [JpegBatch(frames[i * 32:(i + 1) * 32], (8, 4)) for i in range(nb)]
want = []
with engine.math_mode('f32'):
    for jb in batches:
        px = decode_jpeg_batch(jb, dev)
        torch.cuda.synchronize()
        want.append(engine.extract_features(cnn, siam, px).clone())
    torch.cuda.synchronize()
    n = tot = 0
    for _ in range(max(5, reps // 10)):
        for k, (d_, _, _) in enumerate(engine.DevicePrefetcher(((jb, None, None) for jb in batches), dev)):
            tot += 1
            if not torch.equal(engine.extract_features(cnn, siam, d_), want[k]):
                n += 1
    torch.cuda.synchronize()
print('jpeg-fed eval through the prefetcher: %d/%d batches differ from the serial decode' % (n, tot))
bad += n
print('SOAK', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
