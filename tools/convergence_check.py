#!/usr/bin/env python
"""Does the HIP training path LEARN, and do the 'mixed' and 'bf16s' (bf16-storage) datapaths train like exact fp32?

A small synthetic re-id problem (no dataset is available here): N identities, each a fixed low-frequency colour
layout; a clip = its identity's layout + a per-clip and per-frame deviation + pixel noise (256 x 128, T frames).
The reference's training step (SEQTrainer._forward: the 5-term loss, SGD with the reference's hyper-parameters) runs
from random initialisation for a few hundred iterations on P x K = 16 x 2 batches; then FRESH clips of the same
identities are ranked (cosine distance on the 6144-d evaluator features) and Rank-1 / mAP reported.

    python tools/convergence_check.py [iters] > profiles/rNN_convergence.md
"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from grl_amd import engine, train_engine as TE
from grl_amd.reid import models
from grl_amd.reid.train import SEQTrainer
from grl_amd.reid.loss import OIMLoss, PairLoss
from grl_amd.reid.evaluator.eva_functions import evaluate
from grl_amd.synthetic import IMAGENET_MEAN, IMAGENET_STD

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N_ID, T, P = 64, 4, 16
dev = torch.device('cuda:0')
g0 = np.random.Generator(np.random.PCG64(1234))
BASE = torch.from_numpy(g0.uniform(0.0, 1.0, (N_ID, 3, 8, 4)).astype(np.float32))


def clips_of(ids, seed):
    g = np.random.Generator(np.random.PCG64([seed, 5]))
    n = len(ids)
    dev_c = torch.from_numpy(g.uniform(-0.10, 0.10, (n, 1, 3, 8, 4)).astype(np.float32))
    dev_f = torch.from_numpy(g.uniform(-0.05, 0.05, (n, T, 3, 8, 4)).astype(np.float32))
    low = F.interpolate((BASE[ids].unsqueeze(1) + dev_c + dev_f).view(n * T, 3, 8, 4), size=(256, 128), mode='bilinear',
                        align_corners=False)
    noise = torch.from_numpy(g.uniform(-0.1, 0.1, (n * T, 3, 256, 128)).astype(np.float32))
    x = (low + noise).clamp_(0, 1).view(n, T, 3, 256, 128)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 1, 3, 1, 1); std = torch.tensor(IMAGENET_STD).view(1, 1, 3, 1, 1)
    return ((x - mean) / std).to(dev)


def run(math):
    torch.manual_seed(0); np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn, siam, siamv = cnn.to(dev), siam.to(dev), siamv.to(dev)
    tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev),
                    OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), None)
    base_ids = set(map(id, cnn.backbone.parameters()))
    groups = [{'params': list(cnn.backbone.parameters()), 'lr': 1e-3},
              {'params': [p for p in cnn.parameters() if id(p) not in base_ids] + list(siam.parameters()) + list(siamv.parameters()), 'lr': 2e-3}]
    opt = torch.optim.SGD(groups, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)   # mars_train.py:93-103
    TE.set_math(math)
    sched = np.random.Generator(np.random.PCG64(77))
    curve, t0 = [], time.time()
    for it in range(ITERS):
        cnn.train(); siam.train(); siamv.train()
        ids = np.repeat(sched.choice(N_ID, P, replace=False), 2)              # pairs (2i, 2i+1) share the identity
        loss, p_u, p_v, p_f = tr._forward([clips_of(ids, 1000 + it)], torch.from_numpy(ids).to(dev), it, 0)
        opt.zero_grad(); loss.backward(); opt.step()
        if it % 25 == 0 or it == ITERS - 1:
            curve.append((it, float(loss.detach()), float(p_v), float(p_f)))
    torch.cuda.synchronize()
    secs = time.time() - t0
    TE.set_math('f32')
    cnn.eval(); siam.eval()
    q_ids, g_ids = np.arange(N_ID), np.repeat(np.arange(N_ID), 3)
    with torch.no_grad():
        qf = torch.cat([engine.extract_features(cnn, siam, clips_of(q_ids[i:i + 32], 9000 + i)) for i in range(0, N_ID, 32)])
        gf = torch.cat([engine.extract_features(cnn, siam, clips_of(g_ids[i:i + 32], 9500 + i)) for i in range(0, len(g_ids), 32)])
    qn, gn = F.normalize(qf, dim=1), F.normalize(gf, dim=1)
    dist = (-(qn @ gn.t())).cpu().numpy()
    with contextlib.redirect_stdout(io.StringIO()):
        cmc, mAP = evaluate(dist, q_ids, g_ids, np.zeros(N_ID, int), np.ones(len(g_ids), int), max_rank=10)
    return curve, float(cmc[0]), float(mAP), secs


print('# r03: does the HIP training path learn, and do `mixed` / `bf16s` train like exact fp32?\n')
print('`tools/convergence_check.py %d`: %d synthetic identities (fixed colour layouts + per-clip / per-frame deviation + noise), '
      'random initialisation, the reference\'s 5-term loss and SGD settings, P x K = %d x 2 clips of %d frames per step; '
      'then fresh clips of the same identities are ranked on the 6144-d evaluator features (chance Rank-1 = %.1f %%).\n' % (ITERS, N_ID, P, T, 100.0 / N_ID))
res = {}
for m in ('f32', 'mixed', 'bf16s'):
    res[m] = run(m)
print('| iteration | ' + ' | '.join('%s: loss / clip-id acc / frame-id acc' % m for m in res) + ' |')
print('|---|' + '---|' * len(res))
for k in range(len(res['f32'][0])):
    print('| %d | ' % res['f32'][0][k][0] + ' | '.join('%.3f / %.2f / %.2f' % res[m][0][k][1:] for m in res) + ' |')
print()
for m in res:
    print('* `%s`: Rank-1 %.1f %%, mAP %.1f %% after %d iterations (%.0f s wall, data generation included)' % (
        m, 100 * res[m][1], 100 * res[m][2], ITERS, res[m][3]))
