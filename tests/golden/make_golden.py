#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE itself (imported from
/root/reference, CPU, fp32) on the build's deterministic synthetic weights and
clips.  Runs only in the build container (the reference does not exist on the
GPU box); its outputs -- plain data, < 1 MB -- are committed next to this
script and consumed by tests/test_oracle_golden.py (oracle pin) and the
``-m gpu`` parity tests.

The reference needs three import-time stubs here (torchvision, tensorboardX,
cv2 are not installed) and ``model_zoo.load_url`` patched (no network); none of
them touch the math being pinned.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    tv = _stub('torchvision')
    tv.models = _stub('torchvision.models', resnet18=None, resnet34=None, resnet50=None,
                      resnet101=None, resnet152=None)
    tv.utils = _stub('torchvision.utils', save_image=lambda *a, **k: None)
    _stub('tensorboardX', SummaryWriter=lambda *a, **k: None)
    _stub('cv2')
    sys.path.insert(0, REF)
    import reid.models.resnets1 as r1            # noqa: the reference's package
    full = r1.ResNet(r1.Bottleneck, [3, 4, 6, 3]).state_dict()
    r1.model_zoo.load_url = lambda *a, **k: full
    from reid import models as ref_models
    from reid.evaluator import attevaluator, eva_functions, rerank
    from reid.loss import pairloss, triplet
    return ref_models, attevaluator, eva_functions, pairloss, triplet, rerank


def sample(t, n=256):
    """Strided sample + checksums of a big tensor."""
    f = t.detach().reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return dict(sum=f.sum().item(), abssum=f.abs().sum().item(),
                idx=idx.numpy(), val=f[idx].float().numpy(), shape=np.array(t.shape))


def pack(prefix, d, out):
    for k, v in d.items():
        out['%s.%s' % (prefix, k)] = np.asarray(v)


TRAIN_GRAD_KEYS = ('backbone.base.0.weight', 'backbone.base.1.weight', 'backbone.base.1.bias',
              'backbone.base.4.0.conv2.weight', 'backbone.base.5.0.downsample.0.weight',
              'backbone.base.7.2.conv3.weight', 'backbone.base.7.2.bn3.weight',
              'backbone.glo_fc.0.weight', 'backbone.glo_fc.1.bias',
              'backbone.corr_atte.0.weight', 'backbone.corr_atte.5.weight', 'backbone.corr_atte.6.weight',
              'temporal_learning_block.forward_f1.0.weight', 'temporal_learning_block.forward_f1.0.bias',
              'temporal_learning_block.backward_f2.0.weight',
              'temporal_learning_block.channel_atte_foreward_corr.0.weight',
              'temporal_learning_block.channel_atte_backward_corr.2.weight',
              'temporal_learning_block.uncorr_memo_forward.conv1.weight',
              'temporal_learning_block.uncorr_memo_backward.bn3.weight',
              'corr_bn.weight', 'uncorr_bn.bias')
TRAIN_STAT_KEYS = ('backbone.base.1.running_mean', 'backbone.base.1.running_var',
              'backbone.base.7.2.bn3.running_var', 'backbone.glo_fc.1.running_mean',
              'backbone.corr_atte.6.running_var',
              'temporal_learning_block.uncorr_memo_forward.bn1.running_mean',
              'temporal_learning_block.uncorr_memo_forward.bn1.running_var',
              'temporal_learning_block.uncorr_memo_forward.bn1.num_batches_tracked',
              'temporal_learning_block.uncorr_memo_backward.bn3.running_var',
              'corr_bn.running_mean', 'uncorr_bn.running_var', 'corr_bn.num_batches_tracked')


def train_golden(cnn, sd, clips, B, T, path):
    cnn.load_state_dict(sd, strict=True)
    cnn.zero_grad(set_to_none=True)
    cnn.train()
    g = np.random.Generator(np.random.PCG64(7))
    r1 = torch.from_numpy(g.standard_normal((B, 2048)).astype(np.float32))
    r2 = torch.from_numpy(g.standard_normal((B, T, 2048)).astype(np.float32))
    x_in = clips.clone().requires_grad_(True)
    xu, xc = cnn(x_in)
    loss = (xu * r1).sum() + (xc * r2).sum()
    loss.backward()
    out = {'x_uncorr': xu.detach().numpy(), 'x_corr': xc.detach().numpy(),
           'loss': np.array(loss.item())}
    named = dict(cnn.named_parameters())
    for k in TRAIN_GRAD_KEYS:
        pack('grad.' + k, sample(named[k].grad), out)
    pack('grad.input', sample(x_in.grad), out)
    st = cnn.state_dict()
    for k in TRAIN_STAT_KEYS:
        out['stat.' + k] = st[k].numpy().copy()      # (a view would follow the float64 pass below)
    # The same graph in float64 (the reference model cast with .double()): the yardstick for the
    # ill-conditioned B = 2 case -- the tests require the HIP gradients to be as close to these as
    # the reference's own float32 run is.
    cnn.load_state_dict(sd, strict=True)
    cnn.zero_grad(set_to_none=True)
    cnn.double().train()
    x64 = clips.double().clone().requires_grad_(True)
    xu64, xc64 = cnn(x64)
    ((xu64 * r1.double()).sum() + (xc64 * r2.double()).sum()).backward()
    out['f64.x_uncorr'], out['f64.x_corr'] = xu64.detach().numpy(), xc64.detach().numpy()
    named = dict(cnn.named_parameters())
    for k in TRAIN_GRAD_KEYS:
        d = sample(named[k].grad)
        out['f64.grad.' + k + '.val'] = named[k].grad.detach().reshape(-1)[torch.from_numpy(d['idx'])].numpy()
        out['f64.grad.' + k + '.abssum'] = np.asarray(d['abssum'])
    cnn.float()
    cnn.load_state_dict(sd, strict=True)
    cnn.zero_grad(set_to_none=True)
    np.savez_compressed(path, **out)
    print('train golden %s: loss' % os.path.basename(path), loss.item())


def oim_golden(ref_models, path):
    """(F) OIM pinned to the reference (reid/loss/oim.py:8-53).  The reference's OIM is a legacy
    non-static autograd.Function, which torch >= 1.5 refuses to *apply*; its forward/backward
    BODIES (oim.py:14-27) still execute unmodified as plain functions on a stub ``self`` that carries
    what a legacy Function instance carried (lut, momentum, needs_input_grad, save_for_backward ->
    saved_tensors).  A static bridge Function hands them to autograd, ``reid.loss.oim.oim`` is
    pointed at the bridge and ``OIMLoss.forward`` (oim.py:46-53) runs as shipped -- no math touched,
    the same standard as the import stubs above."""
    import importlib
    ref_oim = importlib.import_module('reid.loss.oim')   # (`reid.loss.oim` the attribute is the function)
    order = []

    class Bridge(torch.autograd.Function):
        @staticmethod
        def forward(ctx, inputs, targets, lut, momentum, tag):
            stub = types.SimpleNamespace(lut=lut, momentum=momentum, needs_input_grad=(True, False))
            stub.save_for_backward = lambda *t: setattr(stub, 'saved_tensors', t)
            ctx.stub, ctx.tag = stub, tag
            return ref_oim.OIM.forward(stub, inputs, targets)          # oim.py:14-17

        @staticmethod
        def backward(ctx, grad_outputs):
            order.append(ctx.tag)
            gi, _ = ref_oim.OIM.backward(ctx.stub, grad_outputs)       # oim.py:19-27
            return gi, None, None, None, None

    tag = ['?']
    ref_oim.oim = lambda inputs, targets, lut, momentum=0.5: Bridge.apply(inputs, targets, lut, momentum, tag[0])

    out = {}
    g = np.random.Generator(np.random.PCG64(17))
    D, NC = 256, 12          # (a)-(c); the training-step case (d) is 2048 wide (the Siamese input)

    def unit(a):
        return (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)

    def one(name, x, y, lut0, upstream):
        crit = ref_oim.OIMLoss(D, NC, scalar=30, momentum=0.5)
        crit.lut.copy_(torch.from_numpy(lut0))
        xt = torch.from_numpy(x).clone().requires_grad_(True)
        tag[0] = name
        loss, logits = crit(xt, torch.from_numpy(y))
        (loss * upstream).backward()
        out[name + '.x'], out[name + '.y'], out[name + '.lut0'] = x, y, lut0
        out[name + '.upstream'] = np.array(upstream, np.float32)
        out[name + '.loss'] = np.array(loss.item(), np.float32)
        out[name + '.logits'] = logits.detach().numpy()
        out[name + '.grad_x'] = xt.grad.numpy()
        out[name + '.lut1'] = crit.lut.numpy().copy()
        print('oim golden %-6s loss %.6f' % (name, loss.item()))

    # (a) zero LUT (the state at step 0: oim.py:43), distinct labels
    one('zero', unit(g.standard_normal((8, D))), np.array([0, 1, 2, 3, 4, 5, 6, 7], np.int64),
        np.zeros((NC, D), np.float32), 1.0)
    # (b) unit-norm LUT, distinct labels, non-unit upstream gradient
    one('unit', unit(g.standard_normal((8, D))), np.array([3, 1, 4, 11, 5, 9, 2, 6], np.int64),
        unit(g.standard_normal((NC, D))), 0.7)
    # (c) duplicate labels: the sequential per-sample update is order dependent (oim.py:24-26),
    #     a partly zero LUT
    lut = unit(g.standard_normal((NC, D))); lut[[2, 7]] = 0
    one('dup', unit(g.standard_normal((24, D))),
        np.array([5, 5, 5, 5, 9, 9, 9, 9, 5, 5, 2, 2, 7, 7, 7, 7, 9, 9, 0, 0, 5, 2, 7, 0], np.int64), lut, 1.0)

    # (d) the two same-LUT calls of one training step (trainer.py:117-127,137-139): frame-level OIM
    #     on x_corr [B*T], then clip-level OIM on the reference Siamese's pooled output -- ONE
    #     criterion, ONE LUT; the update order is the autograd engine's (recorded in `order`).
    sys.path.insert(0, REPO)
    from grl_amd.synthetic import synth_state_dict
    B, T, D = 4, 4, 2048
    siam = ref_models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'), strict=True)
    siam.train()
    xc = unit(g.standard_normal((B, T, D)))
    ids = np.array([3, 3, 9, 9], np.int64)               # (anchor, positive) pairs, sampler.py:104-123
    lut0 = unit(g.standard_normal((NC, D))); lut0[9] = 0
    crit = ref_oim.OIMLoss(D, NC, scalar=30, momentum=0.5)
    crit.lut.copy_(torch.from_numpy(lut0))
    xt = torch.from_numpy(xc).clone().requires_grad_(True)
    targets = torch.from_numpy(ids)
    frame = xt.view(B * T, -1)
    targetX = targets.unsqueeze(1).expand(B, T).contiguous().view(B * T, -1).squeeze(1)
    del order[:]
    tag[0] = 'frame'
    l_frame, _ = crit(frame, targetX)                   # trainer.py:126
    tv = targets.view(B // 2, -1)
    target = torch.cat((tv[:, 0], tv[:, 1]))            # trainer.py:130-135
    _, pooled = siam(xt)                                # trainer.py:137
    tag[0] = 'vid'
    l_vid, _ = crit(pooled, target)                     # trainer.py:138
    (l_frame + l_vid).backward()
    out['step.x_corr'], out['step.ids'], out['step.lut0'] = xc, ids, lut0
    out['step.loss_frame'] = np.array(l_frame.item(), np.float32)
    out['step.loss_vid'] = np.array(l_vid.item(), np.float32)
    out['step.pooled'] = pooled.detach().numpy()
    out['step.grad_x_corr'] = xt.grad.numpy()
    out['step.lut1'] = crit.lut.numpy().copy()
    out['step.backward_order'] = np.array(order)       # ['vid', 'frame']: later node first
    print('oim golden step: losses %.6f %.6f, backward order %s' % (l_frame.item(), l_vid.item(), order))
    np.savez_compressed(path, **out)


def install_oim_bridge():
    """Points ``reid.loss.oim.oim`` at a static autograd.Function that runs the reference's OIM.forward /
    OIM.backward BODIES (oim.py:14-27) on a stub ``self`` -- see oim_golden above.  Returns (module, order)."""
    import importlib
    ref_oim = importlib.import_module('reid.loss.oim')
    order = []

    class Bridge(torch.autograd.Function):
        @staticmethod
        def forward(ctx, inputs, targets, lut, momentum):
            stub = types.SimpleNamespace(lut=lut, momentum=momentum, needs_input_grad=(True, False))
            stub.save_for_backward = lambda *t: setattr(stub, 'saved_tensors', t)
            ctx.stub = stub
            return ref_oim.OIM.forward(stub, inputs, targets)

        @staticmethod
        def backward(ctx, grad_outputs):
            order.append(int(ctx.stub.saved_tensors[0].size(0)))
            gi, _ = ref_oim.OIM.backward(ctx.stub, grad_outputs)
            return gi, None, None, None

    ref_oim.oim = lambda inputs, targets, lut, momentum=0.5: Bridge.apply(inputs, targets, lut, momentum)
    return ref_oim, order


def grad_record(prefix, named_grads, out, n=256):
    keys = []
    for k, gten in named_grads:
        if gten is None:
            continue
        ga = gten.detach().reshape(-1).double()
        if float(ga.abs().max()) < 1e-12:
            continue
        idx = torch.linspace(0, ga.numel() - 1, min(n, ga.numel())).long()
        out['%s.%s.val' % (prefix, k)] = ga[idx].float().numpy()
        out['%s.%s.norm' % (prefix, k)] = np.array(ga.norm().item())
        out['%s.%s.proj' % (prefix, k)] = np.array([float((ga * sign_pattern(ga.numel(), sd)).sum()) for sd in range(4)])
        keys.append(k)
    out[prefix + '.keys'] = np.array(keys)


def trainer_step_golden(ref_models, path, B=8, T=4):
    """(I) ONE training step of the reference's own SEQTrainer (reid/train/trainer.py:107-170 `_forward` as
    shipped + `loss.backward()`, :53-55) on CPU: conditioned CNN weights, structured clips, B x T = 8 x 4 (four
    (anchor, positive) pairs), unit-norm LUTs.  OIMLoss goes through the bridge above (the reference's
    forward / backward bodies; its legacy Function cannot be applied on this torch); TripletLoss, PairLoss,
    the Siamese heads and the 5-term composition run unmodified.  Stored: the loss, the three precisions, the
    model outputs handed to the heads, every parameter gradient of the two Siamese heads (samples, norm,
    projections) and of the CNN (same packing), the LUT rows of the batch identities after the step and the
    heads' BatchNorm running statistics."""
    sys.path.insert(0, REPO)
    from grl_amd.synthetic import synth_state_dict, synth_clips_structured
    from reid.train import trainer as ref_trainer
    from reid.loss import pairloss
    ref_oim, order = install_oim_bridge()
    cnn = ref_models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'), strict=True)
    siam = ref_models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'), strict=True)
    siamv = ref_models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'), strict=True)
    g = np.random.Generator(np.random.PCG64(23))
    NC = 625

    def unit(a):
        return (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)
    lut_c, lut_u = unit(g.standard_normal((NC, 2048))), unit(g.standard_normal((NC, 2048)))
    crit_c = ref_oim.OIMLoss(2048, NC, scalar=30, momentum=0.5)
    crit_u = ref_oim.OIMLoss(2048, NC, scalar=30, momentum=0.5)
    crit_c.lut.copy_(torch.from_numpy(lut_c)); crit_u.lut.copy_(torch.from_numpy(lut_u))
    tr = ref_trainer.SEQTrainer(cnn, siam, siamv, pairloss.PairLoss(), crit_c, crit_u, None)
    cnn.train(); siam.train(); siamv.train()
    pids = torch.tensor([5, 5, 9, 9, 300, 300, 77, 77][:B])
    clips = synth_clips_structured(B, T, seed=3)
    taps = {}
    h = cnn.register_forward_hook(lambda m, i, o: taps.update(xu=o[0], xc=o[1]))
    loss, p_u, p_v, p_f = tr._forward([clips], pids, 0, 0)
    h.remove()
    taps['xu'].retain_grad(); taps['xc'].retain_grad()
    loss.backward()
    out = {'meta.B': np.array(B), 'meta.T': np.array(T), 'pids': pids.numpy(),
           'lut_seed': np.array(23), 'loss': np.array(loss.item(), np.float64),
           'prec': np.array([float(p_u), float(p_v), float(p_f)]),
           'x_uncorr': taps['xu'].detach().numpy(), 'x_corr_s4': taps['xc'].detach()[..., ::4].numpy(),
           'grad.x_uncorr': taps['xu'].grad.numpy(), 'grad.x_corr_s4': taps['xc'].grad[..., ::4].numpy(),
           'oim_backward_rows': np.array(order)}
    # (featQ.bias / featK.bias sit in front of a train-mode BatchNorm1d, Siamese.py:84-94: analytically zero, fp32 noise)
    grad_record('gs', [(k, p.grad) for k, p in siam.named_parameters() if k not in ('featQ.bias', 'featK.bias')], out)
    grad_record('gv', [(k, p.grad) for k, p in siamv.named_parameters()], out)
    # (parameters whose gradient is analytically zero -- a bias in front of a train-mode BatchNorm -- hold fp32 noise:
    # the conditioned fixture identified them with its float64 run; the same 194 tensors are recorded here)
    live = set(str(k) for k in np.load(os.path.join(HERE, 'grl_train_cond_b8t4.npz'))['meta.keys'])
    grad_record('gc', [(k, p.grad) for k, p in cnn.named_parameters() if k in live], out)
    rows = sorted(set(pids.tolist()))
    out['lut_rows'] = np.array(rows)
    out['lut_c1'], out['lut_u1'] = crit_c.lut[rows].numpy().copy(), crit_u.lut[rows].numpy().copy()
    for name, m in (('siamese', siam), ('siamese_video', siamv)):
        for k, v in m.state_dict().items():
            if 'running' in k:
                out['stat.%s.%s' % (name, k)] = v.numpy().copy()
    # (round 4) the SAME step in float64 -- the yardstick for the CNN gradients under the real 5-term loss: per tensor the
    # float64 samples and the reference's own fp32-vs-float64 error (`ref_l2err`), as the grl_train_cond_* fixtures carry
    # them, so that the checker can hold each tensor to 1e-3 (or 2.5x the reference's own error where that is larger)
    # instead of a bulk bound.  Fresh modules from the same state dicts, everything cast to double; the OIM bridge and the
    # losses are dtype-agnostic.
    fp32_grads = {k: p.grad.detach().clone() for k, p in cnn.named_parameters() if k in live and p.grad is not None}
    cnn64 = ref_models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625)
    cnn64.load_state_dict(synth_state_dict(cnn64, seed=0, profile='conditioned'), strict=True)
    siam64 = ref_models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siam64.load_state_dict(synth_state_dict(siam64, seed=0, prefix='siamese.'), strict=True)
    siamv64 = ref_models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    siamv64.load_state_dict(synth_state_dict(siamv64, seed=0, prefix='siamese_video.'), strict=True)
    cnn64, siam64, siamv64 = cnn64.double(), siam64.double(), siamv64.double()
    c64 = ref_oim.OIMLoss(2048, NC, scalar=30, momentum=0.5).double()
    u64 = ref_oim.OIMLoss(2048, NC, scalar=30, momentum=0.5).double()
    c64.lut.copy_(torch.from_numpy(lut_c).double()); u64.lut.copy_(torch.from_numpy(lut_u).double())
    tr64 = ref_trainer.SEQTrainer(cnn64, siam64, siamv64, pairloss.PairLoss(), c64, u64, None)
    cnn64.train(); siam64.train(); siamv64.train()
    # (dtype bridge only: pairloss.py:26-36 builds its 0/1 label tensor as float32; BCELoss wants the input's dtype)
    import torch.nn.functional as _F
    _bce = _F.binary_cross_entropy
    _F.binary_cross_entropy = lambda inp, tgt, *a, **k: _bce(inp, tgt.to(inp.dtype), *a, **k)
    try:
        loss64, _, _, _ = tr64._forward([clips.double()], pids, 0, 0)
    finally:
        _F.binary_cross_entropy = _bce
    loss64.backward()
    out['f64.loss'] = np.array(loss64.item(), np.float64)
    worst = []
    for k, p64 in cnn64.named_parameters():
        if k not in fp32_grads:
            continue
        g64 = p64.grad.detach().reshape(-1)
        g32 = fp32_grads[k].reshape(-1).double()
        idx = torch.linspace(0, g64.numel() - 1, min(256, g64.numel())).long()
        v64 = g64[idx].numpy()
        out['gc.%s.f64' % k] = v64
        out['gc.%s.ref_l2err' % k] = np.array(np.linalg.norm(g32[idx].numpy() - v64) / max(np.linalg.norm(v64), 1e-300))
        worst.append(float(out['gc.%s.ref_l2err' % k]))
    worst.sort()
    print('trainer step, reference fp32 vs its float64 run: loss %.3e; CNN gradients median %.1e p90 %.1e max %.1e' % (
        abs(loss.item() - loss64.item()) / abs(loss64.item()), worst[len(worst) // 2], worst[int(0.9 * len(worst))], worst[-1]))
    np.savez_compressed(path, **out)
    print('trainer step golden: loss %.6f prec %s, OIM backward order (rows) %s, %d + %d + %d gradient tensors, %d bytes' % (
        loss.item(), out['prec'], order, len(out['gs.keys']), len(out['gv.keys']), len(out['gc.keys']), os.path.getsize(path)))


def cmc_golden(attev, evaf, path):
    """(G) eva_functions.cmc / mean_ap (eva_functions.py:18-115) with the reference's defaults and
    with first_match_break=True, on the same synthetic features as the evaluator fixture."""
    sys.path.insert(0, REPO)
    from grl_amd.synthetic import synth_eval_features
    qf, gf, qp, qc, gp, gc = synth_eval_features(40, 400, seed=1, n_ids=24, noise=7.0)
    dist = attev.cosin_dist(qf, gf)
    np.savez_compressed(path,
                        cmc_default=evaf.cmc(dist, qp, gp, qc, gc, topk=50),
                        cmc_first=evaf.cmc(dist, qp, gp, qc, gc, topk=50, first_match_break=True),
                        cmc_noids=evaf.cmc(dist[:, :40], topk=10),
                        mean_ap=np.array(evaf.mean_ap(dist, qp, gp, qc, gc)))
    print('cmc golden written')


def sign_pattern(n, salt):
    """+-1 per flat index from an integer hash (reproducible on any device): the projection
    <g, sign> is a linear checksum of the WHOLE tensor."""
    i = torch.arange(n, dtype=torch.int64)
    h = (i * 2654435761 + salt * 40503) & 0xFFFFFFFF
    h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
    return (((h >> 13) & 1) * 2 - 1).double()


def train_golden_conditioned(ref_models, path, B=8, T=4, clip_seed=3, xu_stride=1, xc_stride=4):
    """(B') the tight train-parity fixture: 'conditioned' synthetic weights (near-identity residual
    blocks, ReLU inputs shifted positive -- grl_amd/synthetic.py) on structured clips, B x T = 8 x 4.
    Two fp32 runs of a ReLU network differ mostly by ReLU-mask flips of pre-activations within
    rounding of zero (a fraction ~1e-6 of the elements => ~1e-3 relative L2 per layer with the
    default weights, whatever the arithmetic); this profile keeps that floor at ~2e-4, measured
    below as the reference's fp32 run against its own float64 run, so that 1e-3 is a real pin.
    Stored for EVERY parameter gradient: 256 strided samples (fp32 run and float64 run), the full
    L2 norm and four +-1 projections of the whole tensor."""
    sys.path.insert(0, REPO)
    from grl_amd.synthetic import synth_state_dict, synth_clips_structured
    cnn = ref_models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625)
    sd = synth_state_dict(cnn, seed=0, profile='conditioned')
    clips = synth_clips_structured(B, T, seed=clip_seed)
    g = np.random.Generator(np.random.PCG64(7))
    r1 = torch.from_numpy(g.standard_normal((B, 2048)).astype(np.float32))
    r2 = torch.from_numpy(g.standard_normal((B, T, 2048)).astype(np.float32))
    runs = {}
    for dt in (torch.float32, torch.float64):
        cnn.load_state_dict(sd, strict=True)
        cnn.zero_grad(set_to_none=True)
        cnn.to(dt).train()
        xu, xc = cnn(clips.to(dt))
        ((xu * r1.to(dt)).sum() + (xc * r2.to(dt)).sum()).backward()
        runs[dt] = (xu.detach().double(), xc.detach().double(),
                    {k: p.grad.detach().double() for k, p in cnn.named_parameters() if p.grad is not None},
                    {k: v.detach().double().clone() for k, v in cnn.state_dict().items() if 'running' in k or 'tracked' in k})
    a, b = runs[torch.float32], runs[torch.float64]
    # (outputs: every xu_stride-th / xc_stride-th column -- the big-batch fixtures keep a strided sample)
    out = {'meta.B': np.array(B), 'meta.T': np.array(T), 'meta.clip_seed': np.array(clip_seed),
           'meta.xu_stride': np.array(xu_stride), 'meta.xc_stride': np.array(xc_stride),
           'x_uncorr': a[0][..., ::xu_stride].float().numpy(), 'x_corr_s4': a[1][..., ::xc_stride].float().numpy(),
           'f64.x_uncorr': b[0][..., ::xu_stride].float().numpy(), 'f64.x_corr_s4': b[1][..., ::xc_stride].float().numpy()}
    keys, worst = [], 0.0
    for k in a[2]:
        ga, gb = a[2][k].reshape(-1), b[2][k].reshape(-1)
        if float(gb.abs().max()) < 1e-9:            # analytically zero (e.g. a bias in front of a train-mode BN)
            continue
        idx = torch.linspace(0, ga.numel() - 1, min(256, ga.numel())).long()     # the tests rebuild idx
        out['g.%s.val' % k] = ga[idx].float().numpy()
        out['g.%s.f64' % k] = gb[idx].numpy()
        out['g.%s.norm' % k] = np.array([ga.norm().item(), gb.norm().item()])
        out['g.%s.proj' % k] = np.array([[float((x * sign_pattern(x.numel(), s)).sum()) for s in range(4)]
                                         for x in (ga, gb)])
        e = float((ga - gb).norm() / gb.norm())
        out['g.%s.ref_l2err' % k] = np.array(e)     # the reference's own fp32 run vs its float64 run
        worst = max(worst, e)
        keys.append(k)
    for k in TRAIN_STAT_KEYS:
        out['stat.' + k] = a[3][k].numpy()
    out['meta.keys'] = np.array(keys)
    np.savez_compressed(path, **out)
    errs = sorted(float(out['g.%s.ref_l2err' % k]) for k in keys)
    print('conditioned train golden B x T = %d x %d: %d gradient tensors; reference fp32 vs its float64: outputs %.1e / %.1e, '
          'gradient L2 median %.1e p90 %.1e max %.1e; %d bytes' % (
              B, T, len(keys), float((a[0] - b[0]).abs().max() / b[0].abs().max()),
              float((a[1] - b[1]).abs().max() / b[1].abs().max()),
              errs[len(errs) // 2], errs[int(0.9 * len(errs))], errs[-1], os.path.getsize(path)))


def augment_golden(path):
    """(H) the training input transforms as the reference composes them (dataloader.py:51-57:
    RectScale(256,128) -> RandomHorizontalFlip -> RandomSizedEarser -> ToTensor -> Normalize,
    seqtransforms.py:30-216) on synthetic PIL frames, `random` seeded; plus the frame indices of
    VideoDataset's three sampling modes (video_loader.py:30-141).  Frames are 64 x 32 here to keep
    the fixture small (RectScale(64, 32) is then the identity, as RectScale(256,128) is on MARS's
    own 256 x 128 crops); the transforms are size-agnostic."""
    import random
    from PIL import Image
    sys.path.insert(0, REPO)
    sys.path.insert(0, REF)
    from reid.data import seqtransforms as ST
    from grl_amd.synthetic import synth_clips
    H, W, T, N = 64, 32, 2, 8
    u8 = synth_clips(N, T, seed=21, h=H, w=W, raw=True).numpy()                 # [N,T,3,H,W]
    tf = ST.Compose([ST.RectScale(H, W), ST.RandomHorizontalFlip(), ST.RandomSizedEarser(), ST.ToTensor(),
                     ST.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
    random.seed(20251003)
    outs = []
    for n in range(N):
        frames = [Image.fromarray(np.ascontiguousarray(u8[n, t].transpose(1, 2, 0)), 'RGB') for t in range(T)]
        outs.append(torch.stack(tf([frames])[0], 0).numpy())
    out = {'seed': np.array(20251003), 'shape': np.array([N, T, H, W]), 'clips_seed': np.array(21),
           'out': np.stack(outs).astype(np.float32)}
    # frame sampling: VideoDataset.__get_single_item__ on fake tracklets (paths = indices), no images
    from reid.data import video_loader as VL
    np.random.seed(7)
    samp = {}
    for num in (1, 3, 8, 9, 26, 27, 40):
        for S in (4, 8):
            for mode in ('rrs_train', 'rrs_test', 'dense'):
                ds = VL.VideoDataset.__new__(VL.VideoDataset)
                ds.dataset, ds.seq_len, ds.sample, ds.transform = [(tuple(range(num)), 0, 0)], S, mode, None
                opened = []

                class FakeImage(object):
                    @staticmethod
                    def open(p):
                        opened.append(int(p))

                        class Im(object):
                            def convert(self, m):
                                return torch.zeros(1)
                        return Im()
                VL.Image = FakeImage
                try:
                    ds.__get_single_item__(0)
                except Exception:
                    pass                                   # torch.stack on the fake frames may complain; indices are recorded
                samp['idx.%d.%d.%s' % (num, S, mode)] = np.array(opened, np.int32)
    # RectScale itself (seqtransforms.py:30-47) to a 64 x 32 target from six input sizes: up, down,
    # identity, odd, one axis only
    rs = ST.RectScale(64, 32)
    for k, (hh, ww) in enumerate(((32, 16), (100, 50), (64, 32), (75, 29), (64, 40), (90, 32))):
        src = np.random.Generator(np.random.PCG64(100 + k)).integers(0, 256, (hh, ww, 3), dtype=np.uint8)
        res = rs([[Image.fromarray(src, 'RGB')]])[0][0]
        out['rect.%d.shape' % k] = np.array([hh, ww])
        out['rect.%d.out' % k] = np.asarray(res).copy()
    out.update(samp)
    np.savez_compressed(path, **out)
    print('augment golden: flips/erases exercised, %d sampling cases, %d bytes' % (len(samp), os.path.getsize(path)))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REPO)
    from grl_amd.synthetic import synth_state_dict, synth_clips, synth_eval_features
    ref_models, attev, evaf, pairloss, triplet, rerank = import_reference()

    B, T = 2, 4
    cnn = ref_models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625)
    sd = synth_state_dict(cnn, seed=0)
    cnn.load_state_dict(sd, strict=True)          # also proves the key schema matches
    siam = ref_models.create('siamese', input_num=2048, output_num=512, class_num=2)
    ssd = synth_state_dict(siam, seed=0, prefix='siamese.')
    siam.load_state_dict(ssd, strict=True)
    siamv = ref_models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    svd = synth_state_dict(siamv, seed=0, prefix='siamese_video.')
    siamv.load_state_dict(svd, strict=True)
    clips = synth_clips(B, T, seed=0)

    # ---------------- (A) eval forward + clip features -----------------
    out = {}
    taps = {}
    hooks = []
    base = cnn.backbone.base
    for name, mod in (('stem', base[2]), ('pool', base[3]), ('layer1', base[4]),
                      ('layer2', base[5]), ('layer3', base[6]), ('layer4', base[7])):
        hooks.append(mod.register_forward_hook(
            lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    trl = cnn.temporal_learning_block
    catte = {'fwd': [], 'bwd': []}
    hooks.append(trl.channel_atte_foreward_corr.register_forward_hook(
        lambda m, i, o: catte['fwd'].append(o.detach().clone())))
    hooks.append(trl.channel_atte_backward_corr.register_forward_hook(
        lambda m, i, o: catte['bwd'].append(o.detach().clone())))
    trl_out = {}
    hooks.append(trl.register_forward_hook(
        lambda m, i, o: trl_out.update(f_uncorr=o[0].detach().clone(), f_corr=o[1].detach().clone())))
    bb_out = {}
    hooks.append(cnn.backbone.register_forward_hook(
        lambda m, i, o: bb_out.update(corr_map=o[2].detach().clone())))
    cnn.eval(); siam.eval(); siamv.eval()
    with torch.no_grad():
        xu, xc = cnn(clips)
        pooled = siam.self_attention(xc)
        feat = torch.cat((xu, pooled, xc.mean(dim=1)), dim=1)   # attevaluator.py:109-112
    for h in hooks:
        h.remove()
    out['meta.B'] = np.array(B); out['meta.T'] = np.array(T)
    out['x_uncorr'] = xu.numpy(); out['x_corr'] = xc.numpy(); out['feat'] = feat.numpy()
    out['corr_map'] = bb_out['corr_map'].numpy()
    out['f_uncorr'] = trl_out['f_uncorr'].numpy(); out['f_corr'] = trl_out['f_corr'].numpy()
    for k, v in taps.items():
        pack('tap.' + k, sample(v), out)
    for d in ('fwd', 'bwd'):
        out['catte.' + d] = torch.stack(catte[d], 0)[:, :, ::16].numpy()   # [T,B,128]
    np.savez_compressed(os.path.join(HERE, 'grl_eval_b2t4.npz'), **out)
    print('eval golden: |x_uncorr|', xu.norm(dim=1), 'map range',
          bb_out['corr_map'].min().item(), bb_out['corr_map'].max().item())

    # ---------------- (B) train forward + backward ----------------------
    # B x T = 2 x 4 (BatchNorm1d over 2 rows: ill-conditioned in fp32, kept as the shape the
    # eval fixture uses) and 4 x 2 (better conditioned; the tight gradient pin).
    for (Bt, Tt, seed_c, fname) in ((B, T, 0, 'grl_train_b2t4.npz'), (4, 2, 2, 'grl_train_b4t2.npz')):
        train_golden(cnn, sd, synth_clips(Bt, Tt, seed=seed_c), Bt, Tt, os.path.join(HERE, fname))

    # ---------------- (C) Siamese heads ---------------------------------
    g = np.random.Generator(np.random.PCG64(11))
    xs = g.standard_normal((4, T, 2048)).astype(np.float32)
    xs /= np.linalg.norm(xs, axis=2, keepdims=True)
    xs = torch.from_numpy(xs)
    out = {'x': xs.numpy()}
    siam.eval(); siamv.eval()
    with torch.no_grad():
        out['eval.attn'] = siam.self_attention(xs).numpy()
        cls, so = siam(xs)
        out['eval.cls'] = cls.numpy(); out['eval.out'] = so.numpy()
        cls, so = siamv(xs[:, 0])
        out['eval.v_cls'] = cls.numpy(); out['eval.v_out'] = so.numpy()
    siam.load_state_dict(ssd); siam.train()
    xg = xs.clone().requires_grad_(True)
    cls, so = siam(xg)
    rr = torch.from_numpy(g.standard_normal(tuple(so.shape)).astype(np.float32))
    rc = torch.from_numpy(g.standard_normal(tuple(cls.shape)).astype(np.float32))
    ((so * rr).sum() + (cls * rc).sum()).backward()
    out['train.cls'] = cls.detach().numpy(); out['train.out'] = so.detach().numpy()
    out['train.rr'] = rr.numpy(); out['train.rc'] = rc.numpy()
    out['train.grad_x'] = xg.grad.numpy()
    out['train.grad_featQ_w'] = siam.featQ.weight.grad[:, ::64].numpy()
    out['train.grad_cls_w'] = siam.classifierlinear.weight.grad.numpy()
    out['train.featQ_bn_rm'] = siam.featQ_bn.running_mean.numpy()
    np.savez_compressed(os.path.join(HERE, 'siamese_b4t4.npz'), **out)

    # ---------------- (D) evaluator -------------------------------------
    qf, gf, qp, qc, gp, gc = synth_eval_features(40, 400, seed=1, n_ids=24, noise=7.0)
    dist = attev.cosin_dist(qf, gf).numpy()
    cmc, mAP = evaf.evaluate(dist, qp, gp, qc, gc)
    euc = attev.pairwise_distance_tensor(qf, qf).numpy()
    # non-unit-norm rows (dense test_all.py mode averages clips: attevaluator.py:84,95)
    qd = (qf.view(20, 2, -1).mean(1)); gd = torch.cat((qd, gf[40:240]), 0)
    dist_d = attev.cosin_dist(qd, gd).numpy()
    euc_d = attev.pairwise_distance_tensor(qd, gd).numpy()
    np.savez_compressed(os.path.join(HERE, 'evaluator_q40_g400.npz'),
                        dist=dist, indices=np.argsort(dist, axis=1).astype(np.int32),
                        cmc=cmc[:20], mAP=np.array(mAP), euclid_qq=euc,
                        dist_dense=dist_d, euclid_dense=euc_d,
                        idx_dense=np.argsort(dist_d, axis=1).astype(np.int32),
                        idx_dense_euclid=np.argsort(euc_d, axis=1).astype(np.int32))
    print('evaluator golden: mAP %.4f rank1 %.4f' % (mAP, cmc[0]))
    # re-ranking (rerank.py:37-104) exactly as ATTEvaluator.evaluate feeds it (attevaluator.py:150-155),
    # on a smaller case so that the three INPUT matrices can be stored with the output: the
    # neighbour sets are discrete, so the pin must be input-exact.
    qf2, gf2, qp2, qc2, gp2, gc2 = synth_eval_features(16, 120, seed=5, n_ids=10, noise=3.0)
    d2 = attev.cosin_dist(qf2, gf2).numpy()
    qq2 = attev.pairwise_distance_tensor(qf2, qf2).numpy()
    gg2 = attev.pairwise_distance_tensor(gf2, gf2).numpy()
    rr = rerank.re_ranking(d2, qq2, gg2)
    cmc_r, mAP_r = evaf.evaluate(rr, qp2, gp2, qc2, gc2)
    np.savez_compressed(os.path.join(HERE, 'rerank_q16_g120.npz'), dist=d2, qq=qq2, gg=gg2,
                        final=rr.astype(np.float32), cmc=cmc_r[:20], mAP=np.array(mAP_r))
    print('rerank golden: mAP %.4f rank1 %.4f' % (mAP_r, cmc_r[0]))

    # ---------------- (E) losses that still run on this torch -----------
    g = np.random.Generator(np.random.PCG64(13))
    feat = torch.from_numpy(g.standard_normal((8, 2048)).astype(np.float32))
    feat = feat / feat.norm(dim=1, keepdim=True)
    ids = torch.tensor([3, 3, 9, 9, 4, 4, 3, 3])
    torch.Tensor.__xor__  # noqa
    # TripletLoss uses `bool ^ 1` / byte eye, which modern torch rejects; emulate the
    # two masks exactly as triplet.py:30-34 defines them and call its cdist.
    tl = triplet.TripletLoss('soft', True)
    dist_t = tl.cdist(feat, feat)
    same = ids.unsqueeze(1).eq(ids.unsqueeze(0))
    eye = torch.eye(8).bool()
    max_pos = (dist_t * (same ^ eye).float()).max(1)[0]
    min_neg = (dist_t + 1e5 * same.float()).min(1)[0]
    tri = torch.log(1 + torch.exp(max_pos - min_neg))
    score = torch.from_numpy(g.uniform(0.05, 0.95, (4, 4)).astype(np.float32))
    tp = torch.tensor([3, 9, 4, 3]); tg = torch.tensor([3, 9, 4, 3])
    pl, prec = pairloss.PairLoss()(score, tp, tg)
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), feat=feat.numpy(), ids=ids.numpy(),
                        triplet=tri.numpy(), score=score.numpy(), tp=tp.numpy(), tg=tg.numpy(),
                        pair_loss=np.array(pl.item()), pair_prec=np.array(float(prec)))
    train_golden_conditioned(ref_models, os.path.join(HERE, 'grl_train_cond_b8t4.npz'))
    round3_goldens(ref_models)
    oim_golden(ref_models, os.path.join(HERE, 'oim.npz'))
    cmc_golden(attev, evaf, os.path.join(HERE, 'cmc_q40_g400.npz'))
    augment_golden(os.path.join(HERE, 'augment.npz'))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


def round3_goldens(ref_models):
    """BASELINE configs[1] at full size (32 x 4: a strided sample of the outputs), the TRL recurrence length of
    configs[2] (T = 8) in train mode, and one step of the reference's trainer."""
    train_golden_conditioned(ref_models, os.path.join(HERE, 'grl_train_cond_b4t8.npz'), B=4, T=8, clip_seed=5)
    train_golden_conditioned(ref_models, os.path.join(HERE, 'grl_train_cond_b32t4.npz'), B=32, T=4, clip_seed=9,
                             xu_stride=8, xc_stride=32)
    trainer_step_golden(ref_models, os.path.join(HERE, 'trainer_step_cond_b8t4.npz'))


def main_train_only():
    """Regenerates only the two train fixtures (python make_golden.py train)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REPO)
    from grl_amd.synthetic import synth_state_dict, synth_clips
    ref_models = import_reference()[0]
    cnn = ref_models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625)
    sd = synth_state_dict(cnn, seed=0)
    for (Bt, Tt, seed_c, fname) in ((2, 4, 0, 'grl_train_b2t4.npz'), (4, 2, 2, 'grl_train_b4t2.npz')):
        train_golden(cnn, sd, synth_clips(Bt, Tt, seed=seed_c), Bt, Tt, os.path.join(HERE, fname))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'train':
        main_train_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cond':
        torch.manual_seed(0); torch.set_num_threads(8)
        train_golden_conditioned(import_reference()[0], os.path.join(HERE, 'grl_train_cond_b8t4.npz'))
    elif len(sys.argv) > 1 and sys.argv[1] == 'round3':
        torch.manual_seed(0); torch.set_num_threads(8)
        round3_goldens(import_reference()[0])
    elif len(sys.argv) > 1 and sys.argv[1] == 'b64t4':
        # BASELINE configs[3]'s per-GPU batch (64 clips x 4 frames): the reference's fp32 AND float64 runs (the float64
        # run peaks at ~45 GB of host memory; run it alone)
        torch.manual_seed(0); torch.set_num_threads(8)
        train_golden_conditioned(import_reference()[0], os.path.join(HERE, 'grl_train_cond_b64t4.npz'), B=64, T=4,
                                 clip_seed=11, xu_stride=16, xc_stride=64)
    elif len(sys.argv) > 1 and sys.argv[1] == 'trainer':
        torch.manual_seed(0); torch.set_num_threads(8)
        trainer_step_golden(import_reference()[0], os.path.join(HERE, 'trainer_step_cond_b8t4.npz'))
    elif len(sys.argv) > 1 and sys.argv[1] == 'augment':
        import_reference()
        augment_golden(os.path.join(HERE, 'augment.npz'))
    elif len(sys.argv) > 1 and sys.argv[1] == 'cmc':
        r = import_reference()
        cmc_golden(r[1], r[2], os.path.join(HERE, 'cmc_q40_g400.npz'))
    elif len(sys.argv) > 1 and sys.argv[1] == 'oim':
        torch.manual_seed(0)
        oim_golden(import_reference()[0], os.path.join(HERE, 'oim.npz'))
    else:
        main()
