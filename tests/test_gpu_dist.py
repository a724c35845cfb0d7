"""Data-parallel training on ONE MI355X: two processes share cuda:0 and talk over gloo (device
tensors are staged through the host by grl_amd.dist) -- the same GradSync / tape code path the
8-GPU RCCL run takes, minus the transport.  RCCL itself runs here only in a world of ONE rank
(test_rccl_backend_in_a_world_of_one_rank: the real call sequence on the `nccl` backend, every collective an identity)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T = 8, 2                       # global batch: 4 pairs -> 2 pairs per rank


def _models(dev):
    import contextlib, io
    from grl_amd.reid import models
    from grl_amd.synthetic import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siam = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamv = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn.load_state_dict(synth_state_dict(cnn, seed=0, profile='conditioned'))
    siam.load_state_dict(synth_state_dict(siam, seed=0, prefix='siamese.'))
    siamv.load_state_dict(synth_state_dict(siamv, seed=0, prefix='siamese_video.'))
    return cnn.to(dev), siam.to(dev), siamv.to(dev)


def _trainer(dev):
    from grl_amd.reid.train import SEQTrainer
    from grl_amd.reid.loss import OIMLoss, PairLoss
    cnn, siam, siamv = _models(dev)
    crit_c, crit_u = OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev), OIMLoss(2048, 625, scalar=30, momentum=0.5).to(dev)
    tr = SEQTrainer(cnn, siam, siamv, PairLoss().to(dev), crit_c, crit_u, None)
    params = [p for m in (cnn, siam, siamv) for p in m.parameters()]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
    return tr, opt, (cnn, siam, siamv), (crit_c, crit_u)


def _batches():
    from grl_amd.synthetic import synth_clips_structured
    out = []
    for step in range(2):
        clips = synth_clips_structured(B, T, seed=40 + step)
        pids = torch.tensor([5, 5, 9, 9, 300, 300, 77, 77]) + step
        out.append((clips, pids, torch.zeros(B, dtype=torch.long)))
    return out


def _snapshot(mods, crits):
    sd = {}
    for i, m in enumerate(mods):
        for k, v in m.state_dict().items():
            sd['m%d.%s' % (i, k)] = v.detach().cpu().clone()
    for i, c in enumerate(crits):
        sd['lut%d' % i] = c.lut.detach().cpu().clone()
    return sd


def _worker(rank, world, port, out, backend='gloo'):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    if backend == 'nccl':
        os.environ['GRL_SYNC_FORCE'] = '1'
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda:0'))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda:0')
    tr, opt, mods, crits = _trainer(dev)
    assert tr.device.type == 'cuda'
    launched = []
    snaps = []
    g0 = {}

    def grab(optimizer, args, kwargs):                 # averaged gradients as the optimizer sees them, step 0
        if not g0:
            g0.update({'m%d.%s' % (i, k): p.grad.detach().cpu().clone() for i, m in enumerate(mods)
                       for k, p in m.named_parameters() if p.grad is not None})
    opt.register_step_pre_hook(grab)
    for step, batch in enumerate(_batches()):
        tr.train(step, [batch], opt)              # the trainer keeps this rank's pair shard of the global batch
        launched.append(list(tr._bucket.launched))
        # every gradient is tape-owned: no per-parameter fall-back reduction; 6 buckets (+ the one-off check)
        assert tr._bucket.stray == 0, tr._bucket.stray
        assert tr._bucket.collectives == len(tr._bucket.launched) + (2 if step == 0 else 0), tr._bucket.collectives
        snaps.append(_snapshot(mods, crits))
    out[rank] = (snaps, launched, g0)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_on_one_device_matches_averaged_gradients():
    world, port = 2, 29700 + os.getpid() % 1500
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (s0, l0, dp_grads), (s1, l1, _) = out[0], out[1]
    # (1) both ranks hold identical parameters, BN-free buffers aside, and identical OIM tables after each step
    for step in range(2):
        for k in s0[step]:
            if 'running_' in k or 'num_batches' in k:
                continue                                   # per-rank BatchNorm statistics (= DataParallel replicas)
            assert torch.equal(s0[step][k], s1[step][k]), (step, k)
    # (2) the buckets went out in backward order: TRL + tail, layer 4 + GCE, layer 3, layers 2/1 + stem, then the
    #     two Siamese tapes -- 54.76 M values in all
    labels = [lab for lab, _ in l0[0]]
    assert labels[-4:] == ['trl', 'layer4', 'layer3', 'stem'] or set(['trl', 'layer4', 'layer3', 'stem']) <= set(labels)
    order = [lab for lab in labels if lab in ('trl', 'layer4', 'layer3', 'stem')]
    assert order == ['trl', 'layer4', 'layer3', 'stem']
    assert sum(n for _, n in l0[0]) >= 54758726
    # (3) step 1 equals a single process that runs the two shards one after the other from the same
    #     initial weights, averages the gradients and takes the same SGD step.  The OIM backward reads
    #     its look-up table AFTER the clip-level update of the same step (oim.py:23-26), and in data
    #     parallel that update replays every rank's (feature, label) block in rank order -- the
    #     emulation feeds gather_rank_order the other shard's block, exactly what the peer would send.
    from grl_amd import dist as grl_dist
    import importlib
    oim_mod = importlib.import_module('grl_amd.reid.loss.oim')
    dev = torch.device('cuda:0')
    tr, opt, mods, crits = _trainer(dev)
    params = [p for m in mods for p in m.parameters()]
    init = [{k: v.clone() for k, v in m.state_dict().items()} for m in mods]
    clips, pids, _ = _batches()[0]
    calls = [[], []]                                   # per shard: (x, y) of every OIM call, forward order
    cur = [0]
    real_fwd = oim_mod.OIMLoss.forward

    def rec_fwd(self, inputs, targets):
        calls[cur[0]].append((inputs.contiguous().detach(), targets.detach().clone()))
        return real_fwd(self, inputs, targets)
    oim_mod.OIMLoss.forward = rec_fwd
    losses = []
    try:
        for r in range(world):
            for m, sd in zip(mods, init):
                m.load_state_dict(sd, strict=True)
                m.train()
            cur[0] = r
            lo, hi = r * B // world, (r + 1) * B // world
            losses.append(tr._forward([clips[lo:hi].to(dev)], pids[lo:hi].to(dev), 0, 0)[0])
    finally:
        oim_mod.OIMLoss.forward = real_fwd
    where = {x.data_ptr(): k for r in range(world) for k, (x, y) in enumerate(calls[r])}

    def fake_gather(x, y, group=None):
        k = where[x.data_ptr()]
        return (torch.cat([calls[r][k][0] for r in range(world)]),
                torch.cat([calls[r][k][1].to(y.dtype) for r in range(world)]))
    real_gather = grl_dist.gather_rank_order
    grl_dist.gather_rank_order = fake_gather
    grads = []
    try:
        for r in range(world):
            for c in crits:
                c.lut.zero_()
            opt.zero_grad()
            losses[r].backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in params])
    finally:
        grl_dist.gather_rank_order = real_gather
    names = ['m%d.%s' % (i, k) for i, m in enumerate(mods) for k, _ in m.named_parameters()]
    gerr = {}
    for nme, g0, g1 in zip(names, grads[0], grads[1]):
        if g0 is not None:
            avg = ((g0 + g1) * 0.5).cpu()
            gerr[nme] = float((dp_grads[nme] - avg).abs().max() / avg.abs().max().clamp_min(1e-30))
    assert len(gerr) >= 190 and max(gerr.values()) < 1e-6, sorted(gerr.items(), key=lambda e: -e[1])[:5]
    for m, sd in zip(mods, init):
        m.load_state_dict(sd, strict=True)
    opt.zero_grad()
    for p, g0, g1 in zip(params, grads[0], grads[1]):
        if g0 is not None:
            p.grad = (g0 + g1) * 0.5
    opt.step()
    worst = 0.0
    for mi, m in enumerate(mods):
        for k, v in m.named_parameters():
            ref = v.detach().cpu()
            got = s0[0]['m%d.%s' % (mi, k)]
            worst = max(worst, float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
    print('two-rank step vs averaged single-process gradients: worst relative parameter difference %.2e' % worst)
    assert worst < 1e-6


def _eval_worker(rank, world, port, out, backend='gloo'):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    if backend == 'nccl':
        os.environ['GRL_SYNC_FORCE'] = '1'
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda:0'))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    out[rank] = _run_eval()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def _run_eval():
    import contextlib, io
    from torch.utils.data import DataLoader
    from grl_amd.reid.evaluator import ATTEvaluator
    from grl_amd.reid.data import SyntheticPairs
    dev = torch.device('cuda:0')
    cnn, siam, _ = _models(dev)
    ev = ATTEvaluator(cnn, siam, only_eval=False)
    q = DataLoader(SyntheticPairs(5, T, seed=11), batch_size=4)           # 10 clips: batches of 4, 4, 2
    g = DataLoader(SyntheticPairs(13, T, seed=12), batch_size=4)          # 26 clips: 7 batches, ragged tail
    qf, qp, qc = ev.extract_feature(q)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r1 = ev.evaluate(None, None, q, g, None, False, False)
    lines = [l for l in buf.getvalue().splitlines() if l.startswith(('Mean AP', 'Rank-'))]
    return qf.cpu(), list(qp), list(qc), float(r1), lines


def test_two_rank_evaluation_equals_single_process():
    """ATTEvaluator under world_size 2 (batches dealt round-robin to the ranks, feature rows and the
    gallery-sharded distance matrix all-gathered) prints the same metrics and returns bit-identical
    features as a single process."""
    world, port = 2, 31300 + os.getpid() % 1500
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_eval_worker, args=(world, port, out), nprocs=world, join=True)
    ref = _run_eval()
    for r in range(world):
        assert torch.equal(out[r][0], ref[0]) and out[r][1] == ref[1] and out[r][2] == ref[2]
        assert out[r][3] == ref[3] and out[r][4] == ref[4] and len(ref[4]) == 5


def test_rccl_backend_in_a_world_of_one_rank():
    """The `nccl` (= RCCL) backend on this 1-GPU box: a process group of ONE rank, collectives forced
    (GRL_SYNC_FORCE=1) -- the exact call sequence of the 8-GPU run (async all-reduce of flat gradient slices
    released by the tape while the backward -- TRL side stream, weight-gradient stream -- is still running; device
    all-gathers in the evaluator and the OIM replay), minus the peers.  With one rank every collective is an identity,
    so two trainer steps must leave bit-identical parameters / LUTs to a plain single-process run, and the evaluator
    the same features and metrics."""
    world, port = 1, 32900 + os.getpid() % 1500
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_worker, args=(world, port, out, 'nccl'), nprocs=1, join=True)
    snaps, launched, _ = out[0]
    assert [l for l, _ in launched[0]] == ['trl', 'layer4', 'layer3', 'stem', 'rest'][:len(launched[0])] or len(launched[0]) >= 4
    dev = torch.device('cuda:0')
    tr, opt, mods, crits = _trainer(dev)
    for step, batch in enumerate(_batches()):
        tr.train(step, [batch], opt)
        ref = _snapshot(mods, crits)
        for k in ref:
            assert torch.equal(ref[k], snaps[step][k]), (step, k)
    out2 = ctx.Manager().dict()
    mp.spawn(_eval_worker, args=(world, port + 1, out2, 'nccl'), nprocs=1, join=True)
    ref = _run_eval()
    assert torch.equal(out2[0][0], ref[0]) and out2[0][1:] == ref[1:]


def test_bench_gpus_2_prints_one_line_with_a_train_block():
    """The driver's form `python bench.py --gpus N`: the parent starts N fresh ranks (before any GPU call) and
    rank 0 prints ONE line.  On this 1-GPU box both ranks share cuda:0 and talk over gloo
    (GRL_SINGLE_DEVICE=1 GRL_DIST_BACKEND=gloo); the line must say n_gpus 2, carry the process group's world
    size, and a `train` block (configs[3] per GPU) whose gradient buckets were reduced with no stray
    per-parameter collectives."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(GRL_SINGLE_DEVICE='1', GRL_DIST_BACKEND='gloo')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['world_size'] == 2 and line['dist_backend'] == 'gloo'
    assert line['metric'] == 'clip-features/sec' and line['value'] > 0 and line['scaling'] == 'weak'
    tr = line['train']
    assert tr['n_gpus'] == 2 and tr['dist_backend'] == 'gloo'
    for m in ('f32', 'mixed'):
        blk = tr[m]
        assert blk['stray_reductions'] == 0
        labs = [lab for lab, _ in blk['allreduce_buckets']]      # the two Siamese tapes ('rest') run first in the backward
        assert [l for l in labs if l != 'rest'] == ['trl', 'layer4', 'layer3', 'stem'] and labs.count('rest') == 2
        assert blk['allreduce_bytes_per_step'] >= 4 * 54758726
        assert blk['gradsync_collectives_per_step'] == len(blk['allreduce_buckets'])
        assert blk['clips_per_sec'] > 0


def _global_heads_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GRL_DP_GLOBAL_HEADS='1')
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from grl_amd import dist as grl_dist
    dev = torch.device('cuda:0')
    tr, opt, mods, crits = _trainer(dev)
    for m in mods:
        m.train()
    clips, pids, _ = _batches()[0]
    lo, hi = grl_dist.shard_pairs(B, rank, world)
    loss = tr._forward([clips[lo:hi].to(dev)], pids[lo:hi].to(dev), 0, 0)[0]
    sync = grl_dist.GradSync([p for m in mods for p in m.parameters()])
    opt.zero_grad()
    sync.begin()
    loss.backward()
    sync.finish()
    grads = {'m%d.%s' % (i, k): p.grad.detach().cpu().clone() for i, m in enumerate(mods) for k, p in m.named_parameters()
             if p.grad is not None}
    out[rank] = (float(loss), grads, [c.lut.detach().cpu().clone() for c in crits])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def test_global_heads_switch_equals_gathered_batch_loss():
    """GRL_DP_GLOBAL_HEADS=1 (the reference's nn.DataParallel semantics: only the CNN is replicated, the Siamese
    heads / verification / triplet mining / OIM see the gathered batch -- mars_train.py:80-82, trainer.py:137-162):
    two ranks on one device give, on every rank, the loss and the averaged gradients of a single process that runs
    the CNN on the two shards (per-replica BatchNorm) and the heads + loss block ONCE on the concatenated outputs."""
    world, port = 2, 33900 + os.getpid() % 1500
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_global_heads_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
    assert all(torch.equal(a, b) for a, b in zip(out[0][2], out[1][2]))
    # single-process emulation
    from grl_amd.reid.evaluator import accuracy      # noqa: F401  (import side effects only)
    dev = torch.device('cuda:0')
    tr, opt, mods, crits = _trainer(dev)
    cnn, siam, siamv = mods
    for m in mods:
        m.train()
    clips, pids, _ = _batches()[0]
    half = B // world
    outs = [cnn(clips[r * half:(r + 1) * half].to(dev)) for r in range(world)]
    xu, xc = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])

    class _Shim(torch.nn.Module):
        def forward(self, x):
            return xu, xc
    real = tr.model
    tr.model = _Shim()
    try:
        loss = tr._forward([clips.to(dev)], pids.to(dev), 0, 0)[0]
    finally:
        tr.model = real
    opt.zero_grad()
    loss.backward()
    assert abs(float(loss) - out[0][0]) <= 1e-6 * abs(float(loss)), (float(loss), out[0][0])
    worst = 0.0
    for i, m in enumerate(mods):
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            ref = p.grad.detach().cpu()
            got = out[0][1]['m%d.%s' % (i, k)]
            worst = max(worst, float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
    print('global heads: worst relative gradient difference vs the gathered-batch single process %.2e' % worst)
    assert worst < 1e-5
    assert all(float((a - c.lut.cpu()).abs().max()) < 1e-6 for a, c in zip(out[0][2], crits))
