"""CPU tests of the host-side training glue: losses against the reference golden,
pair-granular sharding, and the N>1 gradient all-reduce over gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_losses_refuse_cpu_tensors():
    """The loss block is HIP only (grl_amd/csrc/loss.hip); the parity tests against the
    reference golden and the oracle are GPU tests (tests/test_gpu_train_kernels.py)."""
    from grl_amd._lib import GrlHipError
    from grl_amd.reid.loss import TripletLoss, PairLoss, OIMLoss
    from grl_amd.reid.loss.pairloss import pair_prob
    ids = torch.tensor([0, 0, 1, 1])
    with pytest.raises(GrlHipError):
        TripletLoss('soft', True)(torch.randn(4, 8), ids)
    with pytest.raises(GrlHipError):
        PairLoss()(torch.rand(2, 2), ids[:2], ids[2:])
    with pytest.raises(GrlHipError):
        OIMLoss(32, 5, scalar=30)(torch.randn(4, 32), ids)
    with pytest.raises(GrlHipError):
        pair_prob(torch.randn(2, 2, 2))
    with pytest.raises(NotImplementedError):
        TripletLoss('soft', False)(torch.randn(4, 8), ids)


def test_shard_rows():
    from grl_amd.dist import shard_rows
    for n, w in ((11310, 8), (7, 3), (2, 4), (16, 4)):
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_shard_pairs():
    from grl_amd.dist import shard_pairs
    assert [shard_pairs(32, r, 4) for r in range(4)] == [(0, 8), (8, 16), (16, 24), (24, 32)]
    with pytest.raises(ValueError):
        shard_pairs(12, 0, 4)
    with pytest.raises(RuntimeError):
        shard_pairs(7, 0, 1)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from grl_amd.dist import is_distributed, gather_rank_order, sharded_distmat
    assert is_distributed()
    torch.manual_seed(0)
    # the OIM look-up tables replay every rank's (feature, label) block in rank order
    x = torch.full((3, 8), float(rank))
    y = torch.tensor([rank, 2, 3])
    xs, ys = gather_rank_order(x, y)
    # gallery-sharded distance matrix (the per-block GEMM is a stand-in here: no GPU on this box)
    g = torch.Generator().manual_seed(3)
    qf, gf = torch.randn(5, 16, generator=g), torch.randn(11, 16, generator=g)
    dm = sharded_distmat(qf, gf, lambda q, gg: -q.mm(gg.t()))
    # GradSync: a tape's flat gradient buffer, released in two pieces while "the backward runs";
    # p (adopted view), q (autograd made a copy) and r (never gets a gradient)
    from grl_amd.dist import GradSync, gather_feature_batches
    p_, q_, r_ = (torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(2, 3)),
                  torch.nn.Parameter(torch.zeros(3)))
    sync = GradSync([p_, q_, r_])
    sync.begin()
    flat = torch.zeros(16)
    for prm, off in ((p_, 0), (q_, 4), (r_, 12)):
        sync.own(prm, flat, off)
    flat[4:10] = float(10 * (rank + 1))            # "TRL" gradients are final first ...
    sync.reduce(flat[4:16], 'late layers')
    flat[0:4] = float(rank + 1)                    # ... the rest of the backward runs meanwhile
    sync.reduce(flat[0:4], 'early layers')
    p_.grad = flat[0:4].view_as(p_)                # autograd adopted the view
    q_.grad = flat[4:10].view_as(q_).clone()       # autograd copied (pre-reduction values)
    sync.finish()
    sync_out = (p_.grad.clone(), q_.grad.clone(), r_.grad, list(sync.launched))
    # the trainer's order: the forward builds the tape BEFORE GradSync.begin(); ownership is declared by
    # Tape.backward from the tape's own offset table, so no gradient is "stray" 
    from grl_amd import train_engine
    tp_params = [torch.nn.Parameter(torch.zeros(6)), torch.nn.Parameter(torch.zeros(3, 3)), torch.nn.Parameter(torch.zeros(5))]
    tape = train_engine.Tape(torch.device('cpu'))
    tape.reserve_param_grads(tp_params, cuts={'early': 0, 'late': 1})        # "forward"
    tape.mark('early'); tape.mark('late')
    sync2 = GradSync(tp_params)
    sync2.begin()                                                          # after the forward, as SEQTrainer does
    tape.pgrad(tp_params[1]).fill_(float(rank + 1))
    tape.pgrad(tp_params[0]).fill_(float(10 * (rank + 1)))
    tape.backward(); tape.flush()
    tp_params[0].grad, tp_params[1].grad = tape.pgrad(tp_params[0]), tape.pgrad(tp_params[1])
    sync2.finish()
    tape_out = (tp_params[0].grad.clone(), tp_params[1].grad.clone(), sync2.stray, sync2.collectives, list(sync2.launched))
    # opt-in global heads (GRL_DP_GLOBAL_HEADS): differentiable rank-ordered all-gather; the backward keeps this
    # rank's slice of the global gradient times the world size (GradSync averages afterwards)
    from grl_amd.dist import gather_global
    xg = torch.full((2, 3), float(rank + 1), requires_grad=True)
    yg = gather_global(xg)
    wg = torch.arange(12, dtype=torch.float32).view(4, 3)
    (yg * wg).sum().backward()
    gg_out = (yg.detach().clone(), xg.grad.clone(), gather_global(torch.tensor([rank, 7])))
    # evaluation features sharded by batch: rank r owns batches r, r + world, ...
    feats = [(i, torch.full((2 + i, 3), float(i)), [i] * (2 + i), [7] * (2 + i)) for i in range(5) if i % world == rank]
    gf, gp, gc = gather_feature_batches(feats, 5)
    out[rank] = (None, None, None, (xs.clone(), ys.clone()), dm.clone(), sync_out,
                 (gf.clone(), gp, gc), tape_out, gg_out)
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    world, port = 2, 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        dm = out[r][4]
        g = torch.Generator().manual_seed(3)
        qf, gf = torch.randn(5, 16, generator=g), torch.randn(11, 16, generator=g)
        assert torch.equal(dm, -qf.mm(gf.t()))
    for r in range(world):
        xs, ys = out[r][3]
        assert torch.equal(xs, torch.cat((torch.zeros(3, 8), torch.ones(3, 8))))
        assert ys.tolist() == [0, 2, 3, 1, 2, 3]
        pg, qg, rg, launched = out[r][5]
        assert torch.equal(pg, torch.full((4,), 1.5)) and torch.equal(qg, torch.full((2, 3), 15.0)) and rg is None
        assert launched == [('late layers', 12), ('early layers', 4)]
        g0, g1, stray, ncoll, launched = out[r][7]
        assert torch.equal(g0, torch.full((6,), 15.0)) and torch.equal(g1, torch.full((3, 3), 1.5))
        assert stray == 0 and launched == [('late', 20), ('early', 8)]      # 4-float aligned slots: 8 | 12 + 8
        assert ncoll == 2 + 2                       # two buckets + the one-off None-pattern check (MIN, MAX)
        yg, xgrad, lab = out[r][8]
        assert torch.equal(yg, torch.tensor([[1.0] * 3] * 2 + [[2.0] * 3] * 2)) and lab.tolist() == [0, 7, 1, 7]
        assert torch.equal(xgrad, torch.arange(12, dtype=torch.float32).view(4, 3)[2 * r:2 * r + 2] * 2)
        gf, gp, gc = out[r][6]
        assert gf.shape == (2 + 3 + 4 + 5 + 6, 3) and gp == [i for i in range(5) for _ in range(2 + i)]
        assert torch.equal(gf[:, 0], torch.tensor([float(i) for i in range(5) for _ in range(2 + i)])) and set(gc) == {7}


def test_sharded_pair_sampler_and_batches():
    """Every rank iterates the same pair stream and keeps whole pairs of each global batch."""
    from grl_amd.dist import ShardedPairSampler, PairShardedBatches
    stream = [v for pair in range(20) for v in (100 + pair, 200 + pair)]      # (index, positive), 20 pairs
    got = [list(ShardedPairSampler(stream, 8, rank=r, world=2)) for r in range(2)]
    assert len(got[0]) == len(got[1]) == len(ShardedPairSampler(stream, 8, rank=0, world=2)) == 20
    for r in range(2):
        assert all(b - a == 100 for a, b in zip(got[r][0::2], got[r][1::2]))        # pairs stay together
    merged = []                                        # local batches of 4 re-interleave to the global batches
    for k in range(5):
        merged += got[0][4 * k:4 * k + 4] + got[1][4 * k:4 * k + 4]
    assert merged == stream
    with pytest.raises(ValueError):
        ShardedPairSampler(stream, 12, rank=0, world=4)
    loader = [(torch.arange(8).view(8, 1), torch.arange(8), torch.zeros(8))]
    parts = [list(PairShardedBatches(loader, rank=r, world=2))[0] for r in range(2)]
    assert parts[0][1].tolist() == [0, 1, 2, 3] and parts[1][1].tolist() == [4, 5, 6, 7]


def test_rank_batch_sampler_shards_the_loader_not_its_output():
    """Evaluation under world > 1: the rank's DataLoader is rebuilt around a rank-filtered batch sampler, so a
    rank's __getitem__ calls cover only its own batches (the host decode must scale)."""
    from torch.utils.data import DataLoader, Dataset
    from grl_amd.dist import shard_loader_batches

    class Counting(Dataset):
        def __init__(self):
            self.seen = []

        def __len__(self):
            return 26

        def __getitem__(self, i):
            self.seen.append(i)
            return torch.tensor([i]), i % 7, i % 2

    for world in (2, 3):
        got = {}
        for rank in range(world):
            ds = Counting()
            dl = shard_loader_batches(DataLoader(ds, batch_size=4), rank, world)
            assert len(dl) == len([i for i in range(7) if i % world == rank])
            got[rank] = [b[0].view(-1).tolist() for b in dl]
            assert sorted(ds.seen) == sorted(x for b in got[rank] for x in b)        # nothing else was loaded
        full = [list(range(i, min(i + 4, 26))) for i in range(0, 26, 4)]
        assert [got[i % world][i // world] for i in range(7)] == full
    assert shard_loader_batches('abc', 0, 1) == 'abc'
    assert list(shard_loader_batches(iter(range(7)), 1, 3)) == [1, 4]
    # a shuffling sampler draws its own permutation per process: the loader is NOT re-built (batch i % world would
    # not partition the data); its output is filtered and the generator the caller seeded is still the one in use
    g = torch.Generator().manual_seed(5)
    shuffled = DataLoader(Counting(), batch_size=4, shuffle=True, generator=g)
    a = [b[0].view(-1).tolist() for b in shard_loader_batches(shuffled, 0, 2)]
    g.manual_seed(5)
    ref = [b[0].view(-1).tolist() for b in shuffled]
    assert a == ref[0::2]
    # the re-built loader keeps generator / pin-memory device of the original
    g2 = torch.Generator().manual_seed(1)
    rebuilt = shard_loader_batches(DataLoader(Counting(), batch_size=4, generator=g2), 1, 2)
    assert rebuilt.generator is g2 and len(rebuilt) == 3


def test_bench_launcher_starts_n_fresh_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent starts
    `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` as a child (it never touches the GPU
    itself), the two ranks rendezvous on 127.0.0.1 and rank 0 prints ONE line (--dry-run: gloo, no GPU)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    line.pop('preflight')                                   # (round 5: the first-contact report, tested below)
    assert line == {"dry_run": True, "world_size": 2, "n_gpus": 2}
    # the command the launcher builds
    import importlib.util
    spec = importlib.util.spec_from_file_location('grl_bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    real = subprocess.call
    subprocess.call = lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0
    try:
        assert bench.launch_ranks(8, ['--gpus', '8', '--steps', '3']) == 0
    finally:
        subprocess.call = real
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '8', '--steps', '3']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


@pytest.mark.parametrize('fault', ['hang', 'raise'])
def test_bench_train_block_failure_is_visible_in_the_exit_code(fault):
    """N > 1: a hung or raising train block (the RCCL all-reduce step) must turn the launcher's exit code red while
    stdout still carries ONE JSON line with `train.error` (VERDICT round 3 item 6 / ADVICE: the watchdog used to
    os._exit(0)).  Two gloo ranks through the real launcher; rank 1 sleeps past the time limit / raises."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(GRL_BENCH_DRY_TRAIN=fault, GRL_BENCH_TRAIN_TIMEOUT='4')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0, res.stdout
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line['dry_run'] and 'error' in line['train']
    assert ('did not finish' in line['train']['error']) if fault == 'hang' else ('injected failure' in line['train']['error'])


_GUARD_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
from grl_amd.rank_guard import TrainBlockGuard
rank, fault, d, key, delay = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5])
time.sleep(delay)                                   # skew between the ranks' guard.start()
g = TrainBlockGuard(rank, 2, 1.5, lambda reason: print(json.dumps({"error": reason}), flush=True), directory=d, key=key)
g.start()
if rank == 1 and fault == 'raise':
    g.leave("train block raised on rank 1: RuntimeError('injected failure')")
if rank == 1 and fault == 'hang':
    time.sleep(60)
# a healthy rank sits in the collective until the watchdog takes it out (peer failure or the limit)
time.sleep(60)
"""


@pytest.mark.parametrize('fault', ['raise', 'hang'])
def test_rank_guard_protocol_20_launches_under_load(fault, tmp_path):
    """rank 0's guard used to delete `.err` / `.out` in start(); a peer that raised first
    lost its reason and rank 0 printed the time-out text (1 of 8 full-suite runs).  20 launches of the two-rank
    protocol (light children: grl_amd.rank_guard imports neither torch nor the GPU) with the failing rank AHEAD of
    rank 0 by a random skew, while four busy loops load the machine: every launch must end with both ranks on exit
    code 3 and rank 0's single line carrying the FIRST failure's reason."""
    import json
    import random
    import subprocess
    burners = [subprocess.Popen([sys.executable, '-c', 'while True: pass']) for _ in range(4)]
    rnd = random.Random(7)
    child = _GUARD_CHILD % {'root': ROOT}
    try:
        for it in range(20):
            key = 'guard_%s_%d' % (fault, it)
            skew = rnd.choice([0.0, 0.05, 0.3, 0.8])                      # rank 0 starts its guard this much LATER
            procs = [subprocess.Popen([sys.executable, '-c', child, str(r), fault, str(tmp_path), key,
                                       str(skew if r == 0 else 0.0)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                     for r in (0, 1)]
            outs = [p.communicate(timeout=60) for p in procs]
            assert [p.returncode for p in procs] == [3, 3], (it, outs)
            lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
            assert len(lines) == 1 and outs[1][0].strip() == '', (it, outs)
            err = json.loads(lines[0])['error']
            if fault == 'raise':
                assert err.startswith("train block raised on rank 1: RuntimeError('injected failure')"), (it, skew, err)
            else:
                assert 'did not finish' in err, (it, skew, err)
    finally:
        for b in burners:
            b.kill()
        for b in burners:
            b.wait()


def test_rank_guard_ignores_files_of_an_earlier_launch(tmp_path):
    """A `.err` older than the launcher process is a leftover (same port / run id / pid reused): it neither takes a
    healthy rank out nor shadows this launch's own reason."""
    import time as _t
    from grl_amd import rank_guard
    got = []
    g = rank_guard.TrainBlockGuard(0, 2, 30, got.append, directory=str(tmp_path), key='stale', exit_fn=lambda c: got.append(c))
    with open(g.err, 'w') as fh:
        fh.write('old failure')
    old = g.not_before - 100
    os.utime(g.err, (old, old))
    assert not g._fresh(g.err)
    g.start(); _t.sleep(0.6)
    assert got == []                                               # the watchdog did not fire on the stale file
    g.leave('new failure')
    assert got == ['new failure', rank_guard.EXIT_TRAIN_BLOCK_FAILED]
    assert open(g.err).read() == 'new failure'
    g.finished()


def test_bench_train_block_ok_path_exits_zero():
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(GRL_BENCH_DRY_TRAIN='ok', GRL_BENCH_TRAIN_TIMEOUT='60')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['train'] == {"ok": True}


def test_trainer_helpers_on_cpu_tensors():
    """SEQTrainer._top1 without the OIM criterion's read-out falls back to the reference's accuracy() (topk / eq /
    sum, eva_functions.py:118-131), and with it uses the count; the head fork is a no-op on CPU tensors."""
    import torch
    from grl_amd.reid.train.trainer import SEQTrainer, _HeadFork
    from grl_amd.reid.evaluator import accuracy
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(12, 7, generator=g)
    y = torch.randint(0, 7, (12,), generator=g)
    want = accuracy(logits, y)[0]
    assert float(SEQTrainer._top1(logits, y)) == float(want)
    logits.grl_top1 = (torch.tensor(5.0), 12)                     # what grl_amd.reid.loss.oim attaches
    assert abs(float(SEQTrainer._top1(logits, y)) - 5.0 / 12) < 1e-7
    fk = _HeadFork(logits)
    assert fk.on is False
    with fk:
        z = logits + 1
    fk.join()
    assert torch.equal(z, logits + 1)


def test_bench_preflight_reports_every_rank():
    """N > 1: the first-contact step (a checked 1 MB all-reduce + what every rank sits on) is part of the line --
    here on two gloo ranks through the launcher's dry run (no GPU); `--no-preflight` leaves it out."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = json.loads(res.stdout.decode().strip().splitlines()[-1])
    pre = line['preflight']
    assert pre['ranks'] == 2 and pre['backend'] == 'gloo'
    assert sorted(r['rank'] for r in pre['per_rank']) == [0, 1]
    assert len({r['pid'] for r in pre['per_rank']}) == 2
    assert pre['allreduce_1MB_ms']['second'] > 0
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--no-preflight'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0
    assert 'preflight' not in json.loads(res.stdout.decode().strip().splitlines()[-1])
