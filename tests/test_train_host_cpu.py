"""CPU tests of the host-side training glue: losses against the reference golden,
pair-granular sharding, and the N>1 gradient all-reduce over gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_losses_match_reference_golden(golden):
    from grl_amd.reid.loss import TripletLoss, PairLoss
    g = golden('losses.npz')
    tri = TripletLoss('soft', True)(torch.from_numpy(g['feat']), torch.from_numpy(g['ids']))
    assert np.allclose(tri.numpy(), g['triplet'], rtol=1e-5, atol=1e-6)
    loss, prec = PairLoss()(torch.from_numpy(g['score']), torch.from_numpy(g['tp']), torch.from_numpy(g['tg']))
    assert abs(loss.item() - float(g['pair_loss'])) < 1e-6
    assert abs(float(prec) - float(g['pair_prec'])) < 1e-6


def test_oim_matches_oracle_restatement():
    """OIM cannot be pinned to the reference (legacy Function); product and oracle are two
    independent restatements of oim.py:14-27,46-53 and must agree."""
    from grl_amd.reid.loss import OIMLoss
    from oracle import grl_oracle as O
    torch.manual_seed(0)
    x = torch.nn.functional.normalize(torch.randn(6, 32), dim=1)
    y = torch.tensor([1, 4, 1, 0, 4, 2])
    crit = OIMLoss(32, 5, scalar=30, momentum=0.5)
    crit.lut.copy_(torch.nn.functional.normalize(torch.randn(5, 32), dim=1))
    lut_o = crit.lut.clone()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    la, _ = crit(xa, y); la.backward()
    lb, _ = O.oim_loss(xb, y, lut_o, 30.0, 0.5); lb.backward()
    assert abs(la.item() - lb.item()) < 1e-6
    assert torch.allclose(xa.grad, xb.grad, atol=1e-6)
    assert torch.allclose(crit.lut, lut_o, atol=1e-6)
    assert torch.allclose(crit.lut[[0, 1, 2, 4]].norm(dim=1), torch.ones(4), atol=1e-6)


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
    from grl_amd.dist import GradBucket, is_distributed
    from grl_amd.reid.loss import OIMLoss
    assert is_distributed()
    torch.manual_seed(0)
    a, b, c = (torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
               torch.nn.Parameter(torch.zeros(2, 2)))
    a.grad = torch.full((5, 3), float(rank + 1))
    b.grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
    # c never receives a gradient (like Siamese.featV): contributes zeros, stays None
    GradBucket([a, b, c]).allreduce_mean()
    # OIM look-up tables stay identical across ranks
    crit = OIMLoss(8, 4, scalar=10, momentum=0.5)
    x = torch.nn.functional.normalize(torch.randn(3, 8) + rank, dim=1).requires_grad_(True)
    y = torch.tensor([rank, 2, 3])
    loss, _ = crit(x, y)
    loss.backward()
    out[rank] = (a.grad.clone(), b.grad.clone(), c.grad, crit.lut.clone())
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    world, port = 2, 29500 + os.getpid() % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ga, gb, gc, _ = out[r]
        assert torch.allclose(ga, torch.full((5, 3), 1.5))
        assert torch.allclose(gb, torch.arange(7, dtype=torch.float32) * 1.5)
        assert gc is None
    assert torch.allclose(out[0][3], out[1][3])
    assert float(out[0][3].abs().sum()) > 0
