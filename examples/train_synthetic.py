#!/usr/bin/env python
"""The reference's training script flow (mars_train.py:46-142) on synthetic pairs, through
the drop-in `reid` / `utils` packages: same factory calls, the same nn.DataParallel wrap,
the same two-group SGD (lr_mult 1 for cnn_model.module.backbone, 2 for the rest), the same
trainer / evaluator / checkpoint calls.  Run:

    PYTHONPATH=.:dropin python examples/train_synthetic.py --epochs 1 --iters 2 -b 4 --seq_len 2
    torchrun --nproc-per-node 8 examples/train_synthetic.py ...      # one process per GPU
"""
import argparse
import os
import os.path as osp
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
for p in (ROOT, osp.join(ROOT, 'dropin')):
    if p not in sys.path:
        sys.path.insert(0, p)

from reid import models                                             # noqa: E402
from reid.loss import PairLoss, OIMLoss                             # noqa: E402
from reid.train import SEQTrainer                                   # noqa: E402
from reid.evaluator import ATTEvaluator                             # noqa: E402
from reid.data import SyntheticPairs                                # noqa: E402
from utils.serialization import load_checkpoint, save_cnn_checkpoint, save_siamese_checkpoint  # noqa: E402


def main(args):
    if args.train_math:
        from grl_amd import train_engine
        train_engine.set_math(args.train_math)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ.get('GRL_SINGLE_DEVICE'):          # functional test on a 1-GPU box: every rank on cuda:0, gloo
            local = 0
        torch.cuda.set_device(local)
        if os.environ.get('GRL_DIST_BACKEND', 'nccl') == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(os.environ['GRL_DIST_BACKEND'])
    device = torch.device('cuda', local)
    num_classes = 625
    from torch.utils.data import DataLoader
    # every rank iterates the SAME global batches; SEQTrainer keeps this rank's pair shard of each
    train_loader = DataLoader(SyntheticPairs(args.iters * args.batch_size // 2, args.seq_len, seed=0, augment=args.augment),
                              batch_size=args.batch_size, drop_last=True)
    query_loader = DataLoader(SyntheticPairs(4, args.seq_len, seed=11), batch_size=8)
    gallery_loader = DataLoader(SyntheticPairs(12, args.seq_len, seed=12), batch_size=8)

    cnn_model = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=num_classes)
    siamese_model = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    siamese_model_uncorr = models.create('siamese_video', input_num=2048, output_num=512, class_num=2)
    cnn_model = torch.nn.DataParallel(cnn_model.to(device), device_ids=[local])   # one device per process
    siamese_model, siamese_model_uncorr = siamese_model.to(device), siamese_model_uncorr.to(device)

    criterion_corr = OIMLoss(2048, num_classes, scalar=30, momentum=0.5).to(device)
    criterion_uncorr = OIMLoss(2048, num_classes, scalar=30, momentum=0.5).to(device)
    criterion_veri = PairLoss().to(device)

    base_param_ids = set(map(id, cnn_model.module.backbone.parameters()))
    new_params = [p for p in cnn_model.parameters() if id(p) not in base_param_ids]
    param_groups = [{'params': cnn_model.module.backbone.parameters(), 'lr_mult': 1},
                    {'params': new_params, 'lr_mult': 2},
                    {'params': siamese_model.parameters(), 'lr_mult': 2},
                    {'params': siamese_model_uncorr.parameters(), 'lr_mult': 2}]
    optimizer = torch.optim.SGD(param_groups, lr=args.lr, momentum=0.9, weight_decay=5e-4, nesterov=True)

    evaluator = ATTEvaluator(cnn_model, siamese_model, only_eval=False)
    trainer = SEQTrainer(cnn_model, siamese_model, siamese_model_uncorr, criterion_veri, criterion_corr,
                         criterion_uncorr, osp.join(args.logs_dir, 'train_log'))
    best_top1 = 0
    for epoch in range(args.epochs):
        lr = args.lr * (0.1 ** (epoch // 15))
        for g in optimizer.param_groups:
            g['lr'] = lr * g.get('lr_mult', 1)
        trainer.train(epoch, train_loader, optimizer)
        top1 = evaluator.evaluate(None, None, query_loader, gallery_loader, args.logs_dir, False, False)
        is_best = top1 >= best_top1
        best_top1 = max(best_top1, top1)
        if rank == 0:
            save_cnn_checkpoint({'state_dict': cnn_model.state_dict(), 'epoch': epoch + 1, 'best_top1': best_top1},
                                is_best, fpath=osp.join(args.logs_dir, 'cnn_checkpoint.pth.tar'))
            save_siamese_checkpoint({'state_dict': siamese_model.state_dict(), 'epoch': epoch + 1,
                                     'best_top1': best_top1}, is_best,
                                    fpath=osp.join(args.logs_dir, 'siamese_checkpoint.pth.tar'))
    if world > 1:
        torch.distributed.barrier()
    # eval-only reload (mars_train.py:38-43) -- on every rank: the evaluator is collective under data parallel
    if True:
        cnn_model.load_state_dict(load_checkpoint(osp.join(args.logs_dir, 'cnnmodel_best.pth.tar'))['state_dict'])
        siamese_model.load_state_dict(
            load_checkpoint(osp.join(args.logs_dir, 'siamesemodel_best.pth.tar'))['state_dict'])
        top1b = evaluator.evaluate(None, None, query_loader, gallery_loader, args.logs_dir, False, False)
        print('best rank-1 accuracy is', top1b)
    return best_top1


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-b', '--batch-size', type=int, default=8)
    ap.add_argument('--seq_len', type=int, default=4)
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--iters', type=int, default=4)
    ap.add_argument('--lr', type=float, default=1e-3)
    ap.add_argument('--augment', action='store_true', help='raw uint8 clips + on-device flip / erase / normalise')
    ap.add_argument('--train-math', default=None, choices=['f32', 'mixed', 'bf16x3', 'bf16', 'bf16s'],
                    help="training GEMM datapath (grl_amd.train_engine.set_math; default exact fp32; 'mixed' = fp32 forward, "
                         "split-bf16 backward GEMMs)")
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--logs-dir', type=str, default='/tmp/grl_logs')
    main(ap.parse_args())
