#!/usr/bin/env python
"""The reference's evaluation flow (test_all.py:76-120 / mars_train.py:110-114) on synthetic
tracklets through the drop-in `reid` package: build the three models, (optionally) load the
checkpoints the training script wrote, and run ATTEvaluator in the rrs-test mode or the dense
`only_eval` mode, with or without k-reciprocal re-ranking.  Everything after the loader --
features, distance matrices, ranking, CMC/mAP, re-ranking -- runs on the MI355X.

    PYTHONPATH=.:dropin python examples/eval_synthetic.py --queries 8 --gallery 40 --rerank
    PYTHONPATH=.:dropin python examples/eval_synthetic.py --uint8            # raw pixels to the stem
"""
import argparse
import os.path as osp
import sys

import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
for p in (ROOT, osp.join(ROOT, 'dropin')):
    if p not in sys.path:
        sys.path.insert(0, p)

from reid import models                                             # noqa: E402
from reid.evaluator import ATTEvaluator                             # noqa: E402
from reid.data import SyntheticPairs                                # noqa: E402
from utils.serialization import load_checkpoint                     # noqa: E402


def main(args):
    device = torch.device('cuda', 0)
    cnn_model = models.create('resnet50_grl', num_features=2048, dropout=0, numclasses=625, pretrained=False)
    siamese_model = models.create('siamese', input_num=2048, output_num=512, class_num=2)
    if args.cnn_ckpt:
        state = load_checkpoint(args.cnn_ckpt)['state_dict']
        cnn_model.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in state.items()})
        siamese_model.load_state_dict(load_checkpoint(args.siamese_ckpt)['state_dict'])
    else:
        from grl_amd.synthetic import synth_state_dict
        cnn_model.load_state_dict(synth_state_dict(cnn_model, seed=0))
        siamese_model.load_state_dict(synth_state_dict(siamese_model, seed=0, prefix='siamese.'))
    cnn_model, siamese_model = cnn_model.to(device), siamese_model.to(device)

    from torch.utils.data import DataLoader
    q = SyntheticPairs(args.queries // 2, args.seq_len, seed=1, raw=args.uint8)
    g = SyntheticPairs(args.gallery // 2, args.seq_len, seed=2, raw=args.uint8)
    if args.dense:          # test_all.py mode: one tracklet per item, [1, n_clips, T, 3, H, W]
        class Tracklets(torch.utils.data.Dataset):
            def __init__(self, base, clips):
                self.base, self.clips = base, clips

            def __len__(self):
                return len(self.base) // self.clips

            def __getitem__(self, i):          # tracklets 2k and 2k+1: one identity seen by two cameras
                items = [self.base[i * self.clips + c] for c in range(self.clips)]
                return torch.stack([it[0] for it in items]), i // 2, i % 2
        q, g = Tracklets(q, 2), Tracklets(g, 2)
        loaders = DataLoader(q, batch_size=1), DataLoader(g, batch_size=1)
    else:
        loaders = DataLoader(q, batch_size=args.batch_size), DataLoader(g, batch_size=args.batch_size)
    evaluator = ATTEvaluator(cnn_model, siamese_model, only_eval=args.dense)
    rank1 = evaluator.evaluate(None, None, loaders[0], loaders[1], None, False, args.rerank)
    print('Rank-1: %.4f' % float(rank1))
    return float(rank1)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=24)
    ap.add_argument('--gallery', type=int, default=120)
    ap.add_argument('--seq_len', type=int, default=4)
    ap.add_argument('-b', '--batch_size', type=int, default=32)
    ap.add_argument('--dense', action='store_true', help='test_all.py mode (only_eval=True): all clips of a tracklet')
    ap.add_argument('--rerank', action='store_true')
    ap.add_argument('--uint8', action='store_true', help='feed raw uint8 clips (normalised inside the stem)')
    ap.add_argument('--cnn_ckpt', default='')
    ap.add_argument('--siamese_ckpt', default='')
    main(ap.parse_args())
