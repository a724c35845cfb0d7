"""Checkpoint helpers with the reference's names and file layout
(utils/serialization.py:25-59; the tensorboard clean-up helper is a no-op here)."""
import json
import os.path as osp
import shutil

import torch

from .osutils import mkdir_if_missing


def read_json(fpath):
    with open(fpath, 'r') as f:
        return json.load(f)


def write_json(obj, fpath):
    mkdir_if_missing(osp.dirname(fpath))
    with open(fpath, 'w') as f:
        json.dump(obj, f, indent=4, separators=(',', ': '))


def _save(state, is_best, fpath, best_name):
    mkdir_if_missing(osp.dirname(fpath))
    torch.save(state, fpath)
    if is_best:
        shutil.copy(fpath, osp.join(osp.dirname(fpath), best_name))


def save_cnn_checkpoint(state, is_best, fpath='checkpoint.pth.tar'):
    _save(state, is_best, fpath, 'cnnmodel_best.pth.tar')


def save_siamese_checkpoint(state, is_best, fpath='checkpoint.pth.tar'):
    _save(state, is_best, fpath, 'siamesemodel_best.pth.tar')


def load_checkpoint(fpath):
    if osp.isfile(fpath):
        # reference checkpoints carry numpy scalars ('best_top1'): plain pickle load, as upstream
        checkpoint = torch.load(fpath, map_location='cpu', weights_only=False)
        print("=> Loaded checkpoint '{}'".format(fpath))
        return checkpoint
    raise ValueError("=> No checkpoint found at '{}'".format(fpath))


def remove_repeat_tensorboard_files(path):
    return None
