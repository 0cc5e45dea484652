"""Checkpoint load/save (reference: yolox/utils/checkpoint.py): shape-tolerant load, latest/best files."""
import os
import shutil

import torch


def load_ckpt(model, ckpt):
    state = model.state_dict()
    keep = {}
    for k, v in state.items():
        if k in ckpt and ckpt[k].shape == v.shape:
            keep[k] = ckpt[k]
    model.load_state_dict(keep, strict=False)
    return model


def save_checkpoint(state, is_best, save_dir, model_name=''):
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, model_name + '_ckpt.pth')
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(save_dir, 'best_ckpt.pth'))
