#!/usr/bin/env python3
"""development: convolutions of one training forward whose input holds small integers (spikes, SEW sums) but arrives WITHOUT the
small-integer tag (they run the six-term path where three terms are exact).  usage: dev_lost_tags.py [config]"""
import os
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

import eas_snn_amd  # noqa
from eas_snn_amd import ops, workloads

config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = workloads.get(config)
dev = torch.device('cuda:0')
exp = workloads.build_exp(w)
exp.ema = False
exp.output_dir = '/tmp/eas_lost_tags'
tr = exp.get_trainer(types.SimpleNamespace(batch_size=4, fp16=False, experiment_name='lt', ckpt=None, resume=False))
model = tr.setup()
for n, m in model.named_modules():
    object.__setattr__(m, '_eas_name', n)
names = {id(m): n for n, m in model.named_modules()}
raw, inputs_fn = workloads.device_inputs(w, 4, 200_000, dev)
found = []


def looks_int(x):
    x = ops.dense(x)
    return bool((x == x.round()).all()) and float(x.abs().max()) <= 16 and float(x.abs().max()) > 0


orig = ops.conv2d


def conv2d(x, conv, small_int=None):
    si = ops.is_small_int(x) if small_int is None else small_int
    if not si and looks_int(x):
        found.append((names.get(id(conv), '?'), tuple(x.shape)))
    return orig(x, conv, small_int)


orig_dual = ops.conv2d_dual


def conv2d_dual(x, a, b, owner=None, key=None):
    if not ops.is_small_int(x) and looks_int(x):
        found.append((names.get(id(a), '?') + ' | dual', tuple(x.shape)))
    return orig_dual(x, a, b, owner, key)


ops.conv2d, ops.conv2d_dual = conv2d, conv2d_dual
x, tg = inputs_fn()
out = model(x, tg)
print('loss', float(out['total_loss']))
print('convolutions fed untagged small-integer tensors:', len(found))
for f in found:
    print('  ', f)
