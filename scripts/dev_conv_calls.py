#!/usr/bin/env python3
"""development: every eas_conv_fwd / eas_conv_dgrad_s2 / eas_conv_wgrad call of one bench.py training step with its geometry and
its duration (HIP events), sorted by time; shows which layers are badly quantised over the 256 CUs."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import eas_snn_amd
from eas_snn_amd import _lib, data, ops
from spikingjelly.activation_based import functional
from yolox.exp import get_exp

dev = torch.device('cuda:0')
eas_snn_amd.hip_library()
ops.set_state_writeback(False)
exp = get_exp(None, 'e-yolox-s')
exp.merge(bench.OPTS)
torch.manual_seed(80)
model = exp.get_model().to(dev)
model.head.use_l1 = True
opt = exp.get_optimizer(64)
ev = data.events_to_device(data.synth_event_batch(64, 200_000, *bench.SENSOR, seed=0), dev)
targets = data.synth_targets(64, bench.CANVAS, dev)
recs = []
L = _lib.lib()


def wrap(name, argnames):
    real = getattr(L, name)

    def f(*args):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = real(*args)
        e.record()
        recs.append((name, tuple(a for a in args if isinstance(a, int) and abs(a) < (1 << 20)), s, e))
        return rc
    setattr(L, name, f)


for n in ('eas_conv_fwd', 'eas_conv_fwd_stats', 'eas_conv_dgrad_s2', 'eas_conv_wgrad_partial', 'eas_conv_fwd_u8'):
    wrap(n, None)


def step():
    frames = data.events_to_frames(ev, exp.Tm, bench.SENSOR, bench.CANVAS)
    out = model(frames, targets)
    opt.zero_grad(set_to_none=True)
    out['total_loss'].backward()
    opt.step()
    functional.reset_net(model)


for _ in range(3):
    step()
torch.cuda.synchronize()
recs.clear()
step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, ints, s, e in recs:
    a = agg[(name, ints)]
    a[0] += 1
    a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print(f'{len(recs)} calls, {tot:.3f} ms')
print('name (NI, Cin, Cout, Hi, Wi, k, stride, x_terms)  calls  ms/call  ms')
for (name, ints), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{name[4:]:20s} {str(ints):44s} {n:3d} {ms / n * 1e3:8.1f} us {ms:7.3f}')
