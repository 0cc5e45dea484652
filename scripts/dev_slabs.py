"""development: split-K slab traffic of the weight gradients of one training step (slabs x weight bytes per layer)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eas_snn_amd  # noqa: E402
from eas_snn_amd import ops, workloads  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream())
w = workloads.get(cfg)
trainer, model, step = workloads.build_trainer(w, w['batch'], dev)
rows = []
orig = ops._partial_call


def rec(name, nbytes, fn, *args, **kw):
    rc = orig(name, nbytes, fn, *args, **kw)
    NI, Cin, Cout, H, W, k, s = args[3:10]
    rows.append((rc * Cout * Cin * k * k * 4 / 1e6, rc, fn.__name__, NI, Cin, Cout, H, W, k, s))
    return rc


with ops.no_state_writeback():
    step.eager()
    from eas_snn_amd import ops_conv
    ops_conv._partial_call = rec            # (the name the weight-gradient operators call lives in their family module)
    step.eager()
    ops_conv._partial_call = orig
torch.cuda.synchronize()
rows.sort(reverse=True)
print(f'config {cfg}: {len(rows)} weight gradients, slabs written + read once each: {sum(r[0] for r in rows):.0f} MB per step')
for r in rows[:30]:
    print('%8.1f MB  %3d slabs  %-30s NI %d %d->%d %dx%d k%d s%d' % r)
