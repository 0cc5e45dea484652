#!/usr/bin/env python3
"""development: eas_bn_lif_bwd through the fused BatchNorm + PLIF layer at backbone shapes, per form (EAS_BNLIF_BWD = two / fused / unset):
HIP-event time of the backward call.  usage: EAS_BNLIF_BWD=<form> python scripts/dev_bnlif_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eas_snn_amd  # noqa: E402,F401
from eas_snn_amd import ops  # noqa: E402
from spikingjelly.activation_based import layer, neuron, surrogate  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for T, N, C, H, W in ((3, 64, 128, 32, 40), (3, 64, 64, 32, 40), (3, 64, 256, 16, 20), (3, 64, 128, 16, 20), (3, 64, 32, 64, 80)):
    y = (torch.randn(T, N, C, H, W, generator=g) * 1.5 + 0.3).to(dev)
    go = torch.randn(T, N, C, H, W, generator=g).to(dev)
    bn = layer.BatchNorm2d(C, eps=1e-3, momentum=0.03, step_mode='m').to(dev).train()
    node = neuron.ParametricLIFNode(init_tau=2.0, decay_input=False, v_reset=None, surrogate_function=surrogate.ATan(2.0), step_mode='m').to(dev)
    times = []
    for it in range(6):
        yy = y.clone().requires_grad_(True)
        s = bn.fused_with(node, yy)
        torch.cuda.synchronize()
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        s.backward(go)
        torch.cuda.synchronize()
        ops.set_timer(None)
        recs = timer.rec.get('eas_bn_lif_bwd', [])
        times.append(sum(r[0].elapsed_time(r[1]) for r in recs) * 1e3)
    print(f'T{T} N{N} C{C} {H}x{W}: {min(times[2:]):.1f} us  (form {os.environ.get("EAS_BNLIF_BWD", "default")})', flush=True)
