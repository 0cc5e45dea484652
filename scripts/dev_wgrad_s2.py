#!/usr/bin/env python3
"""development: the stride-2 3x3 weight gradients of config 2 (slab kernel + reduction through ops.conv_wgrad): HIP-event time per call and
the error against fp64.  Spike-plane inputs for the backbone layers, fp32 three-term inputs for dark2.0 and the neck's bottom-up convolutions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eas_snn_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(2)
# NI, Cin, Cout, H, W, planes
CASES = [(64, 32, 64, 128, 160, False), (192, 64, 128, 64, 80, True), (192, 128, 256, 32, 40, True), (192, 256, 512, 16, 20, True),
         (64, 128, 128, 32, 40, False), (64, 256, 256, 16, 20, False)]
for NI, Cin, Cout, H, W, planes in CASES:
    if planes:
        x = (torch.rand(NI, Cin, H, W, generator=g) < 0.2).float().to(dev)
        ops.mark_small_int(x)
        xs = ops.to_planes(x)
    else:
        x = torch.randn(NI, Cin, H, W, generator=g).to(dev)
        xs = None
    gy = torch.randn(NI, Cout, H // 2, W // 2, generator=g).to(dev)

    def call():
        return ops.conv_wgrad(x, gy, 3, 2, 1 if planes else 3, xs)
    gw = call()
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), gy.double(), stride=2, padding=1)
    err = float((gw.double() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f'{NI}x{Cin}->{Cout}@{H}x{W} {"planes" if planes else "fp32"}: {e0.elapsed_time(e1) * 50:.1f} us  err {err:.1e}', flush=True)
