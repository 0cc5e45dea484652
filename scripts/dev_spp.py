#!/usr/bin/env python3
"""development: the SPP block on spike planes at the config-2 geometry (192 x 256 channels at 8x10, pools 5 / 9 / 13): forward and backward
HIP-event times, gradient against three nn.MaxPool2d + cat."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eas_snn_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
g = torch.Generator().manual_seed(4)
for N, C, H, W in ((192, 256, 8, 10), (160, 384, 8, 10), (96, 384, 12, 20)):
    x = ((torch.rand(N, C, H, W, generator=g) < 0.3).float() + (torch.rand(N, C, H, W, generator=g) < 0.1).float()).to(dev)
    go = torch.randn(N, 4 * C, H, W, generator=g).to(dev)
    gx = torch.empty_like(x)
    xs = ops.to_planes(x)

    def bwd():
        ops.check(L.eas_spp_pool_planes_bwd(ops.ptr(xs), ops.ptr(go), ops.ptr(gx), N, C, H, W, 5, 9, 13, ops.stream()), 'spp bwd')
    bwd()
    xr = x.clone().requires_grad_(True)
    ref = torch.cat([xr] + [torch.nn.functional.max_pool2d(xr, k, 1, k // 2) for k in (5, 9, 13)], 1)
    ref.backward(go)
    err = float((gx - xr.grad).abs().max())
    for _ in range(3):
        bwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    print(f'{N}x{C}@{H}x{W}: backward {e0.elapsed_time(e1) * 50:.1f} us, max abs diff to ATen {err:.1e}', flush=True)
