#!/usr/bin/env python3
"""development: eas_conv_dgrad_small against eas_conv_fwd with mode-1 weights on the stem geometry (HIP-event time per call, max difference)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eas_snn_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
g = torch.Generator().manual_seed(1)
for NI, Cin, Cout, H, W in ((64, 8, 32, 128, 160), (64, 8, 32, 240, 304), (16, 8, 48, 128, 160)):
    gy = torch.randn(NI, Cout, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
    gx = torch.empty(NI, Cin, H, W, device=dev)
    pk = ops.conv_pack_weights(w, 1)

    def new():
        ops.check(L.eas_conv_dgrad_small(ops.ptr(gy), ops.ptr(w), ops.ptr(gx), NI, Cin, Cout, H, W, ops.stream()), 'eas_conv_dgrad_small')

    def old():
        return ops.conv_fwd_packed(gy, pk, None, Cin, 3, 1, 3)
    ref = old()
    new()
    torch.cuda.synchronize()
    diff = float((gx - ref).abs().max() / ref.abs().max())
    res = []
    for fn in (new, old):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 50)
    print(f'{NI}x{Cout}->{Cin}@{H}x{W}: small {res[0]:.1f} us, eas_conv_fwd {res[1]:.1f} us, rel diff {diff:.1e}', flush=True)
