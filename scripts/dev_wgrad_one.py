#!/usr/bin/env python3
"""development: a few eas_conv_wgrad calls of one scripts/dev_conv.py bench shape (driver for scripts/pmc_kernel.sh).  usage: dev_wgrad_one.py <idx>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv, idx = sys.argv[:1], int(sys.argv[1])
import torch

import dev_conv
from eas_snn_amd import ops

name, NI, Cin, Cout, H, W, k, s, sp = dev_conv.bench_shapes()[idx]
x, w = dev_conv.make(NI, Cin, Cout, H, W, k, sp)
gy = torch.randn(NI, Cout, (H + s - 1) // s, (W + s - 1) // s, device=dev_conv.dev)
for _ in range(4):
    ops.conv_wgrad(x, gy, k, s, 1 if sp else 3)
torch.cuda.synchronize()
print('ran wgrad', name)
