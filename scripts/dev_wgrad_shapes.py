#!/usr/bin/env python3
"""development: eas_conv_wgrad on the 3x3 / 1x1 layer shapes of a configuration, one row per shape: time, algorithmic bytes and flops, the
HBM and matrix-core roofline times.  usage: dev_wgrad_shapes.py [s|m] [planes 0/1] [only this row]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from eas_snn_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else 'm'
planes = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
dev = torch.device('cuda:0')
if which == 'm':
    NI, b = 160, 48          # config 3: T=5 x 32 samples, width 0.75
else:
    NI, b = 192, 32          # config 2: T=3 x 64 samples, width 0.5
shapes = []
for lvl, (H, W) in enumerate([(64, 80), (32, 40), (16, 20), (8, 10)]):
    c = 2 * b * 2 ** lvl
    shapes += [(f'dark{lvl + 2}.0 s2', NI, c // 2, c, 2 * H, 2 * W, 3, 2, True), (f'dark{lvl + 2}.m 3x3', NI, c // 2, c // 2, H, W, 3, 1, True),
               (f'dark{lvl + 2}.c12 1x1', NI, c, c, H, W, 1, 1, True), (f'dark{lvl + 2}.c3 1x1', NI, c, c, H, W, 1, 1, True),
               (f'dgrad-like real 3x3 {c // 2}', NI, c // 2, c // 2, H, W, 3, 1, False)]
shapes.append(('stem 8->%d real' % b, NI // (5 if which == 'm' else 3), 8, b, 128, 160, 3, 1, False))
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
for idx, (name, NI_, Cin, Cout, H, W, k, s, sp) in enumerate(shapes):
    if only is not None and idx != only:
        continue
    x = (torch.rand(NI_, Cin, H, W, device=dev) < 0.2).float() if sp else torch.randn(NI_, Cin, H, W, device=dev)
    gy = torch.randn(NI_, Cout, (H + s - 1) // s, (W + s - 1) // s, device=dev)
    x_sp = ops.to_planes(x) if (sp and planes and Cin % 8 == 0) else None
    xt = 1 if sp else 3
    try:
        for _ in range(2):
            ops.conv_wgrad(x, gy, k, s, xt, x_sp)
    except Exception as e:       # noqa
        print(name, 'unsupported', e)
        continue
    torch.cuda.synchronize()
    a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.conv_wgrad(x, gy, k, s, xt, x_sp)
    bb.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(bb) / 5
    nbytes = (2 if x_sp is not None else 4) * x.numel() + 4 * gy.numel()
    fl = 2.0 * gy.numel() * Cin * k * k
    terms = 3 if sp else 6
    t_h, t_m = nbytes / 8e12 * 1e3, fl * terms / 2.5e15 * 1e3
    print(f'{name:28s} NI {NI_} {Cin:4d}->{Cout:4d} {H:3d}x{W:3d} k{k} s{s}  {ms:7.3f} ms  hbm {t_h:6.3f} mfma {t_m:6.3f}  frac {max(t_h, t_m) / ms:5.2f}  '
          f'{fl / ms / 1e9:6.1f} TF  {nbytes / ms / 1e6:7.1f} GB/s')
