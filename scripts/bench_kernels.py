#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written kernels at the benchmark shapes (HIP-event timing, median of repeats).
Development tool; prints achieved GFLOP/s or GB/s per kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import eas_snn_amd  # noqa
from eas_snn_amd import ops


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def smallconv(which):
    dev = torch.device('cuda:0')
    for (N, cin, cout) in ((256, 2, 4), (256, 4, 4), (64, 2, 4), (64, 4, 4)):
        H, W, k = 256, 320, 5
        x = torch.randn(N, cin, H, W, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.1
        b = torch.randn(cout, device=dev)
        gy = torch.randn(N, cout, H, W, device=dev)
        flops = 2.0 * N * H * W * cin * cout * k * k
        for name, fn in (('fwd', lambda: ops.smallconv_fwd(x, w, b, relu=True)),
                         ('dgrad', lambda: ops.smallconv_bwd_input(gy, w, x)),
                         ('wgrad', lambda: ops.smallconv_bwd_weight(gy, x, w))):
            if which and name not in which:
                continue
            ms = timeit(fn)
            print(f'smallconv {name:5s} N={N:3d} {cin}->{cout} k{k}: {ms:7.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s' .replace('TFLOP/s', 'GFLOP/ms')
                  + f'  = {flops / (ms * 1e-3) / 1e12:6.2f} TFLOP/s')


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'smallconv'
    if what == 'smallconv':
        smallconv(sys.argv[2:])
