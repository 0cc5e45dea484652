#!/usr/bin/env python3
"""development: the sampler's 5x5 convolutions, vector-ALU form against matrix-core form (EAS_SC_FORM is read per call): error vs fp64 and
time at the bench shapes.  usage: python scripts/dev_smallconv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from eas_snn_amd import ops

dev = torch.device('cuda:0')


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (N, cin, cout, H, W, spikes) in ((64, 2, 4, 256, 320, True), (64, 4, 4, 256, 320, False), (256, 2, 4, 256, 320, False), (256, 4, 4, 256, 320, False),
                                     (3, 4, 4, 37, 50, False), (2, 2, 2, 16, 130, False)):
    g = torch.Generator().manual_seed(N + cin)
    x = (torch.rand(N, cin, H, W, generator=g) < 0.2).float() if spikes else torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 5, 5, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    gy = torch.randn(N, cout, H, W, generator=g)
    xd, wd, bd, gyd = x.to(dev), w.to(dev), b.to(dev), gy.to(dev)
    nref = min(N, 4)
    ref = F.relu(F.conv2d(x[:nref].double(), w.double(), b.double(), padding=2))
    mask = (torch.rand(N, cin, H, W, generator=g) < 0.5).float().to(dev)
    refg = torch.nn.grad.conv2d_input(x[:nref].shape, w.double(), gy[:nref].double(), padding=2) * mask[:nref].cpu().double()
    out = []
    for form in ('fma', 'mfma'):
        os.environ['EAS_SC_FORM'] = form
        y = ops.smallconv_fwd(xd, wd, bd, relu=True)
        gx = ops.smallconv_bwd_input(gyd, wd, relu_mask=mask)
        e1 = (y[:nref].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        e2 = (gx[:nref].double().cpu() - refg).abs().max().item() / refg.abs().max().item()
        t1 = timeit(lambda: ops.smallconv_fwd(xd, wd, bd, relu=True))
        t2 = timeit(lambda: ops.smallconv_bwd_input(gyd, wd, relu_mask=mask))
        out.append(f'{form}: fwd {t1:7.1f} us err {e1:.1e} | dgrad {t2:7.1f} us err {e2:.1e}')
    print((N, cin, cout, H, W), ' || '.join(out), flush=True)
