#!/usr/bin/env python3
"""development: eas_conv_dgrad_s2 on the stride-2 layers of config 2 (and a few odd geometries): error against an fp64 reference, a
checksum of the result bits (the forms of the kernel must agree bit for bit: run once per EAS_S2_FORM with a DEV=1 library and compare
the lines) and the HIP-event time per call.

  python scripts/dev_s2d.py [--reps 20] [--small]"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eas_snn_amd import _lib, ops  # noqa: E402

# (NI, Cin, Cout, Hi, Wi): grad_x [NI, Cin, Hi, Wi] from grad_y [NI, Cout, Hi/2, Wi/2]
CONFIG2 = [(64, 32, 64, 128, 160), (192, 64, 128, 64, 80), (192, 128, 256, 32, 40), (192, 256, 512, 16, 20), (64, 128, 128, 32, 40),
           (64, 256, 256, 16, 20)]
SMALL = [(3, 32, 64, 12, 20), (2, 40, 24, 9, 12), (5, 72, 48, 7, 8), (2, 16, 16, 64, 96), (4, 96, 200, 10, 36), (1, 8, 8, 2, 4), (7, 64, 64, 16, 20)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--small', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    for NI, Cin, Cout, H, W in (SMALL if a.small else SMALL + CONFIG2):
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        gy = torch.randn(NI, Cout, Ho, Wo, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
        gx = torch.full((NI, Cin, H, W), float('nan'), device=dev)
        pk = ops.conv_pack_weights(w, 2)

        def call():
            ops.check(L.eas_conv_dgrad_s2(ops.ptr(gy), ops.ptr(pk), ops.ptr(gx), NI, Cin, Cout, H, W, ops.stream()), 'eas_conv_dgrad_s2')
        call()
        torch.cuda.synchronize()
        ref = torch.nn.grad.conv2d_input((NI, Cin, H, W), w.double(), gy.double(), stride=2, padding=1)
        err = float((gx.double() - ref).abs().max() / ref.abs().max())
        digest = hashlib.sha1(gx.cpu().numpy().tobytes()).hexdigest()[:12]
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        print(f'{NI}x{Cout}->{Cin}@{H}x{W}  err {err:.2e}  bits {digest}  {us:8.1f} us', flush=True)


if __name__ == '__main__':
    main()
