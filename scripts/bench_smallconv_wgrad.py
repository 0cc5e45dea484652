#!/usr/bin/env python3
"""Times the two weight-gradient kernels of the sampler's tiny-channel convolutions (EAS_SW_FORM=fma|mfma) at the bench shapes
(256 images of 256x320: 4 micro-slices x batch 64) and checks them against each other.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eas_snn_amd import ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    shapes = ((4, 4, 5, 256, 256, 320), (2, 4, 5, 256, 256, 320), (-2, 4, 5, 256, 256, 320), (4, 4, 7, 64, 256, 320), (2, 4, 3, 256, 192, 256))
    if os.environ.get('EAS_SW_FIRST'):          # profiling runs: the bench shape only
        shapes = shapes[:1]
    for cin, cout, k, N, H, W in shapes:
        spikes = cin < 0                      # negative channel count: spike-valued x (exact in one bf16 term)
        cin = abs(cin)
        x = torch.randn(N, cin, H, W, device=dev)
        if spikes:
            x = (x > 0.5).float()
        gy = torch.randn(N, cout, H, W, device=dev)
        w = torch.zeros(cout, cin, k, k, device=dev)
        res = {}
        for form in ('fma', 'mfma'):
            os.environ['EAS_SW_FORM'] = form
            for _ in range(3):
                gw, gb = ops.smallconv_bwd_weight(gy, x, w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gw, gb = ops.smallconv_bwd_weight(gy, x, w)
            e1.record()
            torch.cuda.synchronize()
            res[form] = (e0.elapsed_time(e1) / 10 * 1e3, gw.clone(), gb.clone())
        dw = (res['fma'][1] - res['mfma'][1]).abs().max().item() / res['fma'][1].abs().max().item()
        db = (res['fma'][2] - res['mfma'][2]).abs().max().item() / res['fma'][2].abs().max().item()
        gb_bytes = 4 * (x.numel() + gy.numel())
        print(f'{cin}->{cout} k{k} N{N} {H}x{W}{" spikes" if spikes else ""}: fma {res["fma"][0]:.1f} us, mfma {res["mfma"][0]:.1f} us '
              f'({gb_bytes / res["mfma"][0] / 1e3:.0f} GB/s algorithmic), max rel diff w {dw:.2e} b {db:.2e}', flush=True)


if __name__ == '__main__':
    main()
