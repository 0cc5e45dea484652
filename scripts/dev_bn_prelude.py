#!/usr/bin/env python3
"""development: eas_bn_lif_fwd_ex on one layer with (a) final statistics, (b) 64-chunk partials of eas_bn_stats_partial, (c) nb partials of a
convolution epilogue -- what the in-block finalize costs.  Usage: python scripts/dev_bn_prelude.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from eas_snn_amd import _lib, ops

dev = torch.device('cuda:0')
L = _lib.lib()
for (T, N, Cc, H, W, nb) in ((3, 64, 128, 32, 40, 384), (3, 64, 64, 64, 80, 1536), (3, 64, 256, 16, 20, 96), (3, 64, 512, 8, 10, 60), (3, 64, 64, 32, 40, 384)):
    HW = H * W
    y = torch.randn(T, N, Cc, H, W, device=dev)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    mean, invstd = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    spikes = torch.empty_like(y)
    w = torch.tensor(0.3, device=dev)
    ws = torch.zeros(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=dev)
    chunks = L.eas_bn_stats_partial(ops.ptr(y), 0, T * N, Cc, HW, ops.ptr(ws), ops.stream())
    big = torch.zeros(Cc * nb * 2, dtype=torch.float64, device=dev)
    big.view(Cc, nb, 2)[:, :, 0] = 0.0
    big.view(Cc, nb, 2)[:, :, 1] = float(T * N * HW) / nb
    pends = {'final': None,
             'chunks64': _lib.EasBnPending(ws.data_ptr(), chunks, 1, float(T * N * HW), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), None, 0),
             f'conv{nb}': _lib.EasBnPending(big.data_ptr(), nb, 1, float(T * N * HW), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), None, nb)}
    out = []
    for name, pend in pends.items():
        def run():
            rc = L.eas_bn_lif_fwd_ex(ops.ptr(y), 0, ops.ptr(mean), ops.ptr(invstd), ops.ptr(gamma), ops.ptr(beta), None, None, ops.ptr(w), 0.0, 1.0, 0.0,
                                     ops.FLAG_DECAY_INPUT, ops.ptr(spikes), None, T, N, Cc, HW, 0, C.byref(pend) if pend is not None else None, None, 0, None,
                                     ops.stream())
            assert rc == 0, rc
        for _ in range(5):
            run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            run()
        e.record()
        torch.cuda.synchronize()
        out.append(f'{name} {s.elapsed_time(e) / 50 * 1e3:.1f} us')
    print((T, N, Cc, H, W), ' | '.join(out), flush=True)
