#!/usr/bin/env python3
"""development (GPU box): every convolution-family C-ABI call of one eager training step with its geometry and its HIP-event duration, in call
order -- which launches are far from the time their matrix-core work needs (under-filled grids, long K loops in few blocks).
usage: dev_conv_calls_table.py [config] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import eas_snn_amd
from eas_snn_amd import ops, workloads

FWD = {'eas_conv_fwd': (4, 12, 11), 'eas_conv_fwd_stats': (3, 11, 10), 'eas_conv_fwd_planes': (4, 11, None), 'eas_conv_dgrad_s2': (3, 8, None),
       'eas_conv_dgrad_small': (3, 8, None)}
WG = {'eas_conv_wgrad_partial': (3, 11), 'eas_conv_wgrad_planes_partial': (3, 10)}


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    w = workloads.get(config)
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else w['batch']
    dev = torch.device('cuda:0')
    torch.cuda.set_stream(torch.cuda.Stream())
    ops.set_state_writeback(False)
    ops.WGRAD_SIDE_BATCH = 0          # one stream: event pairs then measure kernels, not overlap
    trainer, model, step = workloads.build_trainer(w, batch, dev, 200_000, out_dir='/tmp/eas_calls_table')
    for _ in range(3):
        step.eager()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    with ops.kernel_trace() as tr:
        step.eager()
    torch.cuda.synchronize()
    ops.set_timer(None)
    rows = []
    for fam, names in (('eas_conv_fwd', set(FWD) | {'eas_conv_fwd_group'}), ('eas_conv_wgrad', set(WG) | {'eas_conv_wgrad_group_partial'})):
        calls = [(n, a) for n, a in tr.calls if n in names]
        recs = [r for r in timer.rec.get(fam, []) if r[3] > 0]          # (start, end, bytes, flops, issue): the slab reductions carry no flops
        assert len(calls) == len(recs), (fam, len(calls), len(recs))
        for (n, a), (s, e, nbytes, fl, isf) in zip(calls, recs):
            us = s.elapsed_time(e) * 1e3
            if n.endswith('group') or n.endswith('group_partial'):
                geo = ' | '.join(f'{q.NI}x{q.Cin}->{q.Cout}@{q.Hi}x{q.Wi}' for q in a[0][:a[1]]) + f' k{a[2]}'
            elif n in FWD:
                lo, hi, _ = FWD[n]
                g = a[lo:hi]
                geo = (f'{g[0]}x{g[1]}->{g[2]}@{g[3]}x{g[4]}' + (f' k{g[5]} s{g[6]}' if len(g) > 6 else (' k3 s2 dgrad' if n == 'eas_conv_dgrad_s2' else ' k3 s1 dgrad (stacked taps)')) + (f' xt{g[7]}' if len(g) > 7 else ''))
            else:
                lo, hi = WG[n]
                g = a[lo:hi]
                geo = f'{g[0]}x{g[1]}->{g[2]}@{g[3]}x{g[4]} k{g[5]} s{g[6]}' + (f' xt{g[7]}' if len(g) > 7 else ' planes')
            rows.append((us, fam, n.replace('eas_conv_', ''), geo, fl, isf, nbytes))
    tot = sum(r[0] for r in rows)
    print(f'config {config} batch {batch}: {len(rows)} convolution-family calls, {tot / 1e3:.3f} ms')
    print('     us   need_us  excess  entry                 geometry')
    rows.sort(key=lambda r: -(r[0] - max(r[5] / 1.3e15, r[6] / 5e12) * 1e6))
    for us, fam, n, geo, fl, isf, nbytes in rows:
        need = max(isf / 1.3e15, nbytes / 5e12) * 1e6       # practical ceilings: 1.3 PFLOP/s bf16 issue, 5 TB/s
        print(f'{us:7.1f} {need:8.1f} {us - need:7.1f}  {n:21s} {geo}')


if __name__ == '__main__':
    main()
