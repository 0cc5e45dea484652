#!/usr/bin/env python3
"""Host-side cost of enqueueing one eager training step (development tool): wall time of step.eager() without waiting for the GPU,
and a cProfile of where the host spends it.  usage: dev_host.py [config]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import eas_snn_amd  # noqa: F401
from eas_snn_amd import ops, workloads


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device('cuda:0')
    torch.cuda.set_stream(torch.cuda.Stream())
    ops.set_state_writeback(False)
    w = workloads.get(config)
    trainer, model, step = workloads.build_trainer(w, w['batch'], dev, 200_000, out_dir='/tmp/eas_dev_host')
    for _ in range(4):
        step.eager()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step.eager()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print('enqueue ms / total ms per step:', [(round(a, 1), round(b, 1)) for a, b in ts])
    calls = []
    orig = ops._call

    def counting(name, *a, **k):
        calls.append(name)
        return orig(name, *a, **k)
    from eas_snn_amd import ops_bn, ops_conv, ops_core, ops_det, ops_events, ops_glue, ops_lif, ops_sampler
    fams = (ops, ops_core, ops_lif, ops_bn, ops_sampler, ops_conv, ops_events, ops_glue, ops_det)     # every family module calls its own ``_call`` name
    for m_ in fams:
        m_._call = counting
    step.eager()
    for m_ in fams:
        m_._call = orig
    print('C-ABI calls through ops._call in one step:', len(calls))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step.eager()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(30)


if __name__ == '__main__':
    main()
