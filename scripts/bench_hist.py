#!/usr/bin/env python3
"""Time eas_event_histogram (both forms) at the BASELINE size: 64 samples x 200k events, Tm=4, 240x304 (development tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import eas_snn_amd
from eas_snn_amd import data, ops

dev = torch.device('cuda:0')
eas_snn_amd.hip_library()
ev = data.events_to_device(data.synth_event_batch(64, 200_000, seed=0), dev)
alg = 9 * ev['t'].numel() + 4 * 64 * 4 * 2 * 240 * 304
ref = None
for form in ('scatter', 'banded'):
    os.environ['EAS_HIST_FORM'] = form
    for _ in range(3):
        out = ops.event_histogram(ev['t'], ev['x'], ev['y'], ev['p'], ev['offsets'], 4, 240, 304)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = ops.event_histogram(ev['t'], ev['x'], ev['y'], ev['p'], ev['offsets'], 4, 240, 304)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f'{form:8s} {ms * 1e3:8.1f} us  {alg / ms / 1e6:8.1f} GB/s algorithmic ({alg / 1e6:.0f} MB)')
    if ref is None:
        ref = out.clone()
    else:
        print('forms agree:', bool(torch.equal(ref, out)))
