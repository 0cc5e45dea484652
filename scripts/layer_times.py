#!/usr/bin/env python3
"""development: per-layer GPU time of the dense convolutions of one training step (HIP events around the C-ABI calls, tagged with the
BaseConv that issued them; the dual convolutions of CSPLayer / head towers are issued outside a BaseConv and appear under the tag of
the previous layer; the ONE batched slab reduction of all weight gradients at the end of the backward pass carries the last tag seen --
the stem -- so the stem's "eas_conv_wgrad, 2 calls" row is its own weight gradient, ~0.15 ms, plus that reduction, ~0.5 ms).
usage: layer_times.py [config] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import types

import torch

import eas_snn_amd
from eas_snn_amd import ops, workloads


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    w = workloads.get(config)
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else w['batch']
    dev = torch.device('cuda:0')
    torch.cuda.set_stream(torch.cuda.Stream())
    ops.set_state_writeback(False)
    ops.WGRAD_SIDE_BATCH = 0          # one stream: event pairs then measure kernels, not overlap
    exp = workloads.build_exp(w)
    exp.ema = False
    exp.output_dir = '/tmp/eas_layer_times'
    torch.manual_seed(80)
    tr = exp.get_trainer(types.SimpleNamespace(batch_size=batch, fp16=False, experiment_name='lt', ckpt=None, resume=False))
    model = tr.setup()
    model.head.use_l1 = True
    for n, m in model.named_modules():
        object.__setattr__(m, '_eas_name', n)
    raw, inputs_fn = workloads.device_inputs(w, batch, 200_000, dev)
    step = tr.step_fn(inputs_fn)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    reps = 3
    for _ in range(reps):
        step.eager()
    torch.cuda.synchronize()
    ops.set_timer(None)
    rows = []
    for (tag, name), items in timer.tagged.items():
        ms = sum(a.elapsed_time(b) for a, b, *_ in items) / reps
        fl = sum(it[3] for it in items) / reps
        isf = sum(it[4] for it in items) / reps
        # per-launch roofline time: algorithmic bytes at 8 TB/s or bf16 term products at 2.5 PFLOP/s, whichever is longer
        roof = sum(max(it[2] / 8e12, it[4] / 2.5e15) for it in items) / reps * 1e3
        hbm_bound = sum(1 for it in items if it[2] / 8e12 >= it[4] / 2.5e15) * 2 > len(items)
        rows.append((ms, tag, name, len(items) // reps, fl, isf, roof, hbm_bound))
    tot = sum(r[0] for r in rows if r[2] in ('eas_conv_fwd', 'eas_conv_wgrad'))
    print(f'config {config} batch {batch}: tagged conv time {tot:.2f} ms/step; all tagged calls {sum(r[0] for r in rows):.2f} ms/step')
    by = os.environ.get('EAS_LT_SORT', 'excess')          # excess: time above 1.5 x the launch's roofline time (what a good kernel could return)
    rows.sort(key=(lambda r: -(r[0] - 1.5 * r[6])) if by == 'excess' else (lambda r: -r[0]))
    print('    time   roofline  excess  calls  entry point            TF     bound  layer')
    for ms, tag, name, calls, fl, isf, roof, hbm_bound in rows[:int(os.environ.get('EAS_LT_ROWS', '70'))]:
        tf = fl / ms / 1e9 if ms > 0 and fl > 0 else 0.0
        print(f'{ms:7.3f} ms {roof:7.3f} {ms - 1.5 * roof:7.3f}  {calls:3d}  {name:24s} {tf:6.1f}  {"hbm " if hbm_bound else "mfma"}  {tag}')

if __name__ == '__main__':
    main()
