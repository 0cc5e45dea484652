#!/usr/bin/env python3
"""development (GPU box): which Python lines of the eager training step launch the ATen glue kernels (fills, adds, copies, sums).
torch.profiler with stacks over ONE eager step of the bench's trainer; device time of every aten:: operator grouped by the innermost
frame inside this repository.
usage: dev_aten_sources.py [config] [eval]      (eval: one eager eval forward + reset_net, as the evaluator runs it, instead of a training step)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import eas_snn_amd
from eas_snn_amd import ops, workloads


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    w = workloads.get(config)
    dev = torch.device('cuda:0')
    torch.cuda.set_stream(torch.cuda.Stream())
    ops.set_state_writeback(False)
    trainer, model, step = workloads.build_trainer(w, w['batch'], dev, 200_000, out_dir='/tmp/eas_aten_sources')
    if len(sys.argv) > 2 and sys.argv[2] == 'eval':
        from spikingjelly.activation_based import functional
        model.eval()
        _, inputs_fn = workloads.device_inputs(w, w['batch'], 200_000, dev, seed=0)

        def run():
            model(inputs_fn()[0])
            functional.reset_net(model)
        ctx = torch.no_grad()
        ctx.__enter__()
        scope = ops.frozen_weights(model)
        scope.__enter__()
    else:
        run = step.eager
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        run()
        torch.cuda.synchronize()
    groups = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.device_time_total <= 0:
            continue
        # only leaf operators that own kernels
        if ev.self_device_time_total <= 0:
            continue
        frame = '?'
        for fr in ev.stack:
            if ROOT in fr or 'eas_snn_amd' in fr or 'bench.py' in fr:
                frame = fr.replace(ROOT + '/', '')
                break
        if frame == '?' and ev.stack:
            frame = 'autograd / ' + ev.stack[0][:80]
        shp = str(ev.input_shapes)[:90] if ev.input_shapes else ''
        key = (ev.name, frame if frame != '?' else shp)
        groups[key][0] += 1
        groups[key][1] += ev.self_device_time_total
    tot = sum(v[1] for v in groups.values())
    print(f'config {config}: ATen operators with device time in one eager step: {sum(v[0] for v in groups.values())} calls, {tot / 1e3:.3f} ms')
    for (name, frame), (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f'{us:8.1f} us {n:4d} x  {name:28s} {frame}')


if __name__ == '__main__':
    main()
