#!/usr/bin/env python3
"""Steady-state per-kernel breakdown of the bench.py training step (development tool).

rocprofv3 over the whole process also counts MIOpen's find-mode trial kernels of the first steps; this script
warms up first and then records only N steady steps with torch.profiler (roctracer), printing GPU time per
kernel name per step, grouped into families.  Usage: python scripts/step_profile.py [steps] [batch]"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device('cuda:0')
    import eas_snn_amd
    from eas_snn_amd import data, ops
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    eas_snn_amd.hip_library()
    ops.set_state_writeback(False)
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(bench.OPTS)
    torch.manual_seed(80)
    model = exp.get_model().to(dev)
    model.head.use_l1 = True
    opt = exp.get_optimizer(batch)
    ev = data.events_to_device(data.synth_event_batch(batch, 200_000, *bench.SENSOR, seed=0), dev)
    targets = data.synth_targets(batch, bench.CANVAS, dev)

    def step():
        frames = data.events_to_frames(ev, exp.Tm, bench.SENSOR, bench.CANVAS)
        out = model(frames, targets)
        opt.zero_grad(set_to_none=True)
        out['total_loss'].backward()
        opt.step()
        functional.reset_net(model)

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    by_ops = bool(os.environ.get('EAS_PROFILE_OPS'))
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=by_ops) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            a = agg[e.name]
            a[0] += 1
            a[1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    print(f'total GPU kernel time per step: {total / steps / 1e3:.3f} ms  ({len(rows)} distinct kernels)')
    fam = collections.defaultdict(float)
    for name, (n, us) in rows:
        own = re.search(r'\(anonymous namespace\)::(\w+)', name)
        key = ('eas:' + own.group(1)) if own and not name.startswith('void at::') and 'ck::' not in name else (
            'winograd' if 'Sp3AsmConv' in name else 'igemm_wrw' if 'igemm_wrw' in name else 'igemm_bwd' if 'igemm_bwd' in name else
            'igemm_fwd' if 'igemm_fwd' in name else 'ck_conv' if 'ck::' in name or name.startswith('_ZN2ck') else
            'gemm(Cijk)' if name.startswith('Cijk') else 'transpose' if 'transpose' in name else
            'miopen_bn' if 'BatchNorm' in name else 'im2col' if 'Im2d2Col' in name or 'Col2Im' in name else
            'aten_elementwise' if 'elementwise' in name else 'aten_other')
        fam[key] += us
    print('--- families (ms/step)')
    for k, us in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f'{us / steps / 1e3:9.3f}  {k}')
    print('--- top kernels (calls/step, ms/step, name)')
    for name, (n, us) in rows[:int(os.environ.get("EAS_PROFILE_ROWS", "70"))]:
        print(f'{n / steps:8.1f} {us / steps / 1e3:9.3f}  {name[:150]}')

    if by_ops:   # ATen glue by operator and input shapes (which tensors the remaining non-HIP time is spent on)
        print('--- aten ops by input shape (calls/step, self device ms/step, op, shapes)')
        ka = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.self_device_time_total > 0]
        for e in sorted(ka, key=lambda e: -e.self_device_time_total)[:60]:
            print(f'{e.count / steps:8.1f} {e.self_device_time_total / steps / 1e3:9.3f}  {e.key:28s} {str(e.input_shapes)[:150]}')


if __name__ == '__main__':
    main()
