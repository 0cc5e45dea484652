#!/usr/bin/env python3
"""Which operators receive non-contiguous tensors (and therefore pay a .contiguous() copy) during one training step."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
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
    opt = exp.get_optimizer(64)
    ev = data.events_to_device(data.synth_event_batch(64, 200_000, *bench.SENSOR, seed=0), dev)
    targets = data.synth_targets(64, bench.CANVAS, dev)
    stats = collections.defaultdict(lambda: [0, 0])
    orig = ops._f32c

    def spy(t):
        if t is not None and not t.is_contiguous():
            fr = traceback.extract_stack(limit=3)[0]
            key = f'{fr.name}:{fr.lineno} shape={tuple(t.shape)} strides={tuple(t.stride())}'
            stats[key][0] += 1
            stats[key][1] += t.numel() * 4
        return orig(t)
    ops._f32c = spy

    def step():
        frames = data.events_to_frames(ev, exp.Tm, bench.SENSOR, bench.CANVAS)
        out = model(frames, targets)
        opt.zero_grad(set_to_none=True)
        out['total_loss'].backward()
        opt.step()
        functional.reset_net(model)
    step()
    stats.clear()
    step()
    torch.cuda.synchronize()
    tot = sum(v[1] for v in stats.values())
    print(f'non-contiguous inputs copied per step: {sum(v[0] for v in stats.values())} tensors, {tot / 2**20:.0f} MiB')
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f'{v[0]:3d} x {v[1] / v[0] / 2**20:7.1f} MiB  {k}')


if __name__ == '__main__':
    main()
