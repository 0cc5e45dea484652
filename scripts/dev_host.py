#!/usr/bin/env python3
"""Host-side cost of enqueueing one eager training step (development tool): wall time of step() without waiting for the GPU,
and a cProfile of where the host spends it."""
import cProfile
import os
import pstats
import sys
import time

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
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print('enqueue ms / total ms per step:', [(round(a, 1), round(b, 1)) for a, b in ts])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
