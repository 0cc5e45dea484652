#!/usr/bin/env python3
"""Feasibility of capturing the whole training step in a HIP graph (development tool)."""
import os
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
    for gr in opt.param_groups:
        gr['capturable'] = True
    ev = data.events_to_device(data.synth_event_batch(64, 200_000, *bench.SENSOR, seed=0), dev)
    targets = data.synth_targets(64, bench.CANVAS, dev)

    def step():
        frames = data.events_to_frames(ev, exp.Tm, bench.SENSOR, bench.CANVAS)
        out = model(frames, targets)
        opt.zero_grad(set_to_none=True)
        out['total_loss'].backward()
        opt.step()
        functional.reset_net(model)
        return out['total_loss']

    def timed(fn, n=10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(4):
            l0 = step()
    torch.cuda.current_stream().wait_stream(s)
    print('eager ms/step', timed(step), 'loss', float(l0))
    g = torch.cuda.CUDAGraph()
    last = ['?']

    def instrument(root):
        seen, stack = set(), [root]
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            fn.register_prehook(lambda gi, n=type(fn).__name__: last.__setitem__(0, n))
            stack.extend(f for f, _ in fn.next_functions)

    def step_dbg():
        frames = data.events_to_frames(ev, exp.Tm, bench.SENSOR, bench.CANVAS)
        out = model(frames, targets)
        opt.zero_grad(set_to_none=True)
        if os.environ.get('EAS_GRAPH_DBG'):
            instrument(out['total_loss'].grad_fn)
        out['total_loss'].backward()
        opt.step()
        functional.reset_net(model)
        return out['total_loss']

    try:
        with torch.cuda.graph(g):
            loss = step_dbg()
    except Exception as e:
        print('capture failed after autograd node', last[0], ':', str(e)[:100])
        raise
    print('captured')
    g.replay()
    torch.cuda.synchronize()
    print('graph ms/step', timed(g.replay), 'loss', float(loss))
    for _ in range(3):
        g.replay()
    print('loss after more replays', float(loss))


if __name__ == '__main__':
    main()
