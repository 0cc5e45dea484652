"""development: eval-mode forward of a BASELINE configuration at the bench batch -- eager launches and HIP-graph replay, with the fused
conv -> BN -> LIF step (default) and with the two-kernel path (EAS_FUSED_EVAL=0), and the per-entry-point kernel time of one forward."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eas_snn_amd  # noqa: E402
from eas_snn_amd import ops, workloads  # noqa: E402
from spikingjelly.activation_based import functional  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream())
w = workloads.get(cfg)
exp = workloads.build_exp(w)
model = exp.get_model().to(dev).eval()
for n_, m_ in model.named_modules():
    object.__setattr__(m_, '_eas_name', n_)
PER_LAYER = {}
raw, inputs_fn = workloads.device_inputs(w, w['batch'], 200_000, dev)
B = w['batch']


def fwd():
    out = model(inputs_fn()[0])
    functional.reset_net(model)
    return out


def clock(fn, n=10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


MODES = ('auto', 'all', False) if not os.environ.get('EAS_DEV_EVAL_ONLY') else (os.environ['EAS_DEV_EVAL_ONLY'],)      # one mode only: for a profile
for fused in MODES:
    ops.FUSED_EVAL = fused
    with torch.no_grad(), ops.no_state_writeback(), ops.frozen_weights(model):      # (frozen_weights: as EventEvaluator.evaluate runs its loop)
        for _ in range(3):
            fwd()
        eager = clock(fwd)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fwd()
        g.replay()
        replay = clock(g.replay, 20)
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        fwd()
        torch.cuda.synchronize()
        ops.set_timer(None)
    summ = timer.summary()
    tot = sum(v['ms'] for v in summ.values())
    print(f'config {cfg} batch {B} fused={fused}: eager {eager:.3f} ms ({B / eager * 1e3:.0f} frames/s), graph replay {replay:.3f} ms '
          f'({B / replay * 1e3:.0f} frames/s); timed entry points {tot:.3f} ms')
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['ms'])[:8]:
        print(f"    {k:28s} {v['calls']:4d} calls {v['ms']:.3f} ms")
    for (tag, name), items in timer.tagged.items():
        PER_LAYER.setdefault(tag, {})[fused] = PER_LAYER.setdefault(tag, {}).get(fused, 0.0) + sum(a.elapsed_time(b) for a, b, *_ in items)
if os.environ.get('EAS_DEV_EVAL_LAYERS'):
    # the modes of the loop above are the keys: 'auto' (the default selection), 'all' (every eligible layer fused), False (two kernels)
    print('per layer (ms): all fused / two-kernel   (auto)')
    for tag, d in sorted(PER_LAYER.items(), key=lambda kv: -(kv[1].get('all', 0) - kv[1].get(False, 0))):
        if tag and (d.get('all') or d.get(False)):
            m_ = model.get_submodule(tag)
            c_ = getattr(m_, 'conv', None)
            c_ = c_[0] if isinstance(c_, torch.nn.Sequential) else c_
            desc = f'{c_.in_channels}->{c_.out_channels} k{c_.kernel_size[0]} s{c_.stride[0]}' if isinstance(c_, torch.nn.Conv2d) else ''
            print(f"  {d.get('all', 0):.3f}  {d.get(False, 0):.3f}  ({d.get('auto', 0):.3f})   {tag}  {desc}")
