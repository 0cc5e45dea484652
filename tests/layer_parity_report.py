#!/usr/bin/env python3
"""Layer-wise parity report: HIP model (cuda:0) vs the CPU oracle, same weights and input.

For every module with the same name in both trees the forward output is compared: spike tensors by flip
fraction, real-valued tensors by the fraction of elements within 1e-4 relative.  With --teacher the HIP
layer is fed the ORACLE's input for that layer (isolates per-layer error from cascaded spike flips).
Test infrastructure (imports oracle/); not part of the product."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import eas_snn_amd  # noqa
from oracle import fill, model_ref, sj_ref
from spikingjelly.activation_based import functional
from yolox.exp import get_exp

OPTS = ['T', '3', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
        'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'spike_fn', 'atan']


def first(t):
    return t[0] if isinstance(t, tuple) else t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hw', type=int, nargs=2, default=[256, 320])
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--use-spike', default='True')
    ap.add_argument('--benchmark', action='store_true')
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = a.benchmark
    dev = torch.device('cuda:0')
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(OPTS + ['use_spike', a.use_spike])
    hip = exp.get_model()
    ref = model_ref.build_model(use_spike=a.use_spike)
    assert fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS[a.use_spike]) == \
        fill.procedural_fill_(ref, 2.0, ann_regex=fill.ANN_KEYS[a.use_spike])
    hip.to(dev).eval(); ref.eval()
    x = torch.from_numpy(fill.poisson_events((a.batch, 1, 4, 2, a.hw[0], a.hw[1]), 0.5, seed=3))
    outs_r, outs_h = {}, {}

    def hook(store):
        def mk(name):
            def f(mod, inp, out):
                o = first(out)
                if isinstance(o, torch.Tensor):
                    store[name] = o.detach().float().cpu()
            return f
        return mk
    for n, m in ref.named_modules():
        if n:
            m.register_forward_hook(hook(outs_r)(n))
    for n, m in hip.named_modules():
        if n:
            m.register_forward_hook(hook(outs_h)(n))
    with torch.no_grad():
        lr = ref(x)
        lh = hip(x.to(dev)).cpu()
    sj_ref.reset_net(ref); functional.reset_net(hip)
    print(f'{"module":58s} {"kind":6s} {"metric":>12s}')
    for n in outs_r:
        if n not in outs_h or outs_r[n].shape != outs_h[n].shape:
            continue
        r, h = outs_r[n].numpy(), outs_h[n].numpy()
        is_spike = np.isin(np.unique(r[:2000].ravel() if r.ndim else r), (0.0, 1.0, 2.0)).all() and r.max() <= 2
        if n.endswith('.act') or n.endswith('.bn') or n.count('.') <= 2:
            if is_spike:
                print(f'{n:58s} spike  flips={float((r != h).mean()):.3e}  rate={r.mean():.3f}')
            else:
                close = np.isclose(h, r, rtol=1e-4, atol=1e-5).mean()
                print(f'{n:58s} real   close={close:.6f}  maxabs={np.abs(h - r).max():.3e}')
    close = np.isclose(lh.numpy(), lr.numpy(), rtol=1e-4, atol=1e-4).mean()
    print(f'logits within 1e-4: {close * 100:.3f}%  max abs diff {float((lh - lr).abs().max()):.3e}')


if __name__ == '__main__':
    main()
