#!/usr/bin/env python3
"""development (GPU box): the grouped (multi-problem) launches of ops_group against the per-problem operators.
  1. grouped convolutions (3x3 pair, 3x3 single, 1x1, ragged-channel input gradient with accumulation) == eas_conv_fwd per problem, bit for bit
  2. the whole head, training mode: loss, input gradients and every parameter gradient, grouped against per-level
usage: dev_group.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn

import eas_snn_amd
from eas_snn_amd import data, ops
from eas_snn_amd import ops_group as G


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def unit(batch):
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    hws = [(32, 40), (16, 20), (8, 10)]
    fails = 0
    # ---- 3x3: pairs (128 -> 256), singles (128 -> 128), 1x1 stems, all with statistics
    for k, cins, cout, dual in ((3, [128] * 3, 256, True), (3, [128] * 3, 128, False), (1, [128, 256, 512], 128, False), (3, [64, 64, 64], 64, False)):
        xs = [torch.randn(batch, ci, h, w, device=dev) for ci, (h, w) in zip(cins, hws)]
        convs = [nn.Conv2d(ci, cout, k, 1, k // 2, bias=False).to(dev) for ci in cins]
        for mode in (0, 1):
            if mode == 0:
                geoms = tuple((x.shape[0], x.shape[1], cout, x.shape[2], x.shape[3]) for x in xs)
                ins = xs
            else:
                ins = [torch.randn(batch, cout, h, w, device=dev) for (h, w) in hws]
                geoms = tuple((x.shape[0], cout, ci, x.shape[2], x.shape[3]) for x, ci in zip(ins, cins))
            nb = G._conv_plan(geoms, k)
            assert nb is not None, (k, cins, cout, mode)
            pks = [ops.conv_pack_weights(c.weight, mode) for c in convs]
            couts = [g[2] for g in geoms]
            ys = [torch.empty((x.shape[0], co, x.shape[2], x.shape[3]), device=dev) for x, co in zip(ins, couts)]
            stats = [torch.zeros(co * b * 2, dtype=torch.float64, device=dev) for co, b in zip(couts, nb)]
            G._launch_conv_group(ins, pks, None, ys, stats, k)
            for x, pk, y, st, co, b in zip(ins, pks, ys, stats, couts, nb):
                ref = ops.conv_fwd_packed(x, pk, None, co, k, 1, 3)
                same = torch.equal(ref, y)
                s = st.view(co, b, 2).sum(1)
                e1 = rel(s[:, 0], ref.double().sum((0, 2, 3)))
                e2 = rel(s[:, 1], (ref.double() ** 2).sum((0, 2, 3)))
                ok = same and e1 < 1e-6 and e2 < 1e-6
                fails += not ok
                print(f'conv k{k} mode{mode} {tuple(x.shape)} -> {co}: bit-identical {same}, stats {e1:.1e} {e2:.1e}, nb {b}', 'ok' if ok else 'FAIL')
    # ---- 1x1 with bias, few output channels; ragged input gradient; accumulation
    xs = [torch.randn(batch, 128, h, w, device=dev) for (h, w) in hws] * 3
    couts = [2] * 3 + [4] * 3 + [1] * 3
    convs = [nn.Conv2d(128, co, 1).to(dev) for co in couts]
    pks = [ops.conv_pack_weights(c.weight, 0) for c in convs]
    ys = [torch.empty((x.shape[0], co, x.shape[2], x.shape[3]), device=dev) for x, co in zip(xs, couts)]
    G._launch_conv_group(xs, pks, [c.bias for c in convs], ys, None, 1)
    for x, c, pk, y in zip(xs, convs, pks, ys):
        ref = ops.conv_fwd_packed(x, pk, c.bias, c.out_channels, 1, 1, 3)
        ok = torch.equal(ref, y)
        fails += not ok
        print(f'pred fwd {tuple(x.shape)} -> {c.out_channels}: bit-identical {ok}')
    gys = [torch.randn_like(y) for y in ys]
    pk1 = [ops.conv_pack_weights(c.weight, 1) for c in convs]
    gx = [torch.empty_like(x) for x in xs[:6]]
    G._launch_conv_group(gys[:6], pk1[:6], None, gx, None, 1)
    first = [g.clone() for g in gx[3:6]]
    G._launch_conv_group(gys[6:], pk1[6:], None, gx[3:6], None, 1, accumulate=True)
    for i in range(6):
        ref = ops.conv_fwd_packed(gys[i], pk1[i], None, 128, 1, 1, 3)
        ok = torch.equal(ref, gx[i] if i < 3 else first[i - 3])
        fails += not ok
        print(f'pred dgrad {i}: bit-identical {ok}')
    for i in range(3):
        ref = first[i] + ops.conv_fwd_packed(gys[6 + i], pk1[6 + i], None, 128, 1, 1, 3)
        ok = torch.equal(ref, gx[3 + i])
        fails += not ok
        print(f'pred dgrad accumulate {i}: bit-identical {ok}')
    return fails


def head(batch):
    from yolox.models.yolo_head import YOLOXHead
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    hd = YOLOXHead(2, width=0.5).to(dev)
    hd.initialize_biases(1e-2)
    hd.train()
    hd.use_l1 = True
    for m in hd.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    xs0 = [torch.randn(batch, c, h, w, device=dev) for c, (h, w) in zip((128, 256, 512), ((32, 40), (16, 20), (8, 10)))]
    labels = data.synth_targets(batch, (256, 320), dev)
    res = {}
    state = {k: v.clone() for k, v in hd.state_dict().items()}
    for flag in (False, True):
        G.ENABLED = flag
        hd.load_state_dict(state)
        hd.zero_grad(set_to_none=True)
        xs = [x.clone().requires_grad_(True) for x in xs0]
        with ops.packed_weights(hd):
            out = hd(xs, labels)
        loss = out[0]
        loss.backward()
        torch.cuda.synchronize()
        res[flag] = (float(loss), [x.grad.clone() for x in xs], {n: p.grad.clone() for n, p in hd.named_parameters()},
                     {n: b.clone() for n, b in hd.named_buffers()})
    fails = 0
    la, lb = res[False][0], res[True][0]
    print(f'loss per-level {la:.6f} grouped {lb:.6f}')
    fails += abs(la - lb) > 1e-4 * abs(la)
    for i, (a, b) in enumerate(zip(res[False][1], res[True][1])):
        e = rel(b, a)
        fails += e > 2e-4
        print(f'input gradient level {i}: {e:.2e}')
    worst = 0.0
    for n in res[False][2]:
        e = rel(res[True][2][n], res[False][2][n])
        worst = max(worst, e)
        if e > 5e-4:
            fails += 1
            print(f'  gradient {n}: {e:.2e} FAIL')
    print(f'parameter gradients: worst relative difference {worst:.2e}')
    for n in res[False][3]:
        a, b = res[False][3][n].float(), res[True][3][n].float()
        e = rel(b, a)
        if e > 1e-5:
            fails += 1
            print(f'  buffer {n}: {e:.2e} FAIL')
    # timing: head forward + backward, per-level against grouped
    for flag in (False, True):
        G.ENABLED = flag
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5):
                hd.zero_grad(set_to_none=True)
                xs = [x.clone().requires_grad_(True) for x in xs0]
                with ops.packed_weights(hd), ops.deferred_wgrad_reductions():
                    out = hd(xs, labels)
                    out[0].backward()
            t1.record()
            torch.cuda.synchronize()
        print(f'head fwd + bwd, eager, grouped={flag}: {t0.elapsed_time(t1) / 5:.3f} ms')
    return fails


if __name__ == '__main__':
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.cuda.set_stream(torch.cuda.Stream())
    f = unit(batch)
    f += head(batch)
    print('FAILURES:', f)
    sys.exit(1 if f else 0)
