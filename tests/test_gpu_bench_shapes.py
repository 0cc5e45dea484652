"""Parity at the shapes the bench launches.

Tile, wave-count, LDS-buffering and split-K choices of the convolution kernels are functions of NI = T x batch and of the real per-CU
residency, so the kernel INSTANCES ``bench.py`` runs (batch 64 / 32) are not the ones a B <= 2 test runs.  For every BASELINE
configuration this file

  1. records one training step of the bench's own object (``workloads.build_trainer`` -> ``Trainer.step_fn``) at the bench batch:
     every C-ABI convolution call with its geometry, and -- through the library's kernel-instance trace (``ops.kernel_trace``,
     ``eas_kernel_trace_begin`` / ``_dump``) -- the symbol of every device kernel launched (the names rocprofv3 reports);
  2. replays EVERY distinct convolution geometry of that step -- forward (fp32 one-term / three-term, spike planes, with and without
     bias or the BatchNorm-statistics epilogue), stride-2 input gradient, weight gradient -- on seeded inputs against an fp64
     convolution: 1e-5 of the largest magnitude;
  3. runs the oracle-compared model tests at the configuration's canvas under the trace: every spiking block teacher-forced forward
     (train-mode BatchNorm) and backward against the torch-CPU oracle, the step on spike planes against the step on fp32 spikes
     (bit-identical), for config 2 also the sampler and the real-valued neck + head forward and backward;
  4. asserts that every conv_* / conv1x1_* / *_wgrad_* / bn_* / arsnn_* / smallconv_* / lif / pool / planes instance of (1) was
     launched in (2) or (3).

Reference: yolox/models/network_blocks.py:52-53 (the convolution inside BaseConv), yolox/models/yolo_pafpn.py:88-116, yolo_head.py forward."""
import re
import zlib

import numpy as np
import pytest
import torch

import test_gpu_model as M

pytestmark = pytest.mark.gpu

# kernel families whose every bench instance must have been launched by an oracle-compared run
FAMILIES = ('conv_fwd_mfma', 'conv1x1_mfma', 'conv_dgrad_s2', 'conv_dgrad_small', 'conv_wgrad_mfma', 'conv1x1_wgrad', 'bn_lif', 'bn_silu', 'bn_stats', 'bn_finalize',
            'arsnn_', 'smallconv', 'lif_', 'spp_pool', 'planes_', 'upcat', 'focus', 'time_mean', 'conv3x3_group', 'conv1x1_group',
            'conv_wgrad_group', 'channel_sum')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    import eas_snn_amd
    eas_snn_amd.hip_library()
    return torch.device('cuda:0')


def _short(sym):
    """'void (anonymous namespace)::conv_fwd_mfma_kernel<9, 1, 3, ...>(float const*, ...)' -> 'conv_fwd_mfma_kernel<9, 1, 3, ...>'"""
    s = re.sub(r'^void\s+', '', sym)
    s = s.replace('(anonymous namespace)::', '')
    depth, out = 0, []
    for ch in s:                                    # cut the argument list: the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out).strip()


def _family(name):
    return any(f in name for f in FAMILIES)


# ------------------------------------------------------------------------------------------------ fp64 references
def _conv64(x, w, b, stride, pad):
    """fp64 convolution on the device (ATen's im2col + dgemm path: no library of ours involved), on the host if the device refuses"""
    try:
        return torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad)
    except RuntimeError:
        return torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), stride=stride,
                                          padding=pad).to(x.device)


def _rel(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _inputs(NI, Cin, H, W, spikes, seed, dev):
    g = torch.Generator(device='cpu').manual_seed(seed)
    if spikes:
        # spikes and SEW sums: mostly 0 / 1, a few 2 / 3
        x = (torch.rand(NI, Cin, H, W, generator=g) < 0.2).float() + (torch.rand(NI, Cin, H, W, generator=g) < 0.05).float() * 2
    else:
        x = torch.randn(NI, Cin, H, W, generator=g)
    return x.to(dev)


def _replay_conv_calls(dev, calls):
    """every distinct convolution geometry among ``calls`` against fp64; returns the number of geometries replayed per entry point"""
    from eas_snn_amd import _lib, ops
    L = _lib.lib()
    seen, count = set(), {}
    for name, a in calls:
        if name == 'eas_conv_fwd':
            key = ('fwd',) + tuple(a[4:12]) + (a[2] is not None, False)
        elif name == 'eas_conv_fwd_stats':
            key = ('fwd',) + tuple(a[3:11]) + (False, True)
        elif name == 'eas_conv_fwd_planes':
            key = ('fwd',) + tuple(a[4:11]) + (2, a[2] is not None, a[11] is not None)
        elif name == 'eas_conv_dgrad_s2':
            key = ('dgrad_s2',) + tuple(a[3:8])
        elif name == 'eas_conv_dgrad_small':
            key = ('dgrad_small',) + tuple(a[3:8])
        elif name == 'eas_conv_wgrad_partial':
            key = ('wgrad',) + tuple(a[3:11])
        elif name == 'eas_conv_wgrad_planes_partial':
            key = ('wgrad',) + tuple(a[3:10]) + (2,)
        else:
            continue
        if key in seen:
            continue
        seen.add(key)
        count[key[0]] = count.get(key[0], 0) + 1
        seed = zlib.crc32(repr(key).encode()) % (2 ** 31)
        gw = torch.Generator(device='cpu').manual_seed(seed + 1)
        if key[0] == 'fwd':
            NI, Cin, Cout, H, W, k, s, xt, has_bias, stats = key[1:]
            x = _inputs(NI, Cin, H, W, xt != 3, seed, dev)
            w = (torch.randn(Cout, Cin, k, k, generator=gw) / (Cin * k * k) ** 0.5).to(dev)
            b = torch.randn(Cout, generator=gw).to(dev) if has_bias else None
            pk = ops.conv_pack_weights(w, 0)
            with ops.conv_stats_scope(stats):
                y = ops.conv_fwd_packed(x, pk, b, Cout, k, s, 1 if xt == 2 else xt, ops.to_planes(x) if xt == 2 else None)
            slot = ops._CONV_STATS_SLOT
            ops.clear_conv_stats()
            ref = _conv64(x, w, b, s, k // 2)
            err = _rel(y, ref)
            assert err < 1e-5, f'{key}: forward {err:.2e}'
            if stats:
                assert slot is not None and slot[0] is y, f'{key}: the convolution left no tile sums'
                st = slot[2].view(Cout, slot[1], 2).sum(1)
                y64 = y.double()
                ssum, ssq = y64.sum((0, 2, 3)), (y64 * y64).sum((0, 2, 3))
                assert float(((st[:, 0] - ssum).abs() / (y64.abs().sum((0, 2, 3)) + 1e-30)).max()) < 3e-7, f'{key}: tile sums'
                assert float(((st[:, 1] - ssq).abs() / (ssq + 1e-30)).max()) < 3e-7, f'{key}: tile sums of squares'
        elif key[0] == 'dgrad_small':
            NI, Cin, Cout, H, W = key[1:]
            gy = torch.randn(NI, Cout, H, W, generator=gw).to(dev)
            w = (torch.randn(Cout, Cin, 3, 3, generator=gw) / (Cin * 9) ** 0.5).to(dev)
            gx = torch.full((NI, Cin, H, W), float('nan'), device=dev)
            ops.check(L.eas_conv_dgrad_small(ops.ptr(gy), ops.ptr(w), ops.ptr(gx), NI, Cin, Cout, H, W, ops.stream()), 'eas_conv_dgrad_small')
            ref = torch.nn.grad.conv2d_input((NI, Cin, H, W), w.double(), gy.double(), stride=1, padding=1)
            err = _rel(gx, ref)
            assert err < 1e-5, f'{key}: small-channel input gradient {err:.2e}'
        elif key[0] == 'dgrad_s2':
            NI, Cin, Cout, H, W = key[1:]
            gy = torch.randn(NI, Cout, H // 2, W // 2, generator=gw).to(dev)
            w = (torch.randn(Cout, Cin, 3, 3, generator=gw) / (Cin * 9) ** 0.5).to(dev)
            gx = torch.full((NI, Cin, H, W), float('nan'), device=dev)
            pk = ops.conv_pack_weights(w, 2)
            ops.check(L.eas_conv_dgrad_s2(ops.ptr(gy), ops.ptr(pk), ops.ptr(gx), NI, Cin, Cout, H, W, ops.stream()), 'eas_conv_dgrad_s2')
            ref = torch.nn.grad.conv2d_input((NI, Cin, H, W), w.double(), gy.double(), stride=2, padding=1)
            err = _rel(gx, ref)
            assert err < 1e-5, f'{key}: stride-2 input gradient {err:.2e}'
        else:
            NI, Cin, Cout, H, W, k, s, xt = key[1:]
            x = _inputs(NI, Cin, H, W, xt != 3, seed, dev)
            Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            gy = torch.randn(NI, Cout, Ho, Wo, generator=gw).to(dev)
            got = ops.conv_wgrad(x, gy, k, s, 1 if xt == 2 else xt, ops.to_planes(x) if xt == 2 else None)
            ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), gy.double(), stride=s, padding=k // 2)
            err = _rel(got, ref)
            assert err < 1e-5, f'{key}: weight gradient {err:.2e}'
    torch.cuda.synchronize()
    return count


def _replay_group_calls(dev, calls):
    """every distinct GROUPED launch of the step (eas_conv_fwd_group, eas_conv_wgrad_group_partial, eas_bn_silu_fwd_group / _bwd_group,
    eas_channel_sum_group: the head's pyramid levels as one grid per stage) with the step's own problem tables on seeded inputs against
    fp64: 1e-5 of the largest magnitude (BatchNorm + SiLU: 2e-5, as the single launches)"""
    import torch.nn.functional as F
    from eas_snn_amd import _lib, ops
    from eas_snn_amd import ops_group as G
    L = _lib.lib()
    seen, count = set(), {}
    for name, a in calls:
        if name not in ('eas_conv_fwd_group', 'eas_conv_wgrad_group_partial', 'eas_bn_silu_fwd_group', 'eas_bn_silu_bwd_group', 'eas_channel_sum_group'):
            continue
        arr, n = a[0], a[1]
        if name == 'eas_conv_fwd_group':
            key = (name, a[2]) + tuple((q.NI, q.Cin, q.Cout, q.Hi, q.Wi, bool(q.bias), bool(q.stats), q.accumulate) for q in arr[:n])
        elif name == 'eas_conv_wgrad_group_partial':
            key = (name, a[2]) + tuple((q.NI, q.Cin, q.Cout, q.Hi, q.Wi) for q in arr[:n])
        elif name == 'eas_channel_sum_group':
            key = (name,) + tuple((q.N, q.C, q.HW) for q in arr[:n])
        elif name == 'eas_bn_silu_fwd_group':
            key = (name,) + tuple((q.N, q.C, q.HW, q.y_ctot, q.pending.chunks) for q in arr[:n])
        else:
            key = (name,) + tuple((q.N, q.C, q.HW, q.y_ctot, q.grad_out_ctot) for q in arr[:n])
        if key in seen:
            continue
        seen.add(key)
        count[name] = count.get(name, 0) + 1
        g = torch.Generator(device='cpu').manual_seed(zlib.crc32(repr(key).encode()) % (2 ** 31))
        if name == 'eas_conv_fwd_group':
            k = key[1]
            probs = key[2:]
            xs = [torch.randn(NI, Cin, H, W, generator=g).to(dev) for NI, Cin, Cout, H, W, *_ in probs]
            ws = [(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev) for NI, Cin, Cout, H, W, *_ in probs]
            bs = [torch.randn(p_[2], generator=g).to(dev) if p_[5] else None for p_ in probs]
            pks = [ops.conv_pack_weights(w, 0) for w in ws]
            acc = bool(probs[0][7])
            ys = [torch.randn(p_[0], p_[2], p_[3], p_[4], generator=g).to(dev) for p_ in probs]
            before = [y.clone() for y in ys]
            stats = None
            if probs[0][6]:
                nb = G._conv_plan(tuple(p_[:5] for p_ in probs), k)
                stats = [torch.zeros(p_[2] * b * 2, dtype=torch.float64, device=dev) for p_, b in zip(probs, nb)]
            G._launch_conv_group(xs, pks, bs if any(b is not None for b in bs) else None, ys, stats, k, accumulate=acc)
            for i, (x, w, b, y) in enumerate(zip(xs, ws, bs, ys)):
                ref = _conv64(x, w, b, 1, k // 2)
                if acc:
                    ref = ref + before[i].double()
                assert _rel(y, ref) < 1e-5, f'{key}: problem {i}'
                if stats is not None:
                    st = stats[i].view(w.shape[0], -1, 2).sum(1)
                    y64 = y.double()
                    assert float(((st[:, 0] - y64.sum((0, 2, 3))).abs() / (y64.abs().sum((0, 2, 3)) + 1e-30)).max()) < 3e-7, f'{key}: tile sums'
        elif name == 'eas_conv_wgrad_group_partial':
            k = key[1]
            probs = key[2:]
            xs = [torch.randn(NI, Cin, H, W, generator=g).to(dev) for NI, Cin, Cout, H, W in probs]
            gys = [torch.randn(NI, Cout, H, W, generator=g).to(dev) for NI, Cin, Cout, H, W in probs]
            ns = G._wgrad_plan(probs, k)
            parr = (_lib.EasWgradProblem * len(probs))()
            wss = []
            for q, x, gy, s_ in zip(parr, xs, gys, ns):
                ws = torch.empty(s_ * gy.shape[1] * x.shape[1] * k * k, device=dev)
                wss.append(ws)
                q.x, q.grad_y, q.workspace = ops.ptr(x), ops.ptr(gy), ops.ptr(ws)
                q.NI, q.Cin, q.Cout, q.Hi, q.Wi = x.shape[0], x.shape[1], gy.shape[1], x.shape[2], x.shape[3]
            ops.check(L.eas_conv_wgrad_group_partial(parr, len(probs), k, 3, ops.stream()), name)
            for x, gy, ws, s_ in zip(xs, gys, wss, ns):
                gw = torch.empty((gy.shape[1], x.shape[1], k, k), device=dev)
                ops._wgrad_finish(ws, gw, s_, False)
                ref = torch.nn.grad.conv2d_weight(x.double(), tuple(gw.shape), gy.double(), padding=k // 2)
                assert _rel(gw, ref) < 1e-5, f'{key}'
        elif name == 'eas_channel_sum_group':
            ts = [torch.randn(N, C_, HW, generator=g).to(dev) for N, C_, HW in key[1:]]
            parr = (_lib.EasChannelSumProblem * len(ts))()
            outs = []
            for q, t in zip(parr, ts):
                o = torch.empty(t.shape[1], device=dev)
                outs.append(o)
                q.g, q.out, q.N, q.C, q.HW = ops.ptr(t), ops.ptr(o), t.shape[0], t.shape[1], t.shape[2]
            ops.check(L.eas_channel_sum_group(parr, len(ts), ops.stream()), name)
            for t, o in zip(ts, outs):
                assert _rel(o, t.double().sum((0, 2))) < 1e-5, key
        else:
            # BatchNorm + SiLU forward and backward of the group's layers, each behind a grouped 1x1 convolution that supplies the partial sums
            # (forward and backward tables of a stage describe the same layers: either replays both)
            layers = [(q[0], q[1], q[2], q[3]) for q in key[1:]]
            sig = ('bn', tuple(layers))
            if sig in seen:
                continue
            seen.add(sig)
            ytot = []                    # [(N, HW, Ct), [C of the layers on consecutive channel ranges of one convolution output]] in table order
            for N, C_, HW, yct in layers:
                Ct = yct if yct else C_
                if ytot and ytot[-1][0] == (N, HW, Ct) and sum(ytot[-1][1]) < Ct:
                    ytot[-1][1].append(C_)
                else:
                    ytot.append(((N, HW, Ct), [C_]))
            xs, convs, bns, spec = [], [], [], []
            for (N, HW, Ct), cs in ytot:
                H = 8 if HW % 8 == 0 else 4
                xs.append(torch.randn(N, 16, H, HW // H, generator=g).to(dev))
                convs.append(torch.nn.Conv2d(16, Ct, 1, bias=False).to(dev))
                c0 = 0
                for C_ in cs:
                    bn = torch.nn.BatchNorm2d(C_, eps=1e-3, momentum=0.03).to(dev).train()
                    with torch.no_grad():
                        bn.weight.copy_(torch.rand(C_, generator=g) + 0.5)
                        bn.bias.copy_(torch.randn(C_, generator=g) * 0.3)
                    spec.append((len(xs) - 1, c0, bn))
                    c0 += C_
                assert c0 == Ct, key
            if not G.conv_group_ok(xs, convs, 1):
                continue                                  # (another mix of maps than a 1x1 group takes: the stage's own replay above covers it)
            xr = [x.clone().requires_grad_(True) for x in xs]
            ys, st = G.conv_group(xr, convs, 1, True)
            outs = G.bn_silu_group(ys, st, spec)
            gos = [torch.randn(o.shape, generator=g).to(dev) for o in outs]
            torch.autograd.backward(outs, gos)
            for (yi, c0, bn), o, go in zip(spec, outs, gos):
                y64 = ys[yi].detach().double()[:, c0:c0 + bn.num_features].requires_grad_(True)
                w64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
                ref = F.silu(F.batch_norm(y64, None, None, w64, b64, True, 0.0, 1e-3))
                ref.backward(go.double())
                assert _rel(o.detach(), ref.detach()) < 2e-5, key
                assert _rel(bn.weight.grad, w64.grad) < 2e-5 and _rel(bn.bias.grad, b64.grad) < 2e-5, key
    torch.cuda.synchronize()
    return count


def _replay_bn_calls(dev, calls, T):
    """The BatchNorm kernels whose instance depends on the batch: every distinct BN + SiLU geometry of the step (forward and backward: the
    one-launch block-per-channel backward is chosen from N x HW) against fp64 torch, and every geometry whose batch statistics came from
    the statistics pass (eas_bn_stats_partial: a convolution with more tiles than its epilogue sums) as BN + PLIF against the oracle
    neuron in fp64."""
    import torch.nn.functional as F
    from eas_snn_amd import ops
    from oracle import sj_ref
    from spikingjelly.activation_based import layer, neuron, surrogate
    done = {'bn_silu': 0, 'bn_stats_partial': 0}
    seen = set()
    for name, a in calls:
        if name == 'eas_bn_silu_bwd':
            key = ('bn_silu',) + tuple(a[11:14])
        elif name == 'eas_bn_stats_partial':
            key = ('bn_stats_partial',) + tuple(a[2:6])
        else:
            continue
        if key in seen:
            continue
        seen.add(key)
        done[key[0]] += 1
        g = torch.Generator(device='cpu').manual_seed(zlib.crc32(repr(key).encode()) % (2 ** 31))
        if key[0] == 'bn_silu':
            N, C, HW = key[1:]
            H = 8 if HW % 8 == 0 else 4
            W = HW // H
            bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).to(dev).train()
            with torch.no_grad():
                bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
            y = (torch.randn(N, C, H, W, generator=g) * 1.5 + 0.2).to(dev).requires_grad_(True)
            go = torch.randn(N, C, H, W, generator=g).to(dev)
            out = ops.bn_silu(y, bn)
            out.backward(go)
            y64 = y.detach().double().requires_grad_(True)
            w64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
            ref = F.silu(F.batch_norm(y64, None, None, w64, b64, True, 0.0, 1e-3))
            ref.backward(go.double())
            for what, got, want in (('out', out, ref), ('grad_y', y.grad, y64.grad), ('grad_gamma', bn.weight.grad, w64.grad), ('grad_beta', bn.bias.grad, b64.grad)):
                err = _rel(got.detach(), want.detach())
                assert err < 2e-5, f'{key} {what}: {err:.2e}'
        else:
            TN, C, HW, replicas = key[1:]
            bcast = replicas > 1            # the convolution ran once for T identical frames (the first spiking layer behind the stem)
            N = TN if bcast else TN // T
            assert bcast or N * T == TN
            H = 8 if HW % 8 == 0 else 4
            W = HW // H
            y = (torch.randn(1 if bcast else T, N, C, H, W, generator=g) * 1.5 + 0.3).to(dev)
            go = torch.randn(T, N, C, H, W, generator=g).to(dev)
            gam, bet = (torch.rand(C, generator=g) + 0.6).to(dev), (torch.rand(C, generator=g) * 0.6 - 0.1).to(dev)
            hbn = layer.BatchNorm2d(C, eps=1e-3, momentum=0.03, step_mode='m').to(dev).train()
            hnode = neuron.ParametricLIFNode(init_tau=2.0, decay_input=False, v_reset=None, surrogate_function=surrogate.ATan(2.0), step_mode='m').to(dev)
            rbn = sj_ref.BatchNorm2d(C, eps=1e-3, momentum=0.03, step_mode='m').to(dev).double().train()
            rnode = sj_ref.ParametricLIFNode(init_tau=2.0, decay_input=False, v_reset=None, surrogate_function=sj_ref.ATan(2.0), step_mode='m').to(dev).double()
            with torch.no_grad():
                for bn_ in (hbn, rbn):
                    bn_.weight.copy_(gam); bn_.bias.copy_(bet)
            yh = y.clone().requires_grad_(True)
            if bcast:
                yseq = yh[0].unsqueeze(0).expand(T, N, C, H, W)
                yseq._eas_base = yh[0]
            else:
                yseq = yh
            sh = hbn.fused_with(hnode, yseq)                    # y was made by no convolution of ours: the statistics pass runs
            sh.backward(go)
            yr = y.double().requires_grad_(True)
            sr = rnode(rbn(yr.expand(T, N, C, H, W) if bcast else yr))
            sr.backward(go.double())
            flips = sh.detach().double() != sr.detach()
            assert float(flips.float().mean()) < 2e-5, f'{key}: flip fraction {float(flips.float().mean()):.2e}'
            same = ~flips.any(dim=0, keepdim=True)
            same = same if bcast else same.expand_as(flips)
            nflip = int(flips.any(dim=0).sum())
            d = (yh.grad.double() - yr.grad).abs()[same]
            assert float(d.max()) <= 5e-4 * float(yr.grad.abs().max()) + (1e-3 if nflip else 0.0), f'{key} grad_y: {float(d.max()):.2e}'
            slack = nflip * T * float(go.abs().max()) * 2.0
            for what, got, want in (('grad_gamma', hbn.weight.grad, rbn.weight.grad), ('grad_beta', hbn.bias.grad, rbn.bias.grad), ('grad_w', hnode.w.grad, rnode.w.grad)):
                dd = float((got.double() - want).abs().max())
                assert dd <= 5e-4 * float(want.abs().max()) + slack, f'{key} {what}: {dd:.2e}'
            assert _rel(hbn.running_mean, rbn.running_mean) < 1e-5 and _rel(hbn.running_var, rbn.running_var) < 1e-5
    return done


# ------------------------------------------------------------------------------------------------ the model-level providers
def _planes_vs_fp32_step(dev, exp_name, cfg, shape):
    """one training step of the model at ``shape`` with spike planes and with fp32 spikes: loss, gradients, buffers bit-identical (ties
    the planes-writing / planes-reading kernels to the fp32 kernels the teacher-forced tests compare with the oracle)"""
    from eas_snn_amd import data, ops
    from oracle import fill
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    H, W = shape[-2:]
    exp = get_exp(None, exp_name)
    exp.merge(M._exp_opts(cfg, H, W))
    model = exp.get_model()
    fill.procedural_fill_(model, 2.0, ann_regex=fill.ANN_KEYS[cfg['use_spike']])
    model.to(dev).train()
    model.head.use_l1 = True
    x = torch.from_numpy(fill.poisson_events(shape, 0.5, seed=11)).to(dev)
    tg = data.synth_targets(shape[0], (H, W), dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    prev = ops.SPIKE_PLANES
    try:
        for planes in (True, False):
            ops.SPIKE_PLANES = planes
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            out = model(x, tg)
            out['total_loss'].backward()
            functional.reset_net(model)
            runs.append((out['total_loss'].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()},
                         {n: b.clone() for n, b in model.named_buffers()}))
    finally:
        ops.SPIKE_PLANES = prev
    assert torch.equal(runs[0][0], runs[1][0])
    diff = sorted(n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[1][1][n]))
    assert not diff, f'planes vs fp32 spikes: gradient differences in {diff[:5]}'
    diff = sorted(n for n in runs[0][2] if not torch.equal(runs[0][2][n], runs[1][2][n]))
    assert not diff, f'planes vs fp32 spikes: buffer differences in {diff[:5]}'


def _neck_head_backward(dev, grouped=False):
    """BASELINE configs[1] at 256x320: the real-valued PAFPN neck and the head in train mode, backward.  The oracle runs a whole training
    step; its backbone firing rates go INTO the HIP neck + head, the oracle's gradients at the nine raw prediction maps go into their
    backward, and every neck / head parameter gradient and the three firing-rate gradients are compared elementwise at the ``_grad_close``
    bar.  No neuron behind the cut: no flip excuse."""
    from eas_snn_amd import ops
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    H, W, B = 256, 320, 2
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(M._exp_opts(dict(use_spike='True'), H, W))
    hip = exp.get_model()
    ref = model_ref.build_model(use_spike='True')
    rx = fill.ANN_KEYS['True']
    assert fill.procedural_fill_(hip, 2.0, ann_regex=rx) == fill.procedural_fill_(ref, 2.0, ann_regex=rx)
    hip.to(dev).train(); ref.train()
    ref.head.use_l1 = True
    feats, raws, graws = {}, {}, {}
    hook = ref.backbone.backbone.register_forward_hook(lambda m, i, o: feats.update({k: v.detach() for k, v in o.items()}))
    x = torch.from_numpy(fill.poisson_events((B, 1, 4, 2, H, W), 0.5, seed=21))
    tg = torch.zeros(B, 50, 5)
    tg[:, 0] = torch.tensor([0, 0.3 * W, 0.4 * H, 0.25 * W, 0.3 * H])
    tg[:, 1] = torch.tensor([1, 0.7 * W, 0.6 * H, 0.2 * W, 0.35 * H])
    with torch.no_grad():
        ref(x, tg)
    hook.remove()
    sj_ref.reset_net(ref)
    # the oracle's neck + head on its own backbone outputs as LEAVES (the spiking backbone replaced by a stand-in that hands them out): the
    # gradients that reach the leaves are those through the neck alone (in the full model the stages also feed each other)
    names = ('dark3', 'dark4', 'dark5')
    leaves = {k: feats[k].clone().requires_grad_(True) for k in names}

    class HandOut(torch.nn.Module):
        def forward(self, _x):
            return leaves
    spiking = ref.backbone.backbone
    ref.backbone.backbone = HandOut()
    hooks = []
    for k in range(3):
        for kind in ('reg', 'obj', 'cls'):
            def keep_raw(m, i, o, key=(kind, k)):
                raws[key] = o.detach()
                o.register_hook(lambda g, key=key: graws.__setitem__(key, g.detach()))
            hooks.append(getattr(ref.head, f'{kind}_preds')[k].register_forward_hook(keep_raw))
    ref.zero_grad(set_to_none=True)
    ref(x, tg)['total_loss'].backward()
    for h in hooks:
        h.remove()
    ref.backbone.backbone = spiking
    sj_ref.reset_net(ref)
    assert len(graws) == 9
    # firing rate = mean over T (spiking_yolo_pafpn.py:98): its gradient is the sum over T of the gradients at the T steps
    rates = [feats[f].mean(0).to(dev).requires_grad_(True) for f in names]
    want_rate_grads = [leaves[f].grad.sum(0) for f in names]
    hip.backbone._features = lambda _x: rates
    with ops.packed_weights(hip):
        fpn = hip.backbone(None)
        if grouped:          # every stage of the three levels as one grouped launch (ops_group)
            got_raws = hip.head._levels_grouped([hip.head._prepare(f) for f in fpn])
            assert got_raws is not None
        else:
            got_raws = [hip.head._level(k, hip.head._prepare(f)) for k, f in enumerate(fpn)]
        outs, gouts = [], []
        for k in range(3):
            for j, kind in enumerate(('reg', 'obj', 'cls')):
                np.testing.assert_allclose(got_raws[k][j].detach().cpu().numpy(), raws[(kind, k)].numpy(), rtol=M.RTOL, atol=M.RTOL, err_msg=f'{kind}[{k}]')
                outs.append(got_raws[k][j])
                gouts.append(graws[(kind, k)].to(dev))
        torch.autograd.backward(outs, gouts)
    del hip.backbone._features
    functional.reset_net(hip)
    worst, worst_name, n = 0.0, '', 0
    for r, wg, f in zip(rates, want_rate_grads, names):
        w_ = M._grad_close(r.grad.cpu().numpy(), wg.numpy(), f'firing-rate gradient {f}')
        worst, worst_name = max((worst, worst_name), (w_, f))
    pr = dict(ref.named_parameters())
    for pn, p in hip.named_parameters():
        if pn.startswith('head.') or (pn.startswith('backbone.') and not pn.startswith('backbone.backbone.')):
            assert p.grad is not None and pr[pn].grad is not None, pn
            w_ = M._grad_close(p.grad.cpu().numpy(), pr[pn].grad.numpy(), pn)
            worst, worst_name = max((worst, worst_name), (w_, pn))
            n += 1
    assert n >= 130, n          # 39 conv weights + 78 BatchNorm affine parameters + 9 prediction weights + 9 biases
    print(f'neck + head backward on the oracle firing rates and raw-map gradients: {n} parameter gradients + 3 rate gradients, worst element at '
          f'{worst:.3f} of the tolerance ({worst_name})')


# configuration -> (experiment, oracle configuration, input shape at B = 1, expected spiking blocks)
CANVAS = {
    2: ('e-yolox-s', dict(use_spike='True'), (1, 1, 4, 2, 256, 320), 34),
    3: ('e-yolox-m', dict(M.M_WIDTH, use_spike='full_spike_v2', T=5, Tm=4, num_classes=2), (1, 1, 4, 2, 256, 320), 97),
    4: ('e-yolox-m', dict(M.M_WIDTH, use_spike='full_spike_v2', T=3, Tm=4, num_classes=3), (1, 1, 4, 2, 384, 640), 97),
    5: ('e-yolox-m', dict(M.M_WIDTH, use_spike='full_spike_v2', T=7, Tm=8, Ts=7, num_classes=100, alpha=1.5), (1, 1, 8, 2, 192, 256), 97),
}


def _bench_step_trace(dev, config):
    """one eager training step of the bench's own step object at the bench batch, under the kernel trace"""
    import types
    from eas_snn_amd import ops, workloads
    w = workloads.get(config)
    trainer, model, step = workloads.build_trainer(w, w['batch'], dev, events=200_000)
    with ops.no_state_writeback():
        step.eager()                                  # allocator / lazily built state
        with ops.kernel_trace() as tr:
            step.eager()
    torch.cuda.synchronize()
    assert torch.isfinite(step.loss)
    ops.check_tags(f'config {config} bench step')
    del trainer, model, step
    torch.cuda.empty_cache()
    return tr


@pytest.mark.parametrize('config', [2, 3, 4, 5])
def test_every_kernel_instance_of_the_bench_step_is_oracle_checked(dev, config):
    from eas_snn_amd import ops
    exp_name, cfg, shape, nblocks = CANVAS[config]
    bench = _bench_step_trace(dev, config)
    bench_kernels = sorted({_short(k) for k in bench.kernels})
    assert any('conv_fwd_mfma_kernel' in k for k in bench_kernels) and any('bn_lif' in k for k in bench_kernels)
    tested = set()
    # (2) every convolution geometry of the step at the bench NI against fp64
    with ops.kernel_trace() as tr:
        counts = _replay_conv_calls(dev, bench.calls)
        counts.update(_replay_bn_calls(dev, bench.calls, cfg.get('T', 3)))
        counts.update(_replay_group_calls(dev, bench.calls))
    tested |= {_short(k) for k in tr.kernels}
    print(f'config {config}: replayed against fp64 at the bench batch: {counts}')
    assert counts.get('fwd', 0) >= 20 and counts.get('wgrad', 0) >= 15
    # (3) the oracle-compared model runs at this canvas
    with ops.kernel_trace() as tr:
        st = M._teacher_forced(dev, exp_name, cfg, shape, train=True)
        assert st['layers'] == nblocks
        st = M._teacher_forced_backward(dev, exp_name, cfg, shape, nblocks)
        assert st['layers'] == nblocks
        _planes_vs_fp32_step(dev, exp_name, cfg, (2,) + shape[1:])
        if config == 2:
            M.test_sampler_parity_at_256x320(dev)
            M.test_neck_and_head_teacher_forced_every_logit(dev, True)
            _neck_head_backward(dev)
            _neck_head_backward(dev, grouped=True)
    tested |= {_short(k) for k in tr.kernels}
    missing = [k for k in bench_kernels if _family(k) and k not in tested]
    fams = sorted({k.split('<')[0] for k in bench_kernels if _family(k)})
    print(f'config {config}: {sum(1 for k in bench_kernels if _family(k))} kernel instances in the bench step ({len(fams)} kernels: {fams}); '
          f'{len(tested)} instances launched by the oracle-compared runs')
    assert not missing, f'config {config}: bench kernel instances no oracle-compared run launched:\n  ' + '\n  '.join(missing)
