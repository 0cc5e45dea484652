"""Grouped (multi-problem) launches for layers that are independent of each other -- the three pyramid levels of the detection head
(reference: yolox/models/yolo_head.py:149-200, the per-level loop; spiking_yolo_head.py:149-200).

Each function here takes LISTS of tensors / modules, one entry per problem, and issues ONE launch per kernel family for all of them
(include/eas_hip.h "Grouped (multi-problem) launches"): the convolution (forward, input gradient, weight-gradient slabs), BatchNorm +
SiLU (forward; backward = sums + apply) and the bias-gradient sums.  Per problem the arithmetic is that of the single-problem operators
in ``ops`` (``conv2d`` / ``conv2d_dual`` / ``bn_silu`` / ``bn_silu_pair`` / ``channel_sum``); what changes is who shares a grid.
Training path of the real-valued (ANN) head; everything else keeps the per-layer operators."""
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import check, ptr, stream

ENABLED = os.environ.get('EAS_HEAD_GROUP', '1') == '1'      # development switch: 0 = the per-level operators
_PLANS = {}          # (kind, ksize, geometry tuple) -> per-problem plan values, or None (unsupported: run the problems one by one)


class ShapeProbe:
    """stands for a real-valued fp32 CUDA tensor of a given shape in the ``*_ok`` checks (a stage's input before it exists)"""
    is_cuda, dtype = True, torch.float32

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)

    def dim(self):
        return len(self.shape)


def _conv_plan(geoms, k):
    """statistics partials per channel of every problem of a grouped convolution, or None when the group has no common tile"""
    key = ('conv', k, geoms)
    if key not in _PLANS:
        n = len(geoms)
        arr = (_lib.EasConvProblem * n)()
        for q, (NI, Cin, Cout, H, W) in zip(arr, geoms):
            q.NI, q.Cin, q.Cout, q.Hi, q.Wi = NI, Cin, Cout, H, W
        nb = (C.c_int * n)()
        rc = _lib.lib().eas_conv_fwd_group_plan(arr, n, k, 3, nb)
        _PLANS[key] = tuple(int(v) for v in nb) if rc == 0 else None
    return _PLANS[key]


def _wgrad_plan(geoms, k):
    """slab counts of a grouped weight gradient, or None"""
    key = ('wgrad', k, geoms)
    if key not in _PLANS:
        n = len(geoms)
        arr = (_lib.EasWgradProblem * n)()
        for q, (NI, Cin, Cout, H, W) in zip(arr, geoms):
            q.NI, q.Cin, q.Cout, q.Hi, q.Wi = NI, Cin, Cout, H, W
        ns = (C.c_int * n)()
        rc = _lib.lib().eas_conv_wgrad_group_plan(arr, n, k, 3, ns)
        _PLANS[key] = tuple(int(v) for v in ns) if rc == 0 else None
    return _PLANS[key]


def _launch_conv_group(xs, pks, biases, ys, stats, k, accumulate=False):
    n = len(xs)
    arr = (_lib.EasConvProblem * n)()
    nbytes, fl = 0, 0.0
    for i, q in enumerate(arr):
        x, y = xs[i], ys[i]
        q.x, q.packed_w, q.bias, q.y = ptr(x), ptr(pks[i]), ptr(biases[i]) if biases else None, ptr(y)
        q.stats = ptr(stats[i]) if stats else None
        q.NI, q.Cin, q.Cout, q.Hi, q.Wi = x.shape[0], x.shape[1], y.shape[1], x.shape[2], x.shape[3]
        q.accumulate = 1 if accumulate else 0
        nbytes += 4 * (x.numel() + y.numel())
        fl += 2.0 * y.numel() * x.shape[1] * k * k
    ops._call('eas_conv_fwd', nbytes, _lib.lib().eas_conv_fwd_group, arr, n, k, 3, stream(), flops=fl, issue_flops=6 * fl)


def _pack_of(conv, mode):
    """the packing of ``conv.weight`` made by ``ops.packed_weights`` for the running forward, or a fresh one"""
    packs = getattr(conv, '_eas_packs', None)
    if packs is not None and ops._PACK_SCOPE is not None and packs.get('gen') == ops._PACK_SCOPE and mode in packs:
        return packs
    return None


class _ConvGroupFn(torch.autograd.Function):
    """n convolutions (stride 1, same kernel size, real-valued inputs) as one launch each for forward, input gradients, weight-gradient
    slabs and bias gradients.  tensors = x_0..x_{n-1}, w_0.., [wb_0.. (dual: the convolution of problem i has the concatenated weight
    [w_i ; wb_i])], [bias_0..].  Returns y_0.., then (want_stats) the BatchNorm partial sums of every y (not differentiable)."""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        ctx.set_materialize_grads(False)      # outputs nobody differentiates (the statistics handed to the BatchNorm launch) arrive as None, not as zero-filled tensors
        n, k, dual, has_bias, want_stats = cfg['n'], cfg['k'], cfg['dual'], cfg['has_bias'], cfg['want_stats']
        xs = [ops._f32c(t) for t in tensors[:n]]
        was = tensors[n:2 * n]
        wbs = tensors[2 * n:3 * n] if dual else None
        biases = tensors[(3 if dual else 2) * n:(4 if dual else 3) * n] if has_bias else None
        packs = cfg['packs']
        couts = [was[i].shape[0] + (wbs[i].shape[0] if dual else 0) for i in range(n)]
        pks = []
        for i in range(n):
            if packs[i] is not None and 0 in packs[i]:
                pks.append(packs[i][0])
            else:
                pks.append(ops.conv_pack_weights(torch.cat([was[i], wbs[i]], 0) if dual else was[i], 0))
        ys = [torch.empty((x.shape[0], co, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device) for x, co in zip(xs, couts)]
        stats = None
        if want_stats:
            nb = _conv_plan(tuple((x.shape[0], x.shape[1], co, x.shape[2], x.shape[3]) for x, co in zip(xs, couts)), k)
            stats = [torch.empty(co * b * 2, dtype=torch.float64, device=xs[0].device) for co, b in zip(couts, nb)]
        _launch_conv_group(xs, pks, biases, ys, stats, k)
        ctx.save_for_backward(*xs, *was, *(wbs or ()))
        ctx.cfg = cfg
        ctx.tag = ops._TAG
        if want_stats:
            ctx.mark_non_differentiable(*stats)
            return tuple(ys) + tuple(stats)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        n, k, dual, has_bias = cfg['n'], cfg['k'], cfg['dual'], cfg['has_bias']
        saved = ctx.saved_tensors
        xs, was = saved[:n], saved[n:2 * n]
        wbs = saved[2 * n:3 * n] if dual else None
        packs = cfg['packs']
        if ops._TIMER is not None and ctx.tag is not None:
            ops.set_tag(ctx.tag)
        gys = [ops._f32c(g) if g is not None else torch.zeros((x.shape[0], was[i].shape[0] + (wbs[i].shape[0] if dual else 0)) + tuple(x.shape[2:]),
                                                               dtype=torch.float32, device=x.device)
               for i, (g, x) in enumerate(zip(grads[:n], xs))]
        nw = (3 if dual else 2) + (1 if has_bias else 0)
        out = [None] * (nw * n)
        # ---- input gradients: the same convolution on grad_y with the weights packed transposed + flipped
        need_x = [i for i in range(n) if ctx.needs_input_grad[1 + i]]
        if need_x:
            pks = []
            for i in need_x:
                if packs[i] is not None and 1 in packs[i]:
                    pks.append(packs[i][1])
                else:
                    pks.append(ops.conv_pack_weights(torch.cat([was[i], wbs[i]], 0) if dual else was[i], 1))
            gxs = [torch.empty_like(xs[i]) for i in need_x]
            _launch_conv_group([gys[i] for i in need_x], pks, None, gxs, None, k)
            for i, gx in zip(need_x, gxs):
                out[i] = gx
        # ---- weight gradients: the slab kernels of all problems in one grid, the fixed-order reductions with the step's other weight
        # gradients at the end of the backward pass (ops._wgrad_finish)
        need_w = [i for i in range(n) if ctx.needs_input_grad[1 + n + i] or (dual and ctx.needs_input_grad[1 + 2 * n + i])]
        if need_w:
            geoms = tuple((xs[i].shape[0], xs[i].shape[1], gys[i].shape[1], xs[i].shape[2], xs[i].shape[3]) for i in need_w)
            ns = _wgrad_plan(geoms, k)
            gws = []
            if ns is not None:
                m = len(need_w)
                arr = (_lib.EasWgradProblem * m)()
                wss = []
                nbytes, fl = 0, 0.0
                for q, i, s_ in zip(arr, need_w, ns):
                    x, gy = xs[i], gys[i]
                    ws = torch.empty(s_ * gy.shape[1] * x.shape[1] * k * k, dtype=torch.float32, device=x.device)
                    wss.append(ws)
                    q.x, q.grad_y, q.workspace = ptr(x), ptr(gy), ptr(ws)
                    q.NI, q.Cin, q.Cout, q.Hi, q.Wi = x.shape[0], x.shape[1], gy.shape[1], x.shape[2], x.shape[3]
                    nbytes += 4 * (x.numel() + gy.numel())
                    fl += 2.0 * gy.numel() * x.shape[1] * k * k
                defers = [(ops._can_defer(was[i]) and (not dual or (ctx.needs_input_grad[1 + n + i] and ctx.needs_input_grad[1 + 2 * n + i]
                                                                    and ops._can_defer(wbs[i])))) for i in need_w]

                def job(arr=arr, m=m, nbytes=nbytes, fl=fl):
                    ops._call('eas_conv_wgrad', nbytes, _lib.lib().eas_conv_wgrad_group_partial, arr, m, k, 3, stream(), flops=fl, issue_flops=6 * fl)
                ops._wgrad_launch(job, [xs[i] for i in need_w] + [gys[i] for i in need_w] + wss, all(defers), 6 * fl, nbytes)
                for i, ws, s_, df in zip(need_w, wss, ns, defers):
                    gw = torch.empty((gys[i].shape[1], xs[i].shape[1], k, k), dtype=torch.float32, device=ws.device)
                    if dual:
                        both = ctx.needs_input_grad[1 + n + i] and ctx.needs_input_grad[1 + 2 * n + i]
                        ops._wgrad_finish(ws, gw, s_, df, (was[i], wbs[i]) if both else None, was[i].shape[0] if both else None)
                    else:
                        ops._wgrad_finish(ws, gw, s_, df, was[i])
                    gws.append(gw)
            else:       # a subset of the problems (some weights frozen) the group plan does not take: one by one
                for i in need_w:
                    if dual:
                        both = ctx.needs_input_grad[1 + n + i] and ctx.needs_input_grad[1 + 2 * n + i]
                        gws.append(ops.conv_wgrad(xs[i], gys[i], k, 1, 3, None, defer=both and ops._can_defer(was[i]) and ops._can_defer(wbs[i]),
                                                  w=(was[i], wbs[i]) if both else None, split=was[i].shape[0] if both else None))
                    else:
                        gws.append(ops.conv_wgrad(xs[i], gys[i], k, 1, 3, None, defer=ops._can_defer(was[i]), w=was[i]))
            for i, gw in zip(need_w, gws):
                if dual:
                    ca = was[i].shape[0]
                    out[n + i], out[2 * n + i] = gw[:ca], gw[ca:]
                else:
                    out[n + i] = gw
        # ---- bias gradients
        if has_bias:
            b0 = (3 if dual else 2) * n
            need_b = [i for i in range(n) if ctx.needs_input_grad[1 + b0 + i]]
            if need_b:
                m = len(need_b)
                arr = (_lib.EasChannelSumProblem * m)()
                gbs = []
                for q, i in zip(arr, need_b):
                    gy = gys[i]
                    gb = torch.empty(gy.shape[1], dtype=torch.float32, device=gy.device)
                    gbs.append(gb)
                    q.g, q.out, q.N, q.C, q.HW = ptr(gy), ptr(gb), gy.shape[0], gy.shape[1], gy.shape[2] * gy.shape[3]
                check(_lib.lib().eas_channel_sum_group(arr, m, stream()), 'eas_channel_sum_group')
                for i, gb in zip(need_b, gbs):
                    out[b0 + i] = gb
        return (None,) + tuple(out)


def conv_group_ok(xs, items, k):
    """the convolutions ``items`` (one nn.Conv2d per problem, or a pair (conv_a, conv_b) that reads the same input and runs as one) can run
    as a grouped launch on the inputs ``xs``: plain stride-1 k x k convolutions on real-valued 4-D fp32 CUDA tensors without hooks, with a
    common tile plan for the forward and for the input gradient"""
    if not ENABLED or ops.conv_sink() is not None or torch.nn.modules.module._global_forward_hooks:
        return False
    fwd, bwd = [], []
    for x, it in zip(xs, items):
        convs = it if isinstance(it, tuple) else (it,)
        if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32) or ops.is_small_int(x) or ops.planes_of(x) is not None:
            return False
        for c in convs:
            if (not ops._static_conv_ok(c) or c.kernel_size != (k, k) or c.stride != (1, 1) or c._forward_hooks or c._forward_pre_hooks
                    or c.in_channels != x.shape[1] or (c.bias is None) != (convs[0].bias is None)):
                return False
        if len(convs) == 2 and convs[0].bias is not None:
            return False
        cout = sum(c.out_channels for c in convs)
        if k == 3 and cout % 8 != 0:
            return False
        fwd.append((x.shape[0], x.shape[1], cout, x.shape[2], x.shape[3]))
        bwd.append((x.shape[0], cout, x.shape[1], x.shape[2], x.shape[3]))
    if len({(isinstance(it, tuple), (it[0] if isinstance(it, tuple) else it).bias is None) for it in items}) != 1:
        return False
    # (the weight-gradient slabs too: a map the matrix-core weight gradient does not take -- the 2x3 level of a 64x96 test canvas -- keeps the
    # per-level operators, which know the library fallback)
    return _conv_plan(tuple(fwd), k) is not None and _conv_plan(tuple(bwd), k) is not None and _wgrad_plan(tuple(fwd), k) is not None


def conv_group(xs, items, k, want_stats, owners=None):
    """[conv_i(x_i)] as one launch (caller checked ``conv_group_ok``).  items[i]: an nn.Conv2d, or (conv_a, conv_b) = the two convolutions
    that read x_i computed as one (their outputs concatenated along the channels); owners[i] = (module, key) where ``ops.packed_weights`` left
    the packing of such a pair.  Returns (ys, stats): stats[i] = the BatchNorm partial sums of y_i for ``bn_silu_group`` (None without
    ``want_stats``)."""
    n = len(xs)
    dual = isinstance(items[0], tuple)
    has_bias = (items[0][0] if dual else items[0]).bias is not None
    packs = []
    for i, it in enumerate(items):
        if dual:
            d = None
            if owners is not None and owners[i] is not None:
                d = (getattr(owners[i][0], '_eas_dual_packs', None) or {}).get(owners[i][1])
                if d is not None and (ops._PACK_SCOPE is None or d.get('gen') != ops._PACK_SCOPE):
                    d = None
            packs.append(d)
        else:
            packs.append(_pack_of(it, 0))
    tensors = list(xs) + [(it[0] if dual else it).weight for it in items]
    if dual:
        tensors += [it[1].weight for it in items]
    if has_bias:
        tensors += [it.bias for it in items]
    cfg = dict(n=n, k=k, dual=dual, has_bias=has_bias, want_stats=bool(want_stats), packs=packs)
    res = _ConvGroupFn.apply(cfg, *tensors)
    return list(res[:n]), (list(res[n:]) if want_stats else None)


class _BNSiLUGroupFn(torch.autograd.Function):
    """m BatchNorm2d + SiLU layers in one launch; layer j normalises channels [c0_j, c0_j + C_j) of convolution output ys[yi_j] (two layers
    on the halves of one ``conv_group`` output of a pair).  tensors = ys.., stats.., gamma_0.., beta_0...  Backward: both passes of all
    layers as two launches, each layer writing its channel slice of grad_y."""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        ctx.set_materialize_grads(False)      # outputs nobody differentiates (the statistics handed to the BatchNorm launch) arrive as None, not as zero-filled tensors
        ny, items = cfg['ny'], cfg['items']
        m = len(items)
        ys = [ops._f32c(t) for t in tensors[:ny]]
        stats = tensors[ny:2 * ny]
        gammas, betas = tensors[2 * ny:2 * ny + m], tensors[2 * ny + m:2 * ny + 2 * m]
        arr = (_lib.EasBnSiluFwdProblem * m)()
        outs, saved = [], []
        nbytes = 0
        for j, (q, (yi, c0, state)) in enumerate(zip(arr, items)):
            running_mean, running_var, use_batch_stats, momentum, eps, replicas = state
            y = ys[yi]
            N, Ct, H, W = y.shape
            Cc, HW = gammas[j].shape[0], H * W
            dev = y.device
            assert use_batch_stats, 'grouped BatchNorm + SiLU: training-mode statistics (the eval path keeps the fused per-layer kernels)'
            mean = torch.empty(Cc, dtype=torch.float32, device=dev)
            invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
            nb = stats[yi].numel() // (2 * Ct)
            out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
            q.y = y.data_ptr() + 4 * c0 * HW
            q.mean, q.invstd, q.gamma, q.beta, q.out = ptr(mean), ptr(invstd), ptr(gammas[j]), ptr(betas[j]), ptr(out)
            q.N, q.C, q.HW, q.out_ctot, q.y_ctot = N, Cc, HW, 0, Ct
            q.pending = _lib.EasBnPending(stats[yi].data_ptr() + 16 * c0 * nb, nb, int(replicas), float(N) * HW, float(eps),
                                          float(momentum if momentum is not None else 0.0), ptr(running_mean) if momentum is not None else None,
                                          ptr(running_var) if momentum is not None else None, nb)
            outs.append(out)
            saved += [mean, invstd]
            nbytes += 8 * N * Cc * HW
        ops._call('eas_bn_silu_fwd', nbytes, _lib.lib().eas_bn_silu_fwd_group, arr, m, stream())
        ctx.save_for_backward(*ys, *gammas, *betas, *saved)
        ctx.cfg = cfg
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        cfg = ctx.cfg
        ny, items = cfg['ny'], cfg['items']
        m = len(items)
        saved = ctx.saved_tensors
        ys, gammas, betas, ms = saved[:ny], saved[ny:ny + m], saved[ny + m:ny + 2 * m], saved[ny + 2 * m:]
        L = _lib.lib()
        gys = [torch.empty_like(y) for y in ys]
        arr = (_lib.EasBnSiluBwdProblem * m)()
        keep, res_g, res_b = [], [], []
        nbytes = 0
        for j, (q, (yi, c0, state)) in enumerate(zip(arr, items)):
            y = ys[yi]
            N, Ct, H, W = y.shape
            Cc, HW = gammas[j].shape[0], H * W
            g = gs[j]
            if g is None:
                g = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=y.device)
            ctot = ops._channel_slice_of4(g, Cc) if (g.dim() == 4 and g.dtype == torch.float32) else 0
            if ctot == 0:
                g = ops._f32c(g)
            ggamma, gbeta = torch.empty_like(gammas[j]), torch.empty_like(betas[j])
            ws = torch.empty(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=y.device)
            keep += [g, ws]
            off = 4 * c0 * HW
            q.grad_out, q.y, q.mean, q.invstd, q.gamma, q.beta = ptr(g), y.data_ptr() + off, ptr(ms[2 * j]), ptr(ms[2 * j + 1]), ptr(gammas[j]), ptr(betas[j])
            q.grad_y, q.grad_gamma, q.grad_beta, q.workspace = gys[yi].data_ptr() + off, ptr(ggamma), ptr(gbeta), ptr(ws)
            q.batch_stats, q.N, q.C, q.HW, q.grad_out_ctot, q.y_ctot = 1, N, Cc, HW, ctot, Ct
            res_g.append(ggamma)
            res_b.append(gbeta)
            nbytes += 12 * N * Cc * HW
        ops._call('eas_bn_silu_bwd', nbytes, L.eas_bn_silu_bwd_group, arr, m, stream())
        del keep
        return (None,) + tuple(gys) + (None,) * ny + tuple(res_g) + tuple(res_b)


def bn_silu_group_ok(ys, bns):
    return ENABLED and all(type(bn) is torch.nn.BatchNorm2d and bn.affine and bn.training and bn.track_running_stats and bn.momentum is not None
                           and not bn._forward_hooks and not bn._forward_pre_hooks for bn in bns) and all(ops.bn_silu_supported(y) for y in ys)


def bn_silu_group(ys, stats, layers):
    """[silu(bn_j(ys[yi_j][:, c0_j : c0_j + C_j]))] in one launch.  layers[j] = (yi, c0, bn); stats = ``conv_group``'s partial sums of ys."""
    items = tuple((int(yi), int(c0), ops._bn_state(bn)) for yi, c0, bn in layers)
    cfg = dict(ny=len(ys), items=items)
    tensors = list(ys) + list(stats) + [bn.weight for _, _, bn in layers] + [bn.bias for _, _, bn in layers]
    return list(_BNSiLUGroupFn.apply(cfg, *tensors))


# EAS_PRED_DGRAD=0: the prediction convolutions' input gradients as two grouped matrix-core launches (before eas_pred_dgrad_group)
PRED_DGRAD_FMA = os.environ.get('EAS_PRED_DGRAD', '1') != '0'


class _PredGroupFn(torch.autograd.Function):
    """The 1x1 prediction convolutions of all levels (cls_preds[k](cls_feat_k), reg_preds[k](reg_feat_k), obj_preds[k](reg_feat_k);
    yolo_head.py:161-163 of the reference) as ONE forward launch.  Backward: the input gradients as two launches -- (cls, reg) of all
    levels, then obj ADDED to reg's (reg_feat has two readers: no separate addition) --, the weight-gradient slabs as one launch, the bias
    gradients as one launch.  tensors = cls_feat_0.., reg_feat_0.., then (weight, bias) of cls_0.., reg_0.., obj_0..; returns cls_0.., reg_0.., obj_0.."""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        ctx.set_materialize_grads(False)      # outputs nobody differentiates (the statistics handed to the BatchNorm launch) arrive as None, not as zero-filled tensors
        n = cfg['n']
        cf = [ops._f32c(t) for t in tensors[:n]]
        rf = [ops._f32c(t) for t in tensors[n:2 * n]]
        ws = tensors[2 * n:5 * n]
        bs = tensors[5 * n:8 * n]
        xs = cf + rf + rf
        packs = cfg['packs']
        pks = [packs[i][0] if packs[i] is not None and 0 in packs[i] else ops.conv_pack_weights(ws[i], 0) for i in range(3 * n)]
        ys = [torch.empty((x.shape[0], w.shape[0], x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device) for x, w in zip(xs, ws)]
        _launch_conv_group(xs, pks, bs, ys, None, 1)
        ctx.save_for_backward(*cf, *rf, *ws)
        ctx.cfg = cfg
        return tuple(ys)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        n = cfg['n']
        saved = ctx.saved_tensors
        cf, rf, ws = saved[:n], saved[n:2 * n], saved[2 * n:5 * n]
        xs = list(cf) + list(rf) + list(rf)
        packs = cfg['packs']
        gys = [ops._f32c(g) if g is not None else torch.zeros((x.shape[0], w.shape[0]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
               for g, x, w in zip(grads, xs, ws)]
        gcf = [torch.empty_like(t) for t in cf]
        grf = [torch.empty_like(t) for t in rf]
        if PRED_DGRAD_FMA and 2 * n <= 8 and all(w.dtype == torch.float32 and w.is_contiguous() for w in ws) and all(
                (t.shape[2] * t.shape[3]) % 4 == 0 for t in xs) and all(ws[n + i].shape[0] + ws[2 * n + i].shape[0] <= 8 and ws[i].shape[0] <= 8
                                                                         for i in range(n)):
            # 1 / 4 / num_classes reduction channels: store-bound fp32 FMAs instead of matrix-core tiles with K padded to 16; obj joins reg
            arr = (_lib.EasPredDgradProblem * (2 * n))()
            nbytes = 0
            for i in range(n):
                for q, gy_a, w_a, gy_b, w_b, gx in ((arr[i], gys[i], ws[i], None, None, gcf[i]),
                                                    (arr[n + i], gys[n + i], ws[n + i], gys[2 * n + i], ws[2 * n + i], grf[i])):
                    q.gy_a, q.w_a, q.Ka = ptr(gy_a), ptr(w_a), w_a.shape[0]
                    q.gy_b, q.w_b, q.Kb = (ptr(gy_b), ptr(w_b), w_b.shape[0]) if gy_b is not None else (None, None, 0)
                    q.gx, q.N, q.C, q.HW = ptr(gx), gx.shape[0], gx.shape[1], gx.shape[2] * gx.shape[3]
                    nbytes += 4 * (gx.numel() + gy_a.numel() + (gy_b.numel() if gy_b is not None else 0))
            ops._call('eas_conv_fwd', nbytes, _lib.lib().eas_pred_dgrad_group, arr, 2 * n, stream())
        else:
            pk1 = [packs[i][1] if packs[i] is not None and 1 in packs[i] else ops.conv_pack_weights(ws[i], 1) for i in range(3 * n)]
            _launch_conv_group(gys[:2 * n], pk1[:2 * n], None, gcf + grf, None, 1)
            _launch_conv_group(gys[2 * n:], pk1[2 * n:], None, grf, None, 1, accumulate=True)
        geoms = tuple((x.shape[0], x.shape[1], g.shape[1], x.shape[2], x.shape[3]) for x, g in zip(xs, gys))
        ns = _wgrad_plan(geoms, 1)
        gws = []
        if ns is not None:
            m = 3 * n
            arr = (_lib.EasWgradProblem * m)()
            wss = []
            nbytes, fl = 0, 0.0
            for q, x, gy, s_ in zip(arr, xs, gys, ns):
                wsb = torch.empty(s_ * gy.shape[1] * x.shape[1], dtype=torch.float32, device=x.device)
                wss.append(wsb)
                q.x, q.grad_y, q.workspace = ptr(x), ptr(gy), ptr(wsb)
                q.NI, q.Cin, q.Cout, q.Hi, q.Wi = x.shape[0], x.shape[1], gy.shape[1], x.shape[2], x.shape[3]
                nbytes += 4 * (x.numel() + gy.numel())
                fl += 2.0 * gy.numel() * x.shape[1]
            defers = [ops._can_defer(w) for w in ws]

            def job(arr=arr, m=m, nbytes=nbytes, fl=fl):
                ops._call('eas_conv_wgrad', nbytes, _lib.lib().eas_conv_wgrad_group_partial, arr, m, 1, 3, stream(), flops=fl, issue_flops=6 * fl)
            ops._wgrad_launch(job, list(xs) + list(gys) + wss, all(defers), 6 * fl, nbytes)
            for w, gy, x, wsb, s_, df in zip(ws, gys, xs, wss, ns, defers):
                gw = torch.empty((gy.shape[1], x.shape[1], 1, 1), dtype=torch.float32, device=x.device)
                ops._wgrad_finish(wsb, gw, s_, df, w)
                gws.append(gw)
        else:
            for w, gy, x in zip(ws, gys, xs):
                gws.append(ops.conv_wgrad(x, gy, 1, 1, 3, None, defer=ops._can_defer(w), w=w))
        m = 3 * n
        arr = (_lib.EasChannelSumProblem * m)()
        gbs = []
        for q, gy in zip(arr, gys):
            gb = torch.empty(gy.shape[1], dtype=torch.float32, device=gy.device)
            gbs.append(gb)
            q.g, q.out, q.N, q.C, q.HW = ptr(gy), ptr(gb), gy.shape[0], gy.shape[1], gy.shape[2] * gy.shape[3]
        check(_lib.lib().eas_channel_sum_group(arr, m, stream()), 'eas_channel_sum_group')
        return (None,) + tuple(gcf) + tuple(grf) + tuple(gws) + tuple(gbs)


def pred_group_ok(cls_feats, reg_feats, cls_preds, reg_preds, obj_preds):
    n = len(cls_feats)
    if not ENABLED or 3 * n > 12 or ops.conv_sink() is not None or torch.nn.modules.module._global_forward_hooks:
        return False
    convs = list(cls_preds) + list(reg_preds) + list(obj_preds)
    xs = list(cls_feats) + list(reg_feats) + list(reg_feats)
    fwd, bwd = [], []
    for x, c in zip(xs, convs):
        if (type(c) is not torch.nn.Conv2d or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.padding != (0, 0) or c.groups != 1 or c.bias is None
                or c._forward_hooks or c._forward_pre_hooks or not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32)
                or c.in_channels != x.shape[1] or c.in_channels % 8 != 0 or ops.is_small_int(x) or ops.planes_of(x) is not None):
            return False
        fwd.append((x.shape[0], x.shape[1], c.out_channels, x.shape[2], x.shape[3]))
        bwd.append((x.shape[0], c.out_channels, x.shape[1], x.shape[2], x.shape[3]))
    return (_conv_plan(tuple(fwd), 1) is not None and _conv_plan(tuple(bwd[:2 * n]), 1) is not None
            and _conv_plan(tuple(bwd[2 * n:]), 1) is not None and _wgrad_plan(tuple(fwd), 1) is not None)


def pred_group(cls_feats, reg_feats, cls_preds, reg_preds, obj_preds):
    """(cls_outs, reg_outs, obj_outs) of all levels from one launch (caller checked ``pred_group_ok``)"""
    n = len(cls_feats)
    convs = list(cls_preds) + list(reg_preds) + list(obj_preds)
    packs = [_pack_of(c, 0) for c in convs]
    cfg = dict(n=n, packs=packs)
    res = _PredGroupFn.apply(cfg, *cls_feats, *reg_feats, *[c.weight for c in convs], *[c.bias for c in convs])
    return list(res[:n]), list(res[n:2 * n]), list(res[2 * n:])


# ------------------------------------------------------------------------------------------------ eval mode: conv -> BN -> SiLU of several layers, one launch
def conv_bn_act_eval_group_ok(xs, items, norms, k):
    """items[i] (an nn.Conv2d, or a pair that reads the same input) with norms[i] (their eval-mode BatchNorm2d modules, or -- after
    fuse_model -- bias tensors) can run as ONE grouped eas_conv_bn_act_eval on the inputs xs: every problem passes the single-launch check
    (``ops.fused_ann_eval_ok``) and the group has a common tile plan"""
    if not ENABLED or len(xs) > 8:
        return False
    geoms = []
    for x, it, nm in zip(xs, items, norms):
        convs = it if isinstance(it, tuple) else (it,)
        if not ops.fused_ann_eval_ok(x, it, list(nm)) or convs[0].kernel_size != (k, k) or convs[0].stride != (1, 1):
            return False
        geoms.append((x.shape[0], x.shape[1], sum(c.out_channels for c in convs), x.shape[2], x.shape[3]))
    return _conv_plan(tuple(geoms), k) is not None


def conv_bn_act_eval_group(xs, items, norms, k, owners=None):
    """[act(bn(conv_i(x_i)))] in one launch (caller checked ``conv_bn_act_eval_group_ok``); returns per problem one tensor, or two for a pair"""
    n = len(xs)
    arr = (_lib.EasConvBnActEval * n)()
    keep, results = [], []
    nbytes, fl = 0, 0.0
    for i, (d, x, it, nm) in enumerate(zip(arr, xs, items, norms)):
        convs = it if isinstance(it, tuple) else (it,)
        c0 = convs[0]
        NI, Cin, H, W = x.shape
        Cout = sum(c.out_channels for c in convs)
        if len(convs) == 2:
            packs = None
            if owners is not None and owners[i] is not None:
                packs = (getattr(owners[i][0], '_eas_dual_packs', None) or {}).get(owners[i][1])
                packs = ops.current_packs(packs)
            pk = packs[0] if packs else ops.conv_pack_weights(torch.cat([convs[0].weight, convs[1].weight], 0), 0)
        else:
            pk = _pack_of(c0, 0)
            pk = pk[0] if pk is not None else ops.conv_pack_weights(c0.weight, 0)
        xd = ops._f32c(x)
        keep += [pk, xd]
        d.x, d.x_terms, d.packed_w = ptr(xd), 3, ptr(pk)
        d.NI, d.Cin, d.Cout, d.Hi, d.Wi, d.ksize, d.stride = NI, Cin, Cout, H, W, k, 1
        d.act = 1
        d.csplit = convs[0].out_channels
        outs = []
        for j, bn in enumerate(nm):
            r = d.range[j]
            Cr = convs[j].out_channels if len(convs) > 1 else Cout
            if torch.is_tensor(bn):              # a block folded by fuse_model: z = acc * 1 + bias
                ones, zeros = ops._unit_affine(Cr, x.device)
                r.gamma, r.beta, r.mean, r.invstd = ptr(ones), ptr(bn), ptr(zeros), ptr(ones)
            else:
                invstd = ops._eval_invstd(bn.running_var, bn.eps)
                keep.append(invstd)
                r.gamma, r.beta, r.mean, r.invstd = ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(invstd)
            out = torch.empty((NI, Cr, H, W), dtype=torch.float32, device=x.device)
            r.out, r.out_ctot, r.out_c0 = ptr(out), Cr, 0
            outs.append(out)
        results.append(outs if len(nm) > 1 else outs[0])
        fl += 2.0 * NI * Cout * H * W * Cin * k * k
        nbytes += 4 * NI * (Cin + Cout) * H * W
    ops._call('eas_conv_fwd', nbytes, _lib.lib().eas_conv_bn_act_eval_group, arr, n, stream(), flops=fl, issue_flops=6 * fl)
    del keep
    return results


def pred_eval_ok(xs, convs):
    if not ENABLED or len(xs) > 12 or torch.is_grad_enabled() or ops.conv_sink() is not None or torch.nn.modules.module._global_forward_hooks:
        return False
    geoms = []
    for x, c in zip(xs, convs):
        if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and c.in_channels == x.shape[1] and c.in_channels % 8 == 0
                and c.stride == (1, 1) and c.padding == (0, 0) and c.groups == 1) or ops.is_small_int(x):
            return False
        geoms.append((x.shape[0], x.shape[1], c.out_channels, x.shape[2], x.shape[3]))
    return _conv_plan(tuple(geoms), 1) is not None


def conv_bias_group(xs, convs):
    """[conv_i(x_i) + bias_i] of plain 1x1 convolutions in one launch, no autograd (the prediction convolutions of the eval-mode head)"""
    pks = []
    for c in convs:
        pk = _pack_of(c, 0)
        pks.append(pk[0] if pk is not None else ops.conv_pack_weights(c.weight, 0))
    xs = [ops._f32c(x) for x in xs]
    ys = [torch.empty((x.shape[0], c.out_channels, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device) for x, c in zip(xs, convs)]
    _launch_conv_group(xs, pks, [c.bias for c in convs], ys, None, 1)
    return ys
