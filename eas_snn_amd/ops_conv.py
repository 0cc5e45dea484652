"""Operator layer, dense convolutions on the matrix cores: forward / input gradient / weight gradient (eas_conv_*), weight packing and its
scopes (``packed_weights``, ``frozen_weights``), dual convolutions, the weight-gradient side stream and deferred slab reductions, the
fused eval-mode blocks (conv -> BN -> LIF, conv -> BN -> SiLU).  ``eas_snn_amd.ops`` re-exports everything here (``ops.<name>``)."""
import ctypes as C
import os

import torch

from . import _lib
from ._ctx import ctx as opctx
from ._lib import check, ptr, stream
from .ops_core import FLAG_DECAY_INPUT, FLAG_FIRE_STRICT, FLAG_HARD_RESET, _call, _conv_stats_blocks, _dev, _eval_invstd, _f32c, _invstd_of_eval_model, _tag_flag, _timer_add, _timer_mark, _verify_tags, clear_conv_stats, dense, ghost, is_small_int, mark_small_int, new_planes, planes_of, set_tag, to_planes

def conv2d_weight(x, weight, stride=1, small_int=None):
    """functional form of ``conv2d`` for a weight tensor that is not a module parameter (e.g. two concatenated 1x1 weights)"""
    if small_int is None:
        small_int = is_small_int(x)
    if planes_of(x) is not None and not (small_int and _planes_conv_ok(x, weight, stride)):
        x = dense(x)
    _verify_tags(x, small_int)
    return _ConvFn.apply(x, weight, None, stride, 1 if small_int else 3, None, planes_of(x) if small_int else None)


# ------------------------------------------------------------------------------------------------ fused eval step (conv -> BN -> LIF, ONE kernel)
# In eval mode BatchNorm uses its running statistics, so nothing of a layer has to exist before its neurons can run: eas_conv_bn_lif_eval
# computes the convolution with the time steps of a pixel side by side in the wave tile, normalises in the epilogue and walks the neuron over
# T there -- the convolution output never reaches HBM (2 B per neuron-step written as spike planes instead of 4 + 4 + 2).  Same arithmetic
# as eas_conv_fwd + eas_bn_lif_fwd_ex: bit-identical spikes.  EAS_FUSED_EVAL=0: development switch, the two-kernel path.
# EAS_FUSED_EVAL: 'auto' (default) = where it was measured faster than the two kernels (batch 64, MI355X, scripts/dev_fused_layer.py): the 1x1
# layers on the direct kernel (fewer than 256 input channels: their y round trip is most of their time -- 64->64 at 64x80: 109 us against
# 166) and the 3x3 stride-1 layers on maps of >= 5120 pixels; the small-map and stride-2 layers (whose T-fold larger blocks quantise worse over
# the 256 CUs: 64->64 at 32x40 88 us against 80) and the shared-input first layer keep the two-kernel path.  'all' / '1': every eligible
# layer (the tests); '0': none.

_LIF_EVAL_SUPPORT = {}


def _plain_conv(conv):
    c = conv[0] if isinstance(conv, torch.nn.Sequential) and len(conv) == 1 else conv
    if not _static_conv_ok(c) or c.bias is not None or c._forward_hooks or c._forward_pre_hooks:
        return None
    return c


def _eval_neuron_ok(bn, node):
    """BatchNorm on running statistics in front of a soft-reset (P)LIF neuron with decay_input = False (what utils_snn.py:44-53 builds)"""
    if bn._use_batch_stats() or not bn.affine or bn.running_mean is None:
        return False
    a = node.lif_args()
    return (a['flags'] & (FLAG_HARD_RESET | FLAG_DECAY_INPUT | FLAG_FIRE_STRICT)) == 0


def fused_eval_ok(x_seq, conv, layers):
    """``conv`` (nn.Conv2d, possibly inside its SeqToANNContainer) followed by the (bn, node) pairs of ``layers`` on consecutive output-channel
    ranges can run as eas_conv_bn_lif_eval on ``x_seq`` [T,N,Cin,H,W]"""
    if not opctx.fused_eval or torch.is_grad_enabled() or opctx.conv_sink is not None or torch.nn.modules.module._global_forward_hooks:
        return False
    if not (torch.is_tensor(x_seq) and x_seq.is_cuda and x_seq.dim() == 5 and x_seq.dtype == torch.float32):
        return False
    if isinstance(conv, tuple):
        if not dual_weights_ok(*conv):
            return False
        c, cout = _plain_conv(conv[0]), conv[0].out_channels + conv[1].out_channels
    else:
        c = _plain_conv(conv)
        cout = c.out_channels if c is not None else 0
    if c is None or not all(_eval_neuron_ok(bn, node) for bn, node in layers):
        return False
    # a forward (pre-)hook on anything the fused kernel stands in for -- the convolution's SeqToANNContainer, the BatchNorm, the neuron --
    # would silently stop firing (the modules are not called): such a block takes the module-by-module path
    watched = [m for cv in (conv if isinstance(conv, tuple) else (conv,)) if isinstance(cv, torch.nn.Module) for m in (cv,)]
    watched += [m for pair in layers for m in pair]
    if any(m._forward_hooks or m._forward_pre_hooks for m in watched):
        return False
    if any(bn.num_features % 8 for bn, _ in layers) or len(layers) > 2 or sum(bn.num_features for bn, _ in layers) != cout:
        return False
    T, N, Cin, H, W = x_seq.shape
    base = getattr(x_seq, '_eas_base', None)
    if base is not None:
        xt, shared = (1 if is_small_int(base) else 3), 1
        if planes_of(base) is not None:
            return False
    else:
        if not is_small_int(x_seq) or Cin % 8 or (H * W) % 4:
            return False            # distinct real-valued frames per step: no tile form (the wave tile holds T steps of spike planes)
        xt, shared = 2, 0
    if opctx.fused_eval == 'auto':
        k_, s_ = c.kernel_size[0], c.stride[0]
        if shared or s_ != 1 or (k_ == 1 and Cin >= 256) or (k_ == 3 and H * W < 5120):
            return False
    key = (T, N, Cin, cout, H, W, c.kernel_size[0], c.stride[0], xt, shared)
    r = _LIF_EVAL_SUPPORT.get(key)
    if r is None:
        r = _LIF_EVAL_SUPPORT[key] = bool(_lib.lib().eas_conv_bn_lif_eval_supported(*key))
    return r


def current_packs(packs):
    """``packs`` if it is the packing made by the forward that is running now (``packed_weights``), else None"""
    return packs if (packs is not None and opctx.pack_scope is not None and packs.get('gen') == opctx.pack_scope) else None


def dual_weights_ok(conv_a, conv_b):
    """two convolutions that read the same input and can run as one with concatenated weights (same geometry, no bias)"""
    conv_a, conv_b = _plain_conv(conv_a), _plain_conv(conv_b)
    return (conv_a is not None and conv_b is not None and conv_a.kernel_size == conv_b.kernel_size
            and conv_a.stride == conv_b.stride == (1, 1) and conv_a.in_channels == conv_b.in_channels)


def conv_bn_lif_eval(x_seq, conv, layers, want_mean=False, residual=None, cats=None, planes=None, packs=None):
    """The fused eval step.  layers: [(bn, node)] or two of them (consecutive output-channel ranges of ``conv``: conv1 | conv2 of a CSPLayer,
    then ``conv`` is (conv_a, conv_b) and ``packs`` the packing of their concatenated weight or None); cats[i] = (buffer, first channel[,
    planes of the buffer]) or None; planes[i]: hand the spikes on as spike planes (ghost).  residual: SEW shortcut of a single layer.
    Returns one spike tensor per layer (a view into the concatenation buffer where given); with want_mean (single layer) (spikes, rate).
    Caller checked ``fused_eval_ok``."""
    L = _lib.lib()
    nl = len(layers)
    cats = cats or [None] * nl
    planes = planes or [False] * nl
    T, N, Cin, H, W = x_seq.shape
    dev = x_seq.device
    if isinstance(conv, tuple):
        ca, cb = _plain_conv(conv[0]), _plain_conv(conv[1])
        k, stride, Cout = ca.kernel_size[0], ca.stride[0], ca.out_channels + cb.out_channels
        pk = packs[0] if packs else conv_pack_weights(torch.cat([ca.weight, cb.weight], 0), 0)
    else:
        c = _plain_conv(conv)
        k, stride, Cout = c.kernel_size[0], c.stride[0], c.out_channels
        pk = getattr(c, '_eas_packs', None)
        pk = pk[0] if (pk is not None and opctx.pack_scope is not None and pk.get('gen') == opctx.pack_scope) else conv_pack_weights(c.weight, 0)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    HW = Ho * Wo
    base = getattr(x_seq, '_eas_base', None)
    d = _lib.EasConvBnLifEval()
    keep = [pk]
    if base is not None:
        xb = _f32c(base)
        keep.append(xb)
        d.x, d.x_terms, d.x_shared = ptr(xb), (1 if is_small_int(base) else 3), 1
    else:
        sp = planes_of(x_seq)
        if sp is None:
            _verify_tags(x_seq, True)
            sp = to_planes(x_seq)                       # a block called on its own with fp32 spikes (tests, hooks): one conversion
        sp = sp.contiguous()
        keep.append(sp)
        d.x, d.x_terms, d.x_shared = ptr(sp), 2, 0
    d.packed_w = ptr(pk)
    d.T, d.N, d.Cin, d.Cout, d.Hi, d.Wi, d.ksize, d.stride = T, N, Cin, Cout, H, W, k, stride
    d.csplit = layers[0][0].num_features
    assert sum(bn.num_features for bn, _ in layers) == Cout and (nl == 1 or d.csplit < Cout)
    write_v = opctx.state_writeback
    outs, rate = [], None
    for i, (bn, node) in enumerate(layers):
        r = d.range[i]
        Cr = bn.num_features
        a = node.lif_args()
        invstd = _eval_invstd(bn.running_var, bn.eps)
        keep.append(invstd)
        r.gamma, r.beta, r.mean, r.invstd = ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(invstd)
        w = a['w']
        r.w_logit, r.k_const, r.v_th = (ptr(w) if w is not None else None), float(a['k_const']), float(a['v_th'])
        cat = cats[i]
        cat_sp = cat is not None and len(cat) > 2 and cat[2] is not None
        want_planes = bool(cat_sp or (planes[i] and cat is None)) and opctx.spike_planes
        if cat is not None:
            buf, c0 = cat[0], int(cat[1])
            out = buf.narrow(2, c0, Cr)
            r.out_ctot, r.out_c0 = buf.shape[2], c0
            if cat_sp:
                assert cat[2].is_contiguous() and cat[2].shape == (T, N, buf.shape[2] // 8, HW, 8)
                r.planes, r.out_f32 = ptr(cat[2]), None
            else:
                assert buf.is_contiguous() and buf.dtype == torch.float32
                r.planes, r.out_f32 = None, ptr(buf)
        elif want_planes:
            sp_out = new_planes(T, N, Cr, Ho, Wo, dev)
            out = ghost((T, N, Cr, Ho, Wo), dev, sp_out)
            r.planes, r.out_f32, r.out_ctot, r.out_c0 = ptr(sp_out), None, Cr, 0
        else:
            out = torch.empty((T, N, Cr, Ho, Wo), dtype=torch.float32, device=dev)
            r.planes, r.out_f32, r.out_ctot, r.out_c0 = None, ptr(out), Cr, 0
        r.res_planes = r.res_f32 = None
        r.res_ctot = 0
        small_out = True
        if residual is not None:
            assert nl == 1 and residual.shape == (T, N, Cr, Ho, Wo)
            rs = planes_of(residual)
            if rs is not None:
                rs = rs.contiguous()
                keep.append(rs)
                r.res_planes, r.res_ctot = ptr(rs), Cr
            else:
                rf = _f32c(residual)
                keep.append(rf)
                r.res_f32, r.res_ctot = ptr(rf), Cr
                small_out = is_small_int(residual)
                if want_planes or cat_sp:
                    _verify_tags(residual, True)
                    if not small_out:
                        raise _lib.EasHipError('a real-valued shortcut cannot be added into spike planes')
        if want_mean:
            assert nl == 1
            rate = torch.empty((N, Cr, Ho, Wo), dtype=torch.float32, device=dev)
            r.rate = ptr(rate)
        else:
            r.rate = None
        v = node.v
        if torch.is_tensor(v):
            if tuple(v.shape) != (N, Cr, Ho, Wo):
                raise ValueError(f'membrane state {tuple(v.shape)} does not match the layer output {(N, Cr, Ho, Wo)}; call reset()')
            v_in = _f32c(v.detach())
        else:
            v_in = None if v == 0.0 else torch.full((N, Cr, Ho, Wo), float(v), dtype=torch.float32, device=dev)
        keep.append(v_in)
        r.v_in = ptr(v_in)
        v_out = torch.empty((N, Cr, Ho, Wo), dtype=torch.float32, device=dev) if write_v else None
        r.v_out = ptr(v_out)
        if v_out is not None:
            node.v = v_out
        if small_out:
            mark_small_int(out)
        outs.append(out)
    fl = 2.0 * T * N * Cout * HW * Cin * k * k / (T if d.x_shared else 1)
    nb = (2 if d.x_terms == 2 else 4) * (N if d.x_shared else T * N) * Cin * H * W + 2 * T * N * Cout * HW
    _call('eas_conv_bn_lif_eval', nb, L.eas_conv_bn_lif_eval, C.byref(d), stream(), flops=fl, issue_flops=fl * (6 if d.x_terms == 3 else 3))
    del keep
    if want_mean:
        return outs[0], rate
    return outs if nl > 1 else outs[0]


def conv_pack_weights(w, mode=0):
    """w [Cout,Cin,k,k] fp32 -> MFMA A-fragment order, three exact bf16 terms (mode 1: transposed+flipped for dgrad)."""
    _dev(w)
    w = _f32c(w)
    Cout, Cin, k = w.shape[0], w.shape[1], w.shape[-1]
    L = _lib.lib()
    packed = torch.empty(L.eas_conv_packed_weight_bytes(Cout, Cin, k, mode), dtype=torch.uint8, device=w.device)
    check(L.eas_conv_pack_weights(ptr(w), ptr(packed), Cout, Cin, k, mode, stream()), 'eas_conv_pack_weights')
    return packed


def conv_fwd_packed(x, packed, bias, Cout, ksize, stride, x_terms, x_sp=None, act=0):
    """conv2d(x [NI,Cin,H,W], packed weights), padding ksize//2, on the matrix cores.  x_sp: x is a ghost and these are its spike planes
    [NI,Cin/8,H*W,8] (eas_conv_fwd_planes: same result, half the input bytes, no conversion)."""
    _dev(x, packed, bias)
    NI, Cin, Hi, Wi = x.shape
    pad = ksize // 2
    Ho, Wo = (Hi + 2 * pad - ksize) // stride + 1, (Wi + 2 * pad - ksize) // stride + 1
    y = torch.empty((NI, Cout, Ho, Wo), dtype=torch.float32, device=packed.device)
    fl = 2.0 * y.numel() * Cin * ksize * ksize
    L = _lib.lib()
    if x_sp is not None:
        assert x_terms == 1 and x_sp.dtype == torch.bfloat16 and x_sp.shape == (NI, Cin // 8, Hi * Wi, 8)
        x_sp = x_sp.contiguous()
        stats, nb = None, 0
        if opctx.want_conv_stats and bias is None:
            nb = _conv_stats_blocks(L, (NI, Cin, Cout, Hi, Wi, ksize, stride, 2))
            if 0 < nb <= opctx.conv_stats_max_blocks:
                stats = torch.empty(Cout * nb * 2, dtype=torch.float64, device=y.device)
        _call('eas_conv_fwd', 2 * NI * Cin * Hi * Wi + 4 * y.numel(), L.eas_conv_fwd_planes, ptr(x_sp), ptr(packed), ptr(bias), ptr(y), NI, Cin, Cout,
              Hi, Wi, ksize, stride, ptr(stats), nb if stats is not None else 0, stream(), flops=fl, issue_flops=fl * 3)
        if stats is not None:
            opctx.conv_stats_slot = (y, nb, stats, y._version)
        return y
    x = _f32c(x)
    if act:             # y = act(conv + bias) from the epilogue (eas_conv_fwd_act: an eval-mode BaseConv with its BatchNorm folded in)
        _call('eas_conv_fwd', 4 * (x.numel() + y.numel()), L.eas_conv_fwd_act, ptr(x), ptr(packed), ptr(bias), ptr(y), NI, Cin, Cout, Hi, Wi,
              ksize, stride, x_terms, int(act), _tag_flag(x, x_terms), stream(), flops=fl, issue_flops=fl * (3 if x_terms == 1 else 6))
        return y
    if opctx.want_conv_stats and bias is None:
        nb = _conv_stats_blocks(L, (NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms))
        if 0 < nb <= opctx.conv_stats_max_blocks:
            stats = torch.empty(Cout * nb * 2, dtype=torch.float64, device=x.device)
            _call('eas_conv_fwd', 4 * (x.numel() + y.numel()), L.eas_conv_fwd_stats, ptr(x), ptr(packed), ptr(y), NI, Cin, Cout, Hi, Wi,
                  ksize, stride, x_terms, _tag_flag(x, x_terms), ptr(stats), nb, stream(), flops=fl, issue_flops=fl * (3 if x_terms == 1 else 6))
            opctx.conv_stats_slot = (y, nb, stats, y._version)
            return y
    _call('eas_conv_fwd', 4 * (x.numel() + y.numel()), L.eas_conv_fwd, ptr(x), ptr(packed), ptr(bias), ptr(y), NI, Cin, Cout, Hi, Wi,
          ksize, stride, x_terms, _tag_flag(x, x_terms), stream(), flops=fl, issue_flops=fl * (3 if x_terms == 1 else 6))
    return y


# Deferred slab reductions.  Every weight gradient is slab kernel + fixed-order reduction of the slabs.  A training loop that
# owns its backward pass can switch ``deferred_wgrad_reductions`` on: the ~80 reductions of a step (each a few microseconds of work
# behind a launch) are then collected and done by ONE launch when the autograd engine finishes the pass
# (eas_conv_wgrad_reduce_many via queue_callback) -- the gradient tensors handed to autograd are filled a little later on the same
# stream.  That is only sound when nothing reads a weight gradient DURING the backward pass, so it is off by default and the
# operator falls back to the immediate reduction whenever it cannot know: a parameter that already has a ``.grad`` (accumulation
# over several backward passes, DistributedDataParallel's bucket views), a weight that contributes twice to one pass (everything
# pending is reduced first), any call outside a backward pass.  DistributedDataParallel with copy hooks reads gradients inside the
# pass: leave the switch off there.  Same summation order either way: bit-identical gradients.



class deferred_wgrad_reductions:
    """``with ops.deferred_wgrad_reductions():`` around forward + backward of a training step (see above)"""

    def __init__(self, flag=True):
        self.flag = bool(flag)

    def __enter__(self):
        self.prev, opctx.defer_wgrad_reduce = opctx.defer_wgrad_reduce, self.flag

    def __exit__(self, exc_type, *exc):
        opctx.defer_wgrad_reduce = self.prev
        if exc_type is not None:
            # the backward pass did not finish: the gradient tensors of the pending jobs may already be gone (their addresses are all
            # that is kept) -- nothing is written, the jobs are dropped
            opctx.pending_reduce = []
            opctx.side['pending'], opctx.side['keep'] = [], []
            return
        _flush_wgrad_reductions()


# Weight-gradient slab kernels on a side stream (round 5).  Nothing reads a deferred weight gradient before the end of the backward pass, so
# its slab kernel is off the pass's dependency chain: with opctx.wgrad_side_batch = n > 0 the launches are collected and, n at a time, issued on
# ONE side stream that forks from the main stream where the batch's last grad_y exists (one cross-stream edge per batch, not per launch:
# per-launch forks measured slower than no overlap) and joins in front of the batched reduction.  The chain's many small
# kernels (one or two waves per SIMD on the 8x10 / 16x20 maps) leave most of the chip idle; the slab kernels fill it.  Inputs and
# workspaces of the launches in flight are kept alive until the join, so the allocator cannot hand their memory to the main stream early.



def _wgrad_launch(job, keep, defer, issue_flops=0.0, nbytes=0.0):
    """run the slab-kernel launch ``job()`` now, or -- a deferred gradient with the side stream switched on -- with the next batch"""
    if not (defer and (opctx.wgrad_side_batch > 0 or opctx.wgrad_side_us > 0 or opctx.wgrad_side_at) and opctx.defer_wgrad_reduce):
        job()
        return
    opctx.side['pending'].append(job)
    opctx.side['keep'].extend(keep)
    opctx.side['us'] += max(issue_flops / 1.3e15, nbytes / 5e12) * 1e6
    opctx.side['seen'] += 1
    if ((opctx.wgrad_side_batch > 0 and len(opctx.side['pending']) >= opctx.wgrad_side_batch) or (opctx.wgrad_side_us > 0 and opctx.side['us'] >= opctx.wgrad_side_us)
            or opctx.side['seen'] in opctx.wgrad_side_at):
        _side_flush()


def _side_flush():
    jobs, opctx.side['pending'] = opctx.side['pending'], []
    opctx.side['us'] = 0.0
    if not jobs:
        return
    if opctx.side['stream'] is None:
        opctx.side['stream'] = _lib.private_stream()          # joins graph captures: never a pooled stream (see there)
    side = opctx.side['stream']
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for job in jobs:
            job()
    opctx.side['dirty'] = True


def _side_join():
    _side_flush()
    if opctx.side['dirty']:
        torch.cuda.current_stream().wait_stream(opctx.side['stream'])
        opctx.side['dirty'] = False
    if os.environ.get('EAS_WGRAD_SIDE_LOG') and opctx.side['seen']:
        print(f"[eas] side stream: {opctx.side['seen']} slab launches in this pass", flush=True)
    opctx.side['keep'] = []
    opctx.side['seen'] = 0


def _flush_wgrad_reductions():
    """Reduce every pending slab set into its gradient tensor in ONE launch.  A job is only valid while the tensor handed to autograd is
    still the parameter's ``.grad`` (AccumulateGrad adopted it, nothing has been added to it in place): anything else means the
    deferral was switched on for a backward pass it is not sound for, and fails loudly instead of writing through a stale address."""
    _side_join()                 # slab kernels still pending or running on the side stream: the reduction reads their slabs
    jobs, opctx.pending_reduce = opctx.pending_reduce, []
    if not jobs:
        return
    for w, wptr, wn in (o for job in jobs for o in job[4]):
        g = w.grad
        if g is None or g.data_ptr() != wptr or g.numel() != wn or g._version != 0:
            raise _lib.EasHipError(
                'deferred_wgrad_reductions: a weight gradient handed to autograd is not (or no longer exclusively) the parameter\'s .grad '
                '-- gradient hooks, create_graph, a second consumer of the weight or a non-contiguous weight; run this backward pass '
                'without the deferral (the unreduced buffer was NOT written)')
    arr = (_lib.EasWgradReduceJob * len(jobs))(*[_lib.EasWgradReduceJob(ws.data_ptr(), gptr, gn, n) for ws, gptr, gn, n, _ in jobs])
    t0 = _timer_mark()
    check(_lib.lib().eas_conv_wgrad_reduce_many(arr, len(jobs), stream()), 'eas_conv_wgrad_reduce_many')
    _timer_add('eas_conv_wgrad', t0, 0)


def _can_defer(w):
    """the slab reduction of this weight's gradient may wait for the end of the backward pass: the gradient tensor returned now will be
    adopted as ``w.grad`` by AccumulateGrad as it is (leaf without a gradient yet, plain backward without create_graph, no tensor
    hooks that could read or replace it, the layout autograd expects)"""
    return (opctx.defer_wgrad_reduce and w.is_leaf and w.grad is None and not torch.is_grad_enabled() and not w._backward_hooks
            and not getattr(w, '_post_accumulate_grad_hooks', None) and w.is_contiguous())


def _wgrad_finish(ws, gw, nslabs, defer, w=None, split=None):
    """reduce the slabs in ``ws`` into ``gw``: now, or together with the other weight gradients at the end of the backward pass.
    Only the ADDRESS of ``gw`` is kept (plus the parameter it belongs to): autograd must stay the sole owner of the tensor so that
    AccumulateGrad adopts it as the parameter's ``.grad`` instead of cloning it (a clone would be taken before the reduction has
    run); ``_flush_wgrad_reductions`` checks that this is what happened before it writes."""
    # owners: the parameter(s) whose .grad this gradient becomes -- one, or the two whose row blocks [0, split) / [split, Cout) it holds (the
    # one weight gradient of two convolutions computed as one: each parameter's .grad is then a view of ``gw``)
    owners = []
    if w is not None:
        if split is None:
            owners = [(w, gw.data_ptr(), gw.numel())]
        else:
            na = split * gw[0].numel()
            owners = [(w[0], gw.data_ptr(), na), (w[1], gw.data_ptr() + 4 * na, gw.numel() - na)]
    if defer and owners and any(o[0] is p[0] for j in opctx.pending_reduce for p in j[4] for o in owners):
        _flush_wgrad_reductions()           # a second contribution to the same weight: autograd adds the two as soon as this returns
        defer = False
    if defer and owners:
        if not opctx.pending_reduce:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrad_reductions)
        opctx.pending_reduce.append((ws, gw.data_ptr(), gw.numel(), int(nslabs), owners))
    else:
        arr = (_lib.EasWgradReduceJob * 1)(_lib.EasWgradReduceJob(ws.data_ptr(), gw.data_ptr(), gw.numel(), int(nslabs)))
        check(_lib.lib().eas_conv_wgrad_reduce_many(arr, 1, stream()), 'eas_conv_wgrad_reduce_many')


def _partial_call(name, nbytes, fn, *args, flops=0.0, issue_flops=0.0):
    """like _call for the *_partial entry points, whose non-negative return value is the slab count"""
    if opctx.call_log is not None:
        opctx.call_log.append((fn.__name__, args))
    if opctx.timer is None:
        rc = fn(*args)
    else:
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        rc = fn(*args)
        e_.record()
        opctx.timer.add(name, s_, e_, nbytes, flops, issue_flops)
    if rc <= 0:
        check(rc if rc < 0 else -2, name)
    return rc


def conv_wgrad(x, gy, ksize, stride, x_terms, x_sp=None, defer=False, w=None, split=None):
    """grad_w [Cout,Cin,k,k] of conv2d(x, w, stride, padding k//2) given grad_y, on the matrix cores (deterministic).
    x_sp: x is a ghost, these are its spike planes (eas_conv_wgrad_planes_partial)."""
    _dev(gy)
    gy = _f32c(gy)
    NI, Cin, Hi, Wi = x.shape
    Cout = gy.shape[1]
    L = _lib.lib()
    xt = 2 if x_sp is not None else x_terms
    nws = L.eas_conv_wgrad_workspace_floats(NI, Cin, Cout, Hi, Wi, ksize, stride, xt)
    if nws <= 0:
        raise _lib.EasHipError('eas_conv_wgrad: unsupported configuration')
    ws = torch.empty(nws, dtype=torch.float32, device=gy.device)
    gw = torch.empty((Cout, Cin, ksize, ksize), dtype=torch.float32, device=gy.device)
    fl = 2.0 * gy.numel() * Cin * ksize * ksize
    ns = nws // (Cout * Cin * ksize * ksize)         # the slab count the launch returns (checked in the job)
    if x_sp is not None:
        x_sp = x_sp.contiguous()

        def job():
            got = _partial_call('eas_conv_wgrad', 2 * NI * Cin * Hi * Wi + 4 * gy.numel(), L.eas_conv_wgrad_planes_partial, ptr(x_sp), ptr(gy), ptr(ws),
                                NI, Cin, Cout, Hi, Wi, ksize, stride, stream(), flops=fl, issue_flops=fl * 3)
            assert got == ns, (got, ns)
        keep = (x_sp, gy, ws)
    else:
        _dev(x)
        x = _f32c(x)

        def job():
            got = _partial_call('eas_conv_wgrad', 4 * (x.numel() + gy.numel()), L.eas_conv_wgrad_partial, ptr(x), ptr(gy), ptr(ws), NI, Cin, Cout,
                                Hi, Wi, ksize, stride, x_terms, stream(), flops=fl, issue_flops=fl * (3 if x_terms == 1 else 6))
            assert got == ns, (got, ns)
        keep = (x, gy, ws)
    _wgrad_launch(job, keep, defer, fl * (3 if (x_sp is not None or x_terms == 1) else 6), 4.0 * gy.numel() + (2.0 if x_sp is not None else 4.0) * NI * Cin * Hi * Wi)
    _wgrad_finish(ws, gw, ns, defer, w, split)
    return gw


def conv_eligible(x, conv):
    k = conv.kernel_size[0]
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.groups == 1 and conv.dilation == (1, 1)
            and conv.kernel_size in ((1, 1), (3, 3)) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (k // 2, k // 2)
            and conv.padding_mode == 'zeros' and x.shape[1] % 8 == 0 and (k == 1 or x.shape[-1] % 2 == 0)
            and not (k == 1 and conv.stride != (1, 1)))


_FWD_SUPPORT = {}


def conv_fwd_supported(NI, Cin, Cout, Hi, Wi, k, stride, x_terms):
    """geometry query of eas_conv_fwd (cached): False -> the layer keeps the library convolution"""
    key = (NI, Cin, Cout, Hi, Wi, k, stride, x_terms)
    r = _FWD_SUPPORT.get(key)
    if r is None:
        r = _FWD_SUPPORT[key] = bool(_lib.lib().eas_conv_fwd_supported(*key))
    return r


def _planes_conv_ok(x, w, stride):
    """forward and weight gradient of conv(x [NI,Cin,H,W] as spike planes, w) both have a planes kernel for this geometry"""
    NI, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[-1]
    if not conv_fwd_supported(NI, Cin, Cout, H, W, k, stride, 2):
        return False
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return bool(_lib.lib().eas_conv_wgrad_workspace_floats(NI, Cin, Cout, H, W, k, stride, 2) > 0
                and (k == 1 or (Cout % 8 == 0 and Wo % 2 == 0 and (Ho * Wo) % 4 == 0)))


def channel_sum(g):
    """g [N,C,H,W] -> [C]: the bias gradient of a convolution (g.sum((0, 2, 3))) as one launch of one block per channel, fixed order
    (eas_channel_sum); ATen's reduction takes 12-27 us for the few-channel prediction maps of the head"""
    if not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 4):
        return g.sum((0, 2, 3))
    g = g.contiguous()
    N, C_, H, W = g.shape
    out = torch.empty(C_, dtype=torch.float32, device=g.device)
    check(_lib.lib().eas_channel_sum(ptr(g), ptr(out), N, C_, H * W, stream()), 'eas_channel_sum')
    return out


# EAS_SMALL_DGRAD=0: the stem's input gradient through eas_conv_fwd (8 of 32 tile rows), as before round 5



class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, x_terms, packs, x_sp=None):
        """packs: {mode: packed weights} made from the current values of ``w`` (packed_weights scope), or None.
        x_sp: x is a ghost (see ``ghost``) and these are its spike planes: forward and weight gradient read them."""
        _dev(x, w, bias)
        k, Cout = w.shape[-1], w.shape[0]
        if x_sp is not None:
            assert x_terms == 1
        if conv_fwd_supported(x.shape[0], x.shape[1], Cout, x.shape[2], x.shape[3], k, stride, 2 if x_sp is not None else x_terms):
            pk = packs[0] if packs and 0 in packs else conv_pack_weights(w, 0)
            y = conv_fwd_packed(x, pk, bias, Cout, k, stride, x_terms, x_sp)
        else:       # no tile for this geometry (not reached by the EAS-SNN models): library forward
            if x_sp is not None:
                raise _lib.EasHipError('no matrix-core tile for a convolution on spike planes')
            y = torch.ops.aten.convolution(x, w, bias, (stride, stride), (k // 2, k // 2), (1, 1), False, (0, 0), 1)
        ctx.save_for_backward(x, w, x_sp)
        ctx.cfg = (k, stride, x_terms, bias is not None)
        ctx.packs = packs        # valid for the backward of this forward (same weights; autograd forbids changing them in between)
        ctx.tag = opctx.tag
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, x_sp = ctx.saved_tensors
        k, stride, x_terms, has_bias = ctx.cfg
        packs = ctx.packs
        gy = _f32c(gy)
        gx = gw = gb = None
        Cin = w.shape[1]
        if opctx.timer is not None and ctx.tag is not None:
            set_tag(ctx.tag)
        # the input gradient of a stride-1 convolution is eas_conv_fwd on grad_y with the weights packed transposed + flipped
        # (1x1: any channel count -- the 1/4/num_classes-channel prediction convolutions included; 3x3: Cout in whole 8-groups)
        own_d = (ctx.needs_input_grad[0] and stride == 1 and (k == 1 or w.shape[0] % 8 == 0)
                 and conv_fwd_supported(gy.shape[0], w.shape[0], Cin, gy.shape[2], gy.shape[3], k, 1, 3))
        own_w = ctx.needs_input_grad[1] and _lib.lib().eas_conv_wgrad_workspace_floats(
            x.shape[0], Cin, w.shape[0], x.shape[2], x.shape[3], k, stride, 2 if x_sp is not None else x_terms) > 0 and (
                k == 1 or (w.shape[0] % 8 == 0 and gy.shape[-1] % 2 == 0 and (gy.shape[-1] * gy.shape[-2]) % 4 == 0))
        own_d2 = (ctx.needs_input_grad[0] and stride == 2 and k == 3 and w.shape[0] % 8 == 0
                  and gy.shape[-1] % 2 == 0 and conv_fwd_supported(gy.shape[0], w.shape[0], Cin, gy.shape[2], gy.shape[3] + 2, k, 1, 3))
        small_d = (opctx.small_dgrad and ctx.needs_input_grad[0] and stride == 1 and k == 3 and w.dtype == torch.float32
                   and _lib.lib().eas_conv_dgrad_small_supported(gy.shape[0], Cin, w.shape[0], gy.shape[2], gy.shape[3]))
        if small_d:
            # at most 8 input channels (the stem): the taps stacked along M instead of 8 rows of a 32-row tile (conv_small_dgrad.hip)
            gx = torch.empty((gy.shape[0], Cin, gy.shape[2], gy.shape[3]), dtype=torch.float32, device=gy.device)
            fl = 2.0 * gy.numel() * Cin * 9
            ksteps = (w.shape[0] + 15) // 16
            _call('eas_conv_fwd', 4 * (gx.numel() + gy.numel()), _lib.lib().eas_conv_dgrad_small, ptr(gy), ptr(_f32c(w)), ptr(gx), gy.shape[0], Cin,
                  w.shape[0], gy.shape[2], gy.shape[3], stream(), flops=fl,
                  issue_flops=6 * 2.0 * gy.shape[0] * gy.shape[2] * gy.shape[3] * 96 * 16 * ksteps)
            own_d = True
        elif own_d:
            pk = packs[1] if packs and 1 in packs else conv_pack_weights(w, 1)
            gx = conv_fwd_packed(gy, pk, None, Cin, k, 1, 3)
        elif own_d2:
            gx = torch.empty_like(x)
            fl = 2.0 * gy.numel() * Cin * 9
            pk = packs[2] if packs and 2 in packs else conv_pack_weights(w, 2)
            try:
                _call('eas_conv_fwd', 4 * (x.numel() + gy.numel()), _lib.lib().eas_conv_dgrad_s2, ptr(gy), ptr(pk), ptr(gx),
                      x.shape[0], Cin, w.shape[0], x.shape[2], x.shape[3], stream(), flops=fl, issue_flops=6 * fl)
                own_d = True
            except _lib.EasHipError:        # no tile for this width (checked before anything is launched): library input gradient
                gx = None
        if own_w:
            gw = conv_wgrad(x, gy, k, stride, x_terms, x_sp, defer=_can_defer(w), w=w)
        need_d = ctx.needs_input_grad[0] and not own_d
        need_w = ctx.needs_input_grad[1] and not own_w
        if (need_d or need_w) and x_sp is not None:
            raise _lib.EasHipError('no matrix-core kernel for the gradients of a convolution on spike planes')
        if need_d or need_w:
            rx, rw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, (stride, stride), (k // 2, k // 2), (1, 1), False, (0, 0), 1,
                                                            (need_d, need_w, False))
            gx = rx if need_d else gx
            gw = rw if need_w else gw
        if has_bias and ctx.needs_input_grad[2]:
            gb = channel_sum(gy)
        return gx, gw, gb, None, None, None, None


def _static_conv_ok(conv):
    if type(conv) is not torch.nn.Conv2d:
        return False
    k = conv.kernel_size[0]
    return (conv.groups == 1 and conv.dilation == (1, 1) and conv.kernel_size in ((1, 1), (3, 3))
            and conv.stride in ((1, 1), (2, 2)) and conv.padding == (k // 2, k // 2) and conv.padding_mode == 'zeros'
            and conv.in_channels % 8 == 0 and not (k == 1 and conv.stride != (1, 1)))


class frozen_weights:
    """``with ops.frozen_weights(model):`` around a run of INFERENCE forwards during which nobody writes the model's weights or BatchNorm
    statistics -- the evaluator's loop over the validation set (yolox/evaluators/event_evaluator.py:150-215 of the reference runs under
    ``model.eval()`` + ``torch.no_grad()`` from its first to its last batch).  The packed convolution weights and the per-layer BatchNorm
    constants are computed ONCE when the block is entered, into buffers that persist on the model, and every ``packed_weights`` forward
    inside reuses them instead of launching the packing / constant kernels again (50 + 10 us of a 4.8 ms eval forward of config 2).  A HIP
    graph captured inside the block therefore contains neither: entering the block again (the next evaluation, after more training) refreshes
    the same buffers, so a replay reads current values.  The general rule stays what ``prepack_conv_weights`` says: outside such a block
    nothing is cached across forwards."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.prev = opctx.frozen
        p = next(self.model.parameters(), None)
        if p is None or not p.is_cuda:
            return self
        opctx.frozen = None                                   # (a nested block for another model: pack for real)
        with torch.no_grad():
            gen = prepack_conv_weights(self.model)
            mods = list(self.model.modules())
            fresh = _invstd_of_eval_model(mods)
            inv = None
            if fresh is not None:
                keep = getattr(self.model, '_eas_invstd_persist', None)
                if keep is None or set(keep) != set(fresh) or any(keep[k][1].shape != fresh[k][1].shape for k in fresh):
                    keep = {k: (eps, torch.empty_like(t)) for k, (eps, t) in fresh.items()}
                    object.__setattr__(self.model, '_eas_invstd_persist', keep)
                keys = list(fresh)
                torch._foreach_copy_([keep[k][1] for k in keys], [fresh[k][1] for k in keys])
                inv = {k: (fresh[k][0], keep[k][1]) for k in keys}
        if gen is not None:
            opctx.frozen = {'model': self.model, 'gen': gen, 'inv': inv}
        return self

    def __exit__(self, *exc):
        opctx.frozen = self.prev
        return False


def prepack_conv_weights(model):
    """Pack the weights of every eligible nn.Conv2d of ``model`` for the matrix-core kernels in ONE launch
    (eas_conv_pack_weights_many): forward order, plus the transposed orders the input gradients need.  Returns the
    generation number of the packing.

    Called at the start of EVERY forward (``packed_weights``): there is no cheap, reliable way to learn that a weight has
    changed -- ``torch.optim.Adam(fused=True)`` (event_yolox_base.py get_optimizer) and ``p.data`` writes change the values
    without touching ``Tensor._version`` -- so nothing is cached across forwards.  One launch over all weights (~150 MB of
    traffic for SYOLOX-S, < 0.1 ms) instead of 111 tiny packing kernels; inside a captured HIP graph the launch is part of the
    graph, so every replay packs the weights the optimizer has just written."""
    if opctx.frozen is not None and opctx.frozen['model'] is model and not torch.is_grad_enabled() and not model.training:
        return opctx.frozen['gen']                            # inside ``frozen_weights(model)``: packed when the block was entered
    # without autograd (inference) only the forward order is packed: a third of the work of the training plan; a consumer that still asks
    # for a transposed order packs it on demand
    fwd_only = not torch.is_grad_enabled()
    plan_attr = '_eas_pack_plan_fwd' if fwd_only else '_eas_pack_plan'
    plan = getattr(model, plan_attr, None)
    convs = plan['convs'] if plan else [m for m in model.modules() if _static_conv_ok(m) and m.weight.is_cuda]
    if not convs:
        return None
    # pairs of convolutions that read the same input and run as ONE convolution (``conv2d_dual``): modules that have them list them as
    # eas_dual_convs() -> [(key, conv_a, conv_b)]; their concatenated weight is packed straight from the two parameters
    duals = plan['duals'] if plan else [(m, key, a, b) for m in model.modules() if hasattr(m, 'eas_dual_convs') for key, a, b in m.eas_dual_convs()
                                        if a.weight.is_cuda]
    L = _lib.lib()
    ptrs = tuple(c.weight.data_ptr() for c in convs) + tuple(p for _, _, a, b in duals for p in (a.weight.data_ptr(), b.weight.data_ptr()))
    if not plan or plan['ptrs'] != ptrs:
        jobs, packs, dpacks = [], [], []
        dev = convs[0].weight.device
        for c in convs:
            k, Cout, Cin = c.kernel_size[0], c.out_channels, c.in_channels
            modes = [0] if fwd_only else ([0] + ([1] if c.stride == (1, 1) and (k == 1 or Cout % 8 == 0) else [])
                                          + ([2] if c.stride == (2, 2) and k == 3 and Cout % 8 == 0 else []))
            d = {}
            for m in modes:
                d[m] = torch.empty(L.eas_conv_packed_weight_bytes(Cout, Cin, k, m), dtype=torch.uint8, device=dev)
                jobs.append([c.weight.data_ptr(), d[m].data_ptr(), Cout, Cin, k, m, 0, 0])
            packs.append(d)
        for _, _, a, b in duals:
            k, Ca, Cout, Cin = a.kernel_size[0], a.out_channels, a.out_channels + b.out_channels, a.in_channels
            d = {}
            for m in ((0,) if fwd_only else (0, 1)):
                d[m] = torch.empty(L.eas_conv_packed_weight_bytes(Cout, Cin, k, m), dtype=torch.uint8, device=dev)
                jobs.append([a.weight.data_ptr(), d[m].data_ptr(), Cout, Cin, k, m, b.weight.data_ptr(), Ca])
            dpacks.append(d)
        plan = {'convs': convs, 'packs': packs, 'duals': duals, 'dpacks': dpacks, 'ptrs': ptrs, 'njobs': len(jobs),
                'jobs': torch.tensor(jobs, dtype=torch.int64).to(dev)}
        object.__setattr__(model, plan_attr, plan)
    check(L.eas_conv_pack_weights_many(ptr(plan['jobs']), plan['njobs'], stream()), 'eas_conv_pack_weights_many')
    opctx.pack_gen += 1
    for c, d in zip(convs, plan['packs']):
        d['gen'] = opctx.pack_gen
        object.__setattr__(c, '_eas_packs', d)
    for (owner, key, _, _), d in zip(duals, plan['dpacks']):
        d['gen'] = opctx.pack_gen
        reg = getattr(owner, '_eas_dual_packs', None)
        if reg is None:
            reg = {}
            object.__setattr__(owner, '_eas_dual_packs', reg)
        reg[key] = d
    return opctx.pack_gen


class _ConvDualFn(torch.autograd.Function):
    """conv(x, [wa ; wb]) for two stride-1 convolutions without bias that read the same x (same kernel size): one forward, ONE input
    gradient (no addition of two branch gradients), one weight-gradient launch whose result is handed out as its two row blocks."""

    @staticmethod
    def forward(ctx, x, wa, wb, x_terms, packs, x_sp=None):
        _dev(x, wa, wb)
        k, Ca, Cout = wa.shape[-1], wa.shape[0], wa.shape[0] + wb.shape[0]
        pk = packs[0] if packs else conv_pack_weights(torch.cat([wa, wb], 0), 0)
        y = conv_fwd_packed(x, pk, None, Cout, k, 1, x_terms, x_sp)
        ctx.save_for_backward(x, wa, wb, x_sp)
        ctx.cfg = (k, x_terms, Ca)
        ctx.packs = packs
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wa, wb, x_sp = ctx.saved_tensors
        k, x_terms, Ca = ctx.cfg
        gy = _f32c(gy)
        gx = ga = gb = None
        if ctx.needs_input_grad[0]:
            pk = ctx.packs[1] if ctx.packs and 1 in ctx.packs else conv_pack_weights(torch.cat([wa, wb], 0), 1)
            gx = conv_fwd_packed(gy, pk, None, x.shape[1], k, 1, 3)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # ONE weight gradient for the two parameters: its slab reduction waits for the end of the backward pass with all the others
            # (each parameter adopts its row block of the one tensor as .grad) where that is sound for both, else it runs now
            both = ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
            gw = conv_wgrad(x, gy, k, 1, x_terms, x_sp, defer=both and _can_defer(wa) and _can_defer(wb), w=(wa, wb) if both else None,
                            split=Ca if both else None)
            ga, gb = gw[:Ca], gw[Ca:]
        return gx, ga, gb, None, None, None


def conv_dual_ok(x, conv_a, conv_b):
    """the two convolutions can run as one (``conv2d_dual``) on ``x`` [NI,Cin,H,W]"""
    if os.environ.get('EAS_DUAL_CONV', '1') != '1':       # development switch: concatenate the weights with torch.cat instead
        return False
    if not (_static_conv_ok(conv_a) and _static_conv_ok(conv_b) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return False
    k, Cin, Cout = conv_a.kernel_size[0], conv_a.in_channels, conv_a.out_channels + conv_b.out_channels
    if (conv_b.kernel_size[0] != k or conv_b.in_channels != Cin or conv_a.stride != (1, 1) or conv_b.stride != (1, 1)
            or conv_a.bias is not None or conv_b.bias is not None or Cout % 8 != 0):
        return False
    if any(c._forward_hooks or c._forward_pre_hooks for c in (conv_a, conv_b)) or torch.nn.modules.module._global_forward_hooks:
        return False
    NI, _, H, W = x.shape
    xt = 2 if planes_of(x) is not None else (1 if is_small_int(x) else 3)
    L = _lib.lib()
    return bool(conv_fwd_supported(NI, Cin, Cout, H, W, k, 1, xt) and conv_fwd_supported(NI, Cout, Cin, H, W, k, 1, 3)
                and L.eas_conv_wgrad_workspace_floats(NI, Cin, Cout, H, W, k, 1, xt) > 0 and (k == 1 or (W % 2 == 0 and (H * W) % 4 == 0)))


def conv2d_dual(x, conv_a, conv_b, owner=None, key=None):
    """[conv_a(x) ; conv_b(x)] along the channels as ONE convolution (caller checked ``conv_dual_ok``).  owner / key: where
    ``packed_weights`` left the packing of the concatenated weight (owner.eas_dual_convs())."""
    small = is_small_int(x)
    _verify_tags(x, small)
    packs = None
    if owner is not None:
        packs = (getattr(owner, '_eas_dual_packs', None) or {}).get(key)
        if packs is not None and (opctx.pack_scope is None or packs.get('gen') != opctx.pack_scope):
            packs = None        # not inside the forward that made this packing: pack the weights as they are now
    return _ConvDualFn.apply(x, conv_a.weight, conv_b.weight, 1 if small else 3, packs, planes_of(x) if small else None)


def _foreign_hooks(m):
    """the module carries a forward (pre-)hook that may look at a spike tensor, i.e. one that is not marked ``_eas_planes_safe``"""
    if m._forward_pre_hooks:
        return True
    return any(not getattr(h, '_eas_planes_safe', False) for h in m._forward_hooks.values()) if m._forward_hooks else False


class packed_weights:
    """``with ops.packed_weights(model):`` around a model forward: packs all convolution weights once (see
    prepack_conv_weights) and makes exactly that packing visible to ``conv2d`` for the duration of the block.  Outside the
    block (a sub-module called on its own, a later call after the weights changed) ``conv2d`` packs the weight it is given,
    so a stale packing can never be used."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.prev = (opctx.pack_scope, opctx.planes_scope)
        opctx.pack_scope = prepack_conv_weights(self.model)
        mods = getattr(self.model, '_eas_modules', None)
        if mods is None:
            mods = list(self.model.modules())
            object.__setattr__(self.model, '_eas_modules', mods)
        # a forward hook anywhere in the model may look at a spike tensor: fp32 spikes throughout then (see "SPIKE PLANES")
        # (hooks that declare ``_eas_planes_safe`` -- the trainer's backward cut, which handles ghosts -- do not count)
        opctx.planes_scope = opctx.pack_scope is not None and not any(_foreign_hooks(m) for m in mods)
        self.prev_invstd = opctx.invstd_scope
        if not self.model.training:
            frozen = opctx.frozen is not None and opctx.frozen['model'] is self.model and not torch.is_grad_enabled()
            opctx.invstd_scope = opctx.frozen['inv'] if frozen else _invstd_of_eval_model(mods)

    def __exit__(self, *exc):
        opctx.invstd_scope = self.prev_invstd
        opctx.pack_scope, opctx.planes_scope = self.prev
        if self.prev[0] is None:
            clear_conv_stats()


# Statistics tap (eas_snn_amd/stats.py): when set, every convolution input of the model is shown to it before the
# convolution runs -- ``sink(conv_module, x, replicas)`` for the dense convolutions, and the sampler appends the spike
# tensor entering each micro-step to ``sink.sampler_spikes``.  The kernels that run are the same with or without it.



def set_conv_sink(sink):
    prev, opctx.conv_sink = opctx.conv_sink, sink
    return prev


def conv_sink():
    return opctx.conv_sink


def spike_sop(x, ksize, stride, cout):
    """(sum x, conv(x, ones).sum()) of a convolution input x [NI,Cin,H,W] as a device float64 tensor of 2 (eas_spike_sop)."""
    _dev(x)
    x = _f32c(dense(x))
    NI, Cin, H, W = x.shape
    L = _lib.lib()
    out = torch.empty(2, dtype=torch.float64, device=x.device)
    ws = torch.empty(L.eas_spike_sop_workspace_doubles(), dtype=torch.float64, device=x.device)
    _call('eas_spike_sop', 4 * x.numel(), L.eas_spike_sop, ptr(x), NI, Cin, H, W, int(ksize), int(stride), int(cout), ptr(out), ptr(ws),
          stream())
    return out


def conv_act_eval(x, conv, act='silu'):
    """``act(conv(x))`` for an nn.Conv2d with bias in ONE kernel (eas_conv_fwd_act), no autograd: the eval-mode forward of a real-valued
    BaseConv whose BatchNorm ``fuse_model`` folded into the convolution (network_blocks.py:55-56).  None when the layer is not eligible
    (the caller then runs convolution and activation separately)."""
    if (act != 'silu' or torch.is_grad_enabled() or opctx.conv_sink is not None or conv._forward_hooks or conv._forward_pre_hooks
            or torch.nn.modules.module._global_forward_hooks or not conv_eligible(x, conv)):
        return None
    k, s_ = conv.kernel_size[0], conv.stride[0]
    small = is_small_int(x)
    xd = dense(x)
    xt = 1 if small else 3
    if not conv_fwd_supported(xd.shape[0], xd.shape[1], conv.out_channels, xd.shape[2], xd.shape[3], k, s_, xt):
        return None
    _verify_tags(xd, small)
    packs = getattr(conv, '_eas_packs', None)
    pk = packs[0] if (packs is not None and opctx.pack_scope is not None and packs.get('gen') == opctx.pack_scope) else conv_pack_weights(conv.weight, 0)
    return conv_fwd_packed(xd, pk, conv.bias, conv.out_channels, k, s_, xt, act=1)


# ------------------------------------------------------------------------------------------------ fused real-valued eval block (conv -> BN -> SiLU, ONE kernel)
# The eval-mode forward of an unconverted BaseConv -- the ANN PAFPN neck and head of use_spike=True models -- normalises with running
# statistics, so the BatchNorm and the activation can run on the convolution's accumulators: eas_conv_bn_act_eval, same arithmetic as
# eas_conv_fwd + eas_bn_silu_fwd_ex (bit-identical), one launch and 4 instead of 12 bytes per output element.  Unlike ``fuse_model`` nothing
# is folded into the weights, and the dual convolutions / in-place concatenations of the unfused model stay.  EAS_FUSED_ANN_EVAL=0: the
# two-kernel path (development).



def _ann_eval_bn_ok(bn):
    return (type(bn) is torch.nn.BatchNorm2d and not bn.training and bn.affine and bn.running_mean is not None and bn.running_var is not None
            and bn.num_features % 8 == 0 and not bn._forward_hooks and not bn._forward_pre_hooks)


def fused_ann_eval_ok(x, conv, norms):
    """``conv`` (an nn.Conv2d, or a pair that reads the same input and runs as one) followed by ``norms`` on consecutive output-channel
    ranges and SiLU can run as eas_conv_bn_act_eval on ``x`` [NI,Cin,H,W].  norms[i]: the eval-mode BatchNorm2d behind a bias-free
    convolution, or -- for a block whose BatchNorm ``fuse_model`` folded into the convolution -- that convolution's bias tensor."""
    if not opctx.fused_ann_eval or torch.is_grad_enabled() or opctx.conv_sink is not None or torch.nn.modules.module._global_forward_hooks:
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return False
    convs = conv if isinstance(conv, tuple) else (conv,)
    if len(convs) != len(norms) or len(convs) > 2:
        return False
    for c, nm in zip(convs, norms):
        if type(c) is not torch.nn.Conv2d or c._forward_hooks or c._forward_pre_hooks or not _static_conv_ok(c) or c.out_channels % 8:
            return False
        if torch.is_tensor(nm):
            if nm is not c.bias or not nm.is_cuda or nm.dtype != torch.float32:
                return False
        elif c.bias is not None or not _ann_eval_bn_ok(nm) or c.out_channels != nm.num_features:
            return False
    c0 = convs[0]
    if len(convs) == 2 and not (convs[1].kernel_size == c0.kernel_size and convs[1].stride == c0.stride == (1, 1)
                                and convs[1].in_channels == c0.in_channels):
        return False
    if not conv_eligible(x, c0):
        return False
    if is_small_int(x):
        return False        # the epilogue is instantiated for real-valued inputs (three bf16 terms): what the ANN neck / head read
    cout = sum(c.out_channels for c in convs)
    return conv_fwd_supported(x.shape[0], x.shape[1], cout, x.shape[2], x.shape[3], c0.kernel_size[0], c0.stride[0], 3)


_UNIT_AFFINE = {}


def _unit_affine(C_, device):
    """(ones, zeros) [C]: the BatchNorm constants that make the epilogue compute acc * 1 + bias (a folded block)"""
    key = (int(C_), str(device))
    t = _UNIT_AFFINE.get(key)
    if t is None:
        t = _UNIT_AFFINE[key] = (torch.ones(C_, dtype=torch.float32, device=device), torch.zeros(C_, dtype=torch.float32, device=device))
    return t


def conv_bn_act_eval(x, conv, bns, cats=None, packs=None, act='silu'):
    """act(bn(conv(x))) in one kernel (caller checked ``fused_ann_eval_ok``).  conv / bns: one convolution and its BatchNorm, or two that read
    the same input (``packs``: the packing of their concatenated weight made by ``packed_weights``, or None); a bias tensor in place of a
    BatchNorm = a block folded by ``fuse_model`` (act(conv(x) + bias)).  cats[i] = (buffer [NI,Ctot,H,W], first channel) or None.  Returns one
    tensor per range (a view into the buffer where given)."""
    L = _lib.lib()
    convs = conv if isinstance(conv, tuple) else (conv,)
    cats = cats or [None] * len(bns)
    c0 = convs[0]
    k, stride = c0.kernel_size[0], c0.stride[0]
    Cout = sum(c.out_channels for c in convs)
    NI, Cin, H, W = x.shape
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if len(convs) == 2:
        pk = packs[0] if packs else conv_pack_weights(torch.cat([convs[0].weight, convs[1].weight], 0), 0)
    else:
        pk = getattr(c0, '_eas_packs', None)
        pk = pk[0] if (pk is not None and opctx.pack_scope is not None and pk.get('gen') == opctx.pack_scope) else conv_pack_weights(c0.weight, 0)
    d = _lib.EasConvBnActEval()
    xd = _f32c(x)
    keep = [pk, xd]
    d.x, d.x_terms = ptr(xd), 3
    d.packed_w = ptr(pk)
    d.NI, d.Cin, d.Cout, d.Hi, d.Wi, d.ksize, d.stride = NI, Cin, Cout, H, W, k, stride
    d.act = 1 if act == 'silu' else 0
    d.csplit = convs[0].out_channels
    outs = []
    for i, bn in enumerate(bns):
        r = d.range[i]
        Cr = convs[i].out_channels if len(convs) > 1 else Cout
        if torch.is_tensor(bn):          # folded block: z = fma(acc, 1 * 1, bias - 0 * 1) = acc + bias
            ones, zeros = _unit_affine(Cr, x.device)
            r.gamma, r.beta, r.mean, r.invstd = ptr(ones), ptr(bn), ptr(zeros), ptr(ones)
        else:
            invstd = _eval_invstd(bn.running_var, bn.eps)
            keep.append(invstd)
            r.gamma, r.beta, r.mean, r.invstd = ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(invstd)
        cat = cats[i]
        if cat is not None:
            buf, c0_ = cat[0], int(cat[1])
            assert buf.is_contiguous() and buf.dtype == torch.float32 and tuple(buf.shape) == (NI, buf.shape[1], Ho, Wo)
            out = buf.narrow(1, c0_, Cr)
            r.out, r.out_ctot, r.out_c0 = ptr(buf), buf.shape[1], c0_
        else:
            out = torch.empty((NI, Cr, Ho, Wo), dtype=torch.float32, device=x.device)
            r.out, r.out_ctot, r.out_c0 = ptr(out), Cr, 0
        outs.append(out)
    fl = 2.0 * NI * Cout * Ho * Wo * Cin * k * k
    nb = 4 * NI * Cin * H * W + 4 * NI * Cout * Ho * Wo
    _call('eas_conv_fwd', nb, L.eas_conv_bn_act_eval, C.byref(d), None, stream(), flops=fl, issue_flops=fl * 6)
    del keep
    return outs if len(bns) > 1 else outs[0]


def conv2d(x, conv, small_int=None):
    """``conv(x)`` for an ``nn.Conv2d`` on the matrix-core kernels where eligible (else ATen/MIOpen).
    A module that carries forward hooks (RecordHook in energy_estimation, event_evaluator.py:519-523; thop in
    get_model_info) is called the ordinary way so that the hooks fire."""
    if opctx.conv_sink is not None:
        x = dense(x)
        opctx.conv_sink(conv, x, opctx.replicas)
    if conv._forward_hooks or conv._forward_pre_hooks or torch.nn.modules.module._global_forward_hooks:
        return conv(dense(x))
    if not conv_eligible(x, conv):
        x = dense(x)
        return conv._conv_forward(x, conv.weight, conv.bias)
    if small_int is None:
        small_int = is_small_int(x)
    if planes_of(x) is not None and not (small_int and _planes_conv_ok(x, conv.weight, conv.stride[0])):
        x = dense(x)            # no planes form of this geometry: the fp32 form of the same kernels
    _verify_tags(x, small_int)
    packs = getattr(conv, '_eas_packs', None)
    if packs is not None and (opctx.pack_scope is None or packs.get('gen') != opctx.pack_scope):
        packs = None            # not inside the forward that made this packing: pack the weight as it is now
    return _ConvFn.apply(x, conv.weight, conv.bias, conv.stride[0], 1 if small_int else 3, packs, planes_of(x) if small_int else None)
