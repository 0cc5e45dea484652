"""Host-side operators over the C ABI (include/eas_hip.h): thin ``torch.autograd.Function`` wrappers that
hand raw device pointers and the current HIP stream to libeas_hip.so.  PyTorch is plumbing here (memory,
streams, autograd bookkeeping); all arithmetic of the hot path runs in the HIP kernels.

No CPU path: CPU tensors raise.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, ptr, stream

SURROGATE_IDS = {'atan': 0, 'sigmoid': 1, 'rect': 2, 'patan': 3}
SG_PATAN = 3      # learnable slope: ``alpha`` is a device tensor of one element and receives a gradient
FLAG_HARD_RESET, FLAG_DECAY_INPUT, FLAG_DETACH_RESET, FLAG_FIRE_STRICT = 1, 2, 4, 8
READOUT_IDS = {'sum': 0, 'last': 1, 'avg': 2}

# Write the final membrane potential back after every multi-step call (spikingjelly semantics: ``node.v``
# holds the state until ``reset_net``).  Training/eval loops that reset after every batch (the reference's
# do: yolox/core/trainer.py:115-117, yolox/evaluators/event_evaluator.py:196-198) can switch this off and
# save 4 B per neuron per layer; results are identical because the state is discarded by the reset.
_STATE_WRITEBACK = True


def set_state_writeback(flag):
    global _STATE_WRITEBACK
    _STATE_WRITEBACK = bool(flag)


def state_writeback():
    return _STATE_WRITEBACK


class state_writeback_scope:
    """``with ops.state_writeback_scope(flag):`` -- the setting for the duration of the block, the previous one restored on exit."""

    def __init__(self, flag):
        self.flag = bool(flag)

    def __enter__(self):
        global _STATE_WRITEBACK
        self.prev, _STATE_WRITEBACK = _STATE_WRITEBACK, self.flag
        return self

    def __exit__(self, *exc):
        global _STATE_WRITEBACK
        _STATE_WRITEBACK = self.prev


def no_state_writeback():
    """``with ops.no_state_writeback():`` around a loop whose every iteration ends with ``reset_net`` (the reference's
    train / eval loops): the final membrane potentials are not written to HBM.  The previous setting is restored on exit,
    so stateful use (streaming inference, ``node.v`` inspection) elsewhere in the process is unaffected."""
    return state_writeback_scope(False)


class KernelTimer:
    """Per-call HIP-event timing of the C-ABI entry points (bench.py roofline accounting).  Events are recorded on
    the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.rec = {}
        self.tagged = {}

    def add(self, name, start, end, nbytes, flops=0.0, issue_flops=0.0):
        self.rec.setdefault(name, []).append((start, end, nbytes, flops, issue_flops))
        if _TAG is not None:                      # development: per-layer breakdown (scripts/layer_times.py)
            self.tagged.setdefault((_TAG, name), []).append((start, end, nbytes, flops, issue_flops))

    def summary(self, hbm_gbs=8000.0, mfma_tf=2500.0):
        """per entry point: calls, total ms, algorithmic bytes, algorithmic flops (2 x MAC), matrix-core flops issued
        (algorithmic x number of bf16 term products) and the sum of the launches' roofline times (roof_ms, see below)."""
        out = {}
        for name, items in self.rec.items():
            ms = sum(it[0].elapsed_time(it[1]) for it in items)
            # per-launch roofline: a launch cannot finish before its algorithmic bytes have crossed HBM nor before its bf16 term products
            # have gone through the matrix cores -- the larger of the two, summed over the launches (small-channel layers are HBM-bound,
            # wide ones MFMA-bound; one ceiling for the whole family describes neither)
            roof = sum(max(it[2] / (hbm_gbs * 1e9), it[4] / (mfma_tf * 1e12)) for it in items) * 1e3
            out[name] = dict(calls=len(items), ms=ms, bytes=sum(it[2] for it in items), flops=sum(it[3] for it in items),
                             issue_flops=sum(it[4] for it in items), roof_ms=roof)
        return out


_TIMER = None
_TAG = None          # development: label (layer name, phase) attached to the timed calls, see KernelTimer.tagged


def set_tag(tag):
    global _TAG
    prev, _TAG = _TAG, tag
    return prev


def set_timer(timer):
    global _TIMER
    _TIMER = timer


_CALL_LOG = None     # test infrastructure (``kernel_trace``): list of (C-ABI symbol, argument tuple) of every call made through _call


class kernel_trace:
    """``with ops.kernel_trace() as tr:`` -- afterwards ``tr.kernels`` holds the distinct device kernel symbols the library launched inside
    the block (the names rocprofv3 reports; eas_kernel_trace_begin / _dump) and ``tr.calls`` the C-ABI calls made through the operators
    with their arguments.  Test infrastructure: tests/test_gpu_bench_shapes.py replays every convolution geometry of a bench step against
    fp64 and proves that every kernel instance of the step is one an oracle-compared test has launched."""

    def __enter__(self):
        global _CALL_LOG
        self.prev, _CALL_LOG = _CALL_LOG, []
        self.calls = _CALL_LOG
        _lib.lib().eas_kernel_trace_begin()
        return self

    def __exit__(self, *exc):
        global _CALL_LOG
        _CALL_LOG = self.prev
        L = _lib.lib()
        need = L.eas_kernel_trace_dump(None, 0)
        buf = C.create_string_buffer(int(need))
        L.eas_kernel_trace_dump(buf, need)
        self.kernels = sorted(k for k in buf.value.decode().split('\n') if k)


def _call(name, nbytes, fn, *args, flops=0.0, issue_flops=0.0):
    """Invoke one C-ABI entry point (optionally bracketed by HIP events) and check its status."""
    if _CALL_LOG is not None:
        _CALL_LOG.append((fn.__name__, args))
    if _TIMER is None:
        check(fn(*args), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    check(fn(*args), name)
    e.record()
    _TIMER.add(name, s, e, nbytes, flops, issue_flops)


def _timer_mark():
    """start event of a timed region that is not a single _call (None when no timer is installed)"""
    if _TIMER is None:
        return None
    s = torch.cuda.Event(enable_timing=True)
    s.record()
    return s


def _timer_add(name, start, nbytes):
    if start is not None and _TIMER is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _TIMER.add(name, start, e, nbytes)


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.EasHipError('eas_snn_amd operators run on the GPU only (got a CPU tensor); there is no CPU fallback')


def _alpha_arg(sg_id, alpha):
    """float for the fixed-slope surrogates; the learnable [1] device tensor itself for 'patan' (never read on the host)"""
    if sg_id == SG_PATAN:
        if not torch.is_tensor(alpha) or alpha.numel() != 1 or alpha.dtype != torch.float32:
            raise _lib.EasHipError("surrogate 'patan' needs its learnable alpha as a float32 tensor of one element")
        return alpha
    return float(alpha)


# SPIKE PLANES.  Between the fused layers of the model a spike tensor [T,N,C,H,W] (0/1 spikes, SEW sums: exact in bf16) does not exist as
# fp32 at all: the BN+LIF kernel writes it as bf16 in blocks of 8 channels, planes [T,N,C/8,H*W,8] (include/eas_hip.h "SPIKE PLANES"), the
# form in which every consumer -- 1x1 / 3x3 convolution forward, their weight gradients, the SEW shortcut of the next BN+LIF kernel --
# reads its operand with 16-byte loads and without conversion, at half the HBM bytes.  For autograd the tensor is represented by a GHOST:
# an fp32 tensor of the logical shape that owns no storage (a NaN scalar expanded with stride 0) and carries the planes as ``t._eas_sp``.
# Only code that asks for planes gets them (``planes=True`` of the fused blocks in compat/yolox/models/network_blocks.py, whose consumers are
# the operators below); everything else -- API callers, hooks, statistics taps, kernels that read fp32 NCHW -- goes through ``dense(t)``,
# which unpacks the planes (eas_spike_planes_to_f32).  A ghost that is read by accident yields NaN, never a plausible number.
# EAS_SPIKE_PLANES=0: development switch, fp32 spikes everywhere (the round-2 data path).
SPIKE_PLANES = os.environ.get('EAS_SPIKE_PLANES', '1') == '1'
_GHOST_BASE = {}
_PLANES_SCOPE = False    # inside the forward of a whole model (``packed_weights``) none of whose modules carries a forward hook


def planes_enabled():
    """spike planes may be produced right now: switched on, inside a model's own forward (``packed_weights`` -- a block called on its own
    hands out fp32 spikes), and nothing that wants to see fp32 tensors is attached (module forward hooks, a statistics tap)"""
    return SPIKE_PLANES and _PLANES_SCOPE and _CONV_SINK is None and not torch.nn.modules.module._global_forward_hooks


def planes_of(t):
    """the spike planes [.., C/8, H*W, 8] (bf16) a ghost tensor stands for, or None for an ordinary tensor"""
    return getattr(t, '_eas_sp', None)


def ghost(shape, device, planes=None):
    """fp32 tensor of ``shape`` without storage (every element aliases one NaN); ``planes``: attached as the data it stands for"""
    key = str(device)
    base = _GHOST_BASE.get(key)
    if base is None:
        base = _GHOST_BASE[key] = torch.full((1,), float('nan'), dtype=torch.float32, device=device)
    g = base.expand(tuple(shape))
    if planes is not None:
        g._eas_sp = planes
        g._eas_small_int = True
    return g


def new_planes(T, N, C, H, W, device):
    return torch.empty((T, N, C // 8, H * W, 8), dtype=torch.bfloat16, device=device)


class _DenseFn(torch.autograd.Function):
    """ghost -> real fp32 tensor (eas_spike_planes_to_f32); the gradient passes through unchanged"""

    @staticmethod
    def forward(ctx, g, sp):
        shape = tuple(g.shape)
        C_, H, W = shape[-3:]
        NI = 1
        for d in shape[:-3]:
            NI *= d
        out = torch.empty(shape, dtype=torch.float32, device=sp.device)
        _call('eas_spike_planes', 6 * out.numel(), _lib.lib().eas_spike_planes_to_f32, ptr(sp), 0, ptr(out), 0, NI, C_, H * W, stream())
        return out

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def dense(t):
    """``t`` as a real fp32 tensor: a ghost is unpacked from its planes (one pass, 6 B per element), anything else is returned as it is"""
    sp = planes_of(t)
    if sp is None:
        return t
    out = _DenseFn.apply(t, sp.contiguous())
    mark_small_int(out)
    return out


def to_planes(x):
    """a real fp32 spike tensor [.., C, H, W] (values exact in bf16) as planes (eas_spike_planes_from_f32); for tests and borders"""
    x = _f32c(x)
    shape = tuple(x.shape)
    C_, H, W = shape[-3:]
    NI = x.numel() // (C_ * H * W)
    sp = torch.empty(shape[:-3] + (C_ // 8, H * W, 8), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().eas_spike_planes_from_f32(ptr(x), 0, ptr(sp), 0, NI, C_, H * W, ptr(conv_inexact_flag(x.device)), stream()),
          'eas_spike_planes_from_f32')
    return sp


def fold_time(x_seq):
    """[T, N, ...] -> [T*N, ...] keeping the spike tags (small-integer mark, planes of a ghost)"""
    x = x_seq.flatten(0, 1)
    if is_small_int(x_seq):
        mark_small_int(x)
    sp = planes_of(x_seq)
    if sp is not None:
        x._eas_sp = sp.flatten(0, 1)
    return x


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise _lib.EasHipError(f'expected float32, got {t.dtype}')
    return t.contiguous()


# 1 / sqrt(running_var + eps) of the BatchNorm layers that normalise with their running statistics.  Inside the forward of an eval-mode
# model (``packed_weights``) all of them are computed by two multi-tensor launches when the forward starts instead of two small launches per
# layer (SYOLOX-S: 148 launches, 0.67 ms of a 5.7 ms eval forward at batch 64); same arithmetic (x + eps, rsqrt), so the same bits.  Valid
# for the duration of that forward only -- nothing is cached across forwards (the statistics may change in between).
_INVSTD_SCOPE = None      # {address of running_var: (eps, invstd)} or None


def _eval_invstd(running_var, eps):
    if _INVSTD_SCOPE is not None:
        hit = _INVSTD_SCOPE.get(running_var.data_ptr())
        if hit is not None and hit[0] == float(eps):
            return hit[1]
    return torch.rsqrt(running_var + eps)


def _invstd_of_eval_model(mods):
    bns = [m for m in mods if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.running_var is not None and m.running_var.is_cuda
           and not m.training]
    if not bns:
        return None
    inv = torch._foreach_rsqrt(torch._foreach_add([m.running_var for m in bns], [float(m.eps) for m in bns]))
    return {m.running_var.data_ptr(): (float(m.eps), t) for m, t in zip(bns, inv)}


# ------------------------------------------------------------------------------------------------ K2
class _LIFFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, v_in, w, k_const, v_th, v_reset, flags, sg_id, alpha, want_mean, write_v):
        ctx.set_materialize_grads(False)      # a result nobody differentiates arrives as None in backward, not as a zero tensor
        _dev(x, v_in, w)
        L = _lib.lib()
        x = _f32c(x)
        v_in = _f32c(v_in)
        T, M = x.shape[0], x[0].numel()
        need_grad = ctx.needs_input_grad[0] or (w is not None and ctx.needs_input_grad[2])
        spikes = torch.empty_like(x)
        h = torch.empty_like(x) if need_grad else None
        v_out = torch.empty_like(x[0]) if write_v else None
        mean = torch.empty_like(x[0]) if want_mean else None
        _call('eas_lif_fwd', (8 + (4 if h is not None else 0)) * T * M, L.eas_lif_fwd, ptr(x), ptr(v_in), ptr(v_out), ptr(w),
              k_const, v_th, v_reset, flags, ptr(spikes), ptr(h), ptr(mean), T, M, stream())
        keep_x = x if (flags & FLAG_DECAY_INPUT) and w is not None else None
        learn = sg_id == SG_PATAN
        if learn:
            _dev(alpha)
        ctx.save_for_backward(h, v_in, w, keep_x, alpha if learn else None)
        ctx.cfg = (k_const, v_th, v_reset, flags, sg_id, None if learn else alpha, T, M)
        if v_out is not None:
            ctx.mark_non_differentiable(v_out)
        return spikes, v_out, mean

    @staticmethod
    def backward(ctx, g_s, g_v, g_mean):
        h, v_in, w, x, alpha_t = ctx.saved_tensors
        k_const, v_th, v_reset, flags, sg_id, alpha, T, M = ctx.cfg
        L = _lib.lib()
        if g_s is None and g_mean is None:
            return (torch.zeros_like(h),) + (None,) * 10
        g_s = _f32c(g_s)
        g_mean = _f32c(g_mean)
        gx = torch.empty_like(h)
        want_w = w is not None and ctx.needs_input_grad[2]
        gw = torch.empty_like(w) if want_w else None
        want_a = alpha_t is not None and ctx.needs_input_grad[8]
        ga = torch.empty_like(alpha_t) if want_a else None
        ws = torch.empty(L.eas_reduce_workspace_floats(M), dtype=torch.float32, device=h.device) if (want_w or want_a) else None
        if alpha_t is not None:
            _call('eas_lif_bwd', 12 * T * M, L.eas_lif_bwd_patan, ptr(g_s), ptr(g_mean), ptr(h), ptr(v_in), ptr(x), ptr(w), k_const,
                  v_th, v_reset, flags, ptr(alpha_t), ptr(ga), ptr(gx), ptr(gw), ptr(ws), T, M, stream())
        else:
            _call('eas_lif_bwd', 12 * T * M, L.eas_lif_bwd, ptr(g_s), ptr(g_mean), ptr(h), ptr(v_in), ptr(x), ptr(w), k_const,
                  v_th, v_reset, flags, sg_id, alpha, ptr(gx), ptr(gw), ptr(ws), T, M, stream())
        return gx, None, gw, None, None, None, None, None, ga, None, None


def lif_multistep(x_seq, v_in, w, k_const, v_th, v_reset, flags, surrogate, alpha, want_mean=False, write_v=None):
    """Multi-step (P)LIF over x_seq [T, ...].  Returns (spikes, v_final|None, mean_over_T|None)."""
    if write_v is None:
        write_v = _STATE_WRITEBACK
    sg_id = SURROGATE_IDS[surrogate] if isinstance(surrogate, str) else int(surrogate)
    return _LIFFn.apply(x_seq, v_in, w, float(k_const), float(v_th), float(v_reset), int(flags), sg_id, _alpha_arg(sg_id, alpha),
                        bool(want_mean), bool(write_v))


def time_mean(x_seq):
    """[T, ...] -> mean over T (firing-rate readout), differentiable."""
    return _TimeMeanFn.apply(dense(x_seq))


class _TimeMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _f32c(x)
        out = torch.empty_like(x[0])
        ctx.T = x.shape[0]
        check(_lib.lib().eas_time_mean(ptr(x), ptr(out), x.shape[0], out.numel(), stream()), 'eas_time_mean')
        return out

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.T).unsqueeze(0).expand(ctx.T, *g.shape)


# ------------------------------------------------------------------------------------------------ K4 (BN + LIF)
# Convolution -> BatchNorm hand-over (eas_conv_fwd_stats, north_star's fused conv -> BN -> LIF step): inside ``conv_stats_scope`` a
# matrix-core convolution also leaves the per-channel sums of its output tile by tile, and the BN kernel that consumes exactly that
# tensor next adds them up instead of reading y once more (no eas_bn_stats_partial launch).  The slot holds the convolution output
# itself, so its address cannot be reused while the slot is valid; a BN call on anything else falls back to the statistics pass.
CONV_STATS = os.environ.get('EAS_CONV_STATS', '1') == '1'
CONV_STATS_MAX_BLOCKS = int(os.environ.get('EAS_CONV_STATS_MAX_BLOCKS', '4096'))     # partials per channel the consumers still add cheaply
_WANT_CONV_STATS = False
_CONV_STATS_SLOT = None          # (y [NI,Cout,Ho,Wo], nb, stats [Cout*nb*2] fp64, y._version)
_STATS_BLOCKS = {}


class conv_stats_scope:
    """``with ops.conv_stats_scope(bn_uses_batch_statistics):`` around the convolution whose output goes straight into a fused BN kernel"""

    def __init__(self, flag=True):
        self.flag = bool(flag) and CONV_STATS

    def __enter__(self):
        global _WANT_CONV_STATS
        self.prev, _WANT_CONV_STATS = _WANT_CONV_STATS, self.flag

    def __exit__(self, *exc):
        global _WANT_CONV_STATS
        _WANT_CONV_STATS = self.prev


def _conv_stats_blocks(L, geom):
    nb = _STATS_BLOCKS.get(geom)
    if nb is None:
        nb = _STATS_BLOCKS[geom] = L.eas_conv_fwd_stats_blocks(*geom)
    return nb


def _take_conv_stats(base_ptr, count, Ctot, keep=False):
    """(stats tensor, nb) if the tensor at ``base_ptr`` ([.., Ctot, H, W] with ``count`` elements per channel) is the output of the last
    convolution that summed its tiles, else None; the slot is cleared unless ``keep`` (two consumers of one output)."""
    global _CONV_STATS_SLOT
    slot = _CONV_STATS_SLOT
    if slot is None:
        return None
    y, nb, stats, ver = slot
    if not keep:
        _CONV_STATS_SLOT = None
    if y.data_ptr() != base_ptr or y.shape[1] != Ctot or y.shape[0] * y.shape[2] * y.shape[3] != count or y._version != ver:
        return None                    # another tensor, or y was modified in place since the convolution summed it
    return stats, nb


def clear_conv_stats():
    """drop a convolution's tile sums nobody consumed (end of a model forward: the slot must not pin tensors across steps)"""
    global _CONV_STATS_SLOT
    _CONV_STATS_SLOT = None


def _pending_stats(L, y, TN, Cc, HW, replicas, eps, momentum, running_mean, running_var, dev, y_ptr=None, y_ctot=0, keep_slot=False):
    """batch statistics of the consumer kernel: the producing convolution's tile sums (conv_stats_scope) or the partial-sum launch whose
    result the consumer finalizes (EasBnPending)."""
    got = _take_conv_stats(y.data_ptr(), TN * HW, y_ctot if y_ctot else Cc, keep_slot)
    if got is not None:
        stats, nb = got
        c0 = 0 if y_ptr is None else (y_ptr - y.data_ptr()) // (4 * HW)
        pend = _lib.EasBnPending(stats.data_ptr() + 16 * c0 * nb, nb, int(replicas), float(TN) * HW, float(eps),
                                 float(momentum if momentum is not None else 0.0), ptr(running_mean) if momentum is not None else None,
                                 ptr(running_var) if momentum is not None else None, nb)
        return pend, stats, 0                # no launch: the consumer adds the convolution's tile sums
    ws = torch.empty(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=dev)
    if _CALL_LOG is not None:
        _CALL_LOG.append(('eas_bn_stats_partial', (None, y_ctot, TN, Cc, HW, int(replicas))))
    chunks = L.eas_bn_stats_partial(ptr(y) if y_ptr is None else y_ptr, y_ctot, TN, Cc, HW, ptr(ws), stream())
    if chunks <= 0:
        check(chunks if chunks < 0 else -1, 'eas_bn_stats_partial')
    pend = _lib.EasBnPending(ptr(ws), chunks, int(replicas), float(TN) * HW, float(eps), float(momentum if momentum is not None else 0.0),
                             ptr(running_mean) if momentum is not None else None, ptr(running_var) if momentum is not None else None, 0)
    return pend, ws, 4 * TN * Cc * HW        # the statistics launch read y once


def _channel_slice_of(g, Cc):
    """total channel count if ``g`` [T,N,C,H,W] is a channel slice of a contiguous wider tensor (what the backward of an
    in-place concatenation hands out), else 0."""
    if g.is_contiguous():
        return Cc
    T, N, C_, H, W = g.shape
    st = g.stride()
    if st[4] == 1 and st[3] == W and st[2] == H * W and st[1] % (H * W) == 0 and st[1] // (H * W) > C_ and st[0] == N * st[1] \
            and (g.data_ptr() % 16) == 0:
        return st[1] // (H * W)
    return 0


class _BNLIFFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, bn_state, v_in, w, k_const, v_th, v_reset, flags, sg_id, alpha, want_mean, write_v, t_bcast,
                residual, cat_buf, cat_c0, sp_arg, residual_sp):
        """y: [T,N,C,H,W], or [N,C,H,W] standing for ``t_bcast`` identical time steps.  residual [T,N,C,H,W]: the output is
        spikes + residual (SEW shortcut).  cat_buf [T,N,Ctot,H,W]: the output is written as channels cat_c0.. of it and
        returned as a view (concatenation in place).
        sp_arg: None = fp32 output; True = the output as spike planes (fourth result; the first result is then a ghost, see ``ghost``);
        a planes tensor [T,N,Ctot/8,HW,8] = the planes of ``cat_buf`` (itself a ghost): written into its groups cat_c0/8.. .
        residual_sp: the planes of ``residual`` when that is a ghost."""
        ctx.set_materialize_grads(False)      # a result nobody differentiates arrives as None in backward, not as a zero tensor
        running_mean, running_var, use_batch_stats, momentum, eps = bn_state
        _dev(y, gamma, beta, v_in, w)
        L = _lib.lib()
        y = _f32c(y)
        v_in = _f32c(v_in)
        if t_bcast:
            T, (N, Cc) = int(t_bcast), y.shape[:2]
            plane = y.shape
        else:
            T, N, Cc = y.shape[0], y.shape[1], y.shape[2]
            plane = y.shape[1:]
        HW = plane[-1] * plane[-2]
        dev = y.device
        pend = keep = None
        if use_batch_stats:
            mean = torch.empty(Cc, dtype=torch.float32, device=dev)
            invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
            t0 = _timer_mark()
            pend, keep, nb_ = _pending_stats(L, y, N if t_bcast else T * N, Cc, HW, T if t_bcast else 1, eps, momentum, running_mean,
                                             running_var, dev)
            if nb_:
                _timer_add('eas_bn_stats', t0, nb_)
        else:
            mean = running_mean
            invstd = _eval_invstd(running_var, eps)
        planes = sp_arg is not None and sp_arg is not False
        res_ctot = 0
        if residual is not None:
            assert not want_mean and residual.shape == (T,) + tuple(plane)
            if residual_sp is not None:
                assert planes and residual_sp.is_contiguous() and residual_sp.shape == (T, N, Cc // 8, HW, 8)
            else:
                assert not planes
                residual = _f32c(residual)
        sp = None
        if planes:
            if torch.is_tensor(sp_arg):
                assert cat_buf is not None and sp_arg.dtype == torch.bfloat16 and sp_arg.is_contiguous() and cat_c0 % 8 == 0 and \
                    sp_arg.shape == (T, N, cat_buf.shape[2] // 8, HW, 8)
                sp = sp_arg.narrow(2, cat_c0 // 8, Cc // 8)
                spikes = cat_buf.narrow(2, cat_c0, Cc)
                ctot = cat_buf.shape[2]
            else:
                assert cat_buf is None
                sp = new_planes(T, N, Cc, plane[-2], plane[-1], dev)
                spikes = ghost((T,) + tuple(plane), dev)
                ctot = 0
        elif cat_buf is not None:
            assert cat_buf.is_contiguous() and cat_buf.shape[:2] == (T, N) and cat_buf.shape[3:] == tuple(plane[2:])
            spikes = cat_buf.narrow(2, cat_c0, Cc)
            ctot = cat_buf.shape[2]
        else:
            spikes = torch.empty((T,) + tuple(plane), dtype=torch.float32, device=dev)
            ctot = 0
        v_out = torch.empty(plane, dtype=torch.float32, device=dev) if write_v else None
        mo = torch.empty(plane, dtype=torch.float32, device=dev) if want_mean else None
        nsteps = T * N * Cc * HW
        _call('eas_bn_lif_fwd', (4 * y.numel() + (2 if planes else 4) * nsteps), L.eas_bn_lif_fwd_ex, ptr(y), 0, ptr(mean),
              ptr(invstd), ptr(gamma), ptr(beta), ptr(v_in), ptr(v_out), ptr(w), k_const, v_th, v_reset, flags, None if planes else ptr(spikes), ptr(mo),
              T, N, Cc, HW, int(bool(t_bcast)), C.byref(pend) if pend is not None else None, None if residual_sp is not None else ptr(residual), ctot,
              ptr(sp), ptr(residual_sp), res_ctot, stream())
        del keep
        learn = sg_id == SG_PATAN
        if learn:
            _dev(alpha)
        ctx.save_for_backward(y, mean, invstd, gamma, beta, v_in, w, alpha if learn else None)
        ctx.cfg = (k_const, v_th, v_reset, flags, sg_id, None if learn else alpha, bool(use_batch_stats), T, N, Cc, HW, bool(t_bcast))
        ctx.has_residual = residual is not None
        if v_out is not None:
            ctx.mark_non_differentiable(v_out)
        if sp is not None:
            ctx.mark_non_differentiable(sp)
        return spikes, v_out, mo, sp

    @staticmethod
    def backward(ctx, g_s, g_v, g_mean, _g_sp):
        y, mean, invstd, gamma, beta, v_in, w, alpha_t = ctx.saved_tensors
        k_const, v_th, v_reset, flags, sg_id, alpha, batch_stats, T, N, Cc, HW, bcast = ctx.cfg
        L = _lib.lib()
        nout = 20
        if g_s is None and g_mean is None:
            return (torch.zeros_like(y),) + (None,) * (nout - 1)
        g_res = g_s if ctx.has_residual else None          # d(spikes + residual)/d residual = identity: the same tensor, no copy
        ctot = 0
        if g_s is not None:
            ctot = _channel_slice_of(g_s, Cc) if g_s.dim() == 5 and g_s.dtype == torch.float32 else 0
            if ctot == 0:
                g_s = _f32c(g_s)
        g_mean = _f32c(g_mean)
        gy = torch.empty_like(y)
        ggamma = torch.empty_like(gamma)
        gbeta = torch.empty_like(beta)
        want_w = w is not None and ctx.needs_input_grad[5]
        gw = torch.empty_like(w) if want_w else None
        ws = torch.empty(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=y.device)
        nsteps = T * N * Cc * HW
        nbytes = 4 * (2 * nsteps + gy.numel()) if not bcast else 4 * (nsteps + 2 * gy.numel())
        ga = None
        if alpha_t is not None:
            ga = torch.empty_like(alpha_t) if ctx.needs_input_grad[11] else None
            _call('eas_bn_lif_bwd', nbytes, L.eas_bn_lif_bwd_patan, ptr(g_s), ctot, ptr(g_mean), ptr(y), 0, ptr(mean), ptr(invstd), ptr(gamma),
                  ptr(beta), ptr(v_in), ptr(w), k_const, v_th, v_reset, flags, ptr(alpha_t), ptr(ga), int(batch_stats), ptr(gy), ptr(ggamma),
                  ptr(gbeta), ptr(gw), ptr(ws), T, N, Cc, HW, int(bcast), stream())
        else:
            _call('eas_bn_lif_bwd', nbytes, L.eas_bn_lif_bwd_ex, ptr(g_s), ctot, ptr(g_mean), ptr(y), 0, ptr(mean), ptr(invstd), ptr(gamma),
                  ptr(beta), ptr(v_in), ptr(w), k_const, v_th, v_reset, flags, sg_id, alpha, int(batch_stats), ptr(gy), ptr(ggamma),
                  ptr(gbeta), ptr(gw), ptr(ws), T, N, Cc, HW, int(bcast), stream())
        return (gy, ggamma, gbeta, None, None, gw) + (None,) * 5 + (ga,) + (None,) * 3 + (g_res, None, None, None, None)


class _JoinFn(torch.autograd.Function):
    """The tensor whose channel slices were written in place by the producers of ``parts`` (concatenation without a copy):
    forward hands out ``buf`` itself, backward hands each producer its channel slice of the gradient as a view."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.sizes = [p.shape[-3] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, c = [], 0
        for n in ctx.sizes:
            outs.append(g.narrow(-3, c, n))
            c += n
        return (None,) + tuple(outs)


def join_channels(buf, *parts, sp_buf=None):
    """sp_buf: the spike planes of the whole concatenation (``buf`` is then a ghost; the producers of ``parts`` wrote their channel groups)"""
    out = _JoinFn.apply(buf, *parts)
    if all(is_small_int(p) for p in parts):
        mark_small_int(out)
        if sp_buf is not None:
            out._eas_sp = sp_buf
    return out


class _BNLIF2Fn(torch.autograd.Function):
    """Two BN+LIF layers on the two channel ranges of ONE convolution output y12 [T,N,Ca+Cb,H,W] (the 1x1 branches conv1 /
    conv2 of a CSPLayer computed by one convolution with concatenated weights): each reads its channel slice in place and
    the backward writes both slices of ONE gradient tensor, so the convolution's input gradient needs no addition of two
    branch gradients and its input is read once."""

    @staticmethod
    def forward(ctx, y12, gamma_a, beta_a, w_a, gamma_b, beta_b, w_b, cfg_a, cfg_b):
        ctx.set_materialize_grads(False)      # a result nobody differentiates arrives as None in backward, not as a zero tensor
        L = _lib.lib()
        y12 = _f32c(y12)
        T, N, Ct, H, W = y12.shape
        HW = H * W
        dev = y12.device
        outs, saved, cfgs = [], [], []
        c0 = 0
        for gamma, beta, w, cfg in ((gamma_a, beta_a, w_a, cfg_a), (gamma_b, beta_b, w_b, cfg_b)):
            (running_mean, running_var, use_batch_stats, momentum, eps), v_in, k_const, v_th, v_reset, flags, sg_id, alpha, write_v, cat, Cc = cfg[:11]
            want_sp = cfg[11] if len(cfg) > 11 else False       # the output as spike planes (cat = (ghost buffer, first channel, its planes))
            _dev(gamma, beta, v_in, w)
            v_in = _f32c(v_in)
            yp = y12.data_ptr() + 4 * c0 * HW
            pend = keep = None
            if use_batch_stats:
                mean = torch.empty(Cc, dtype=torch.float32, device=dev)
                invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
                t0 = _timer_mark()
                pend, keep, nb_ = _pending_stats(L, y12, T * N, Cc, HW, 1, eps, momentum, running_mean, running_var, dev, y_ptr=yp, y_ctot=Ct,
                                                 keep_slot=c0 == 0)
                if nb_:
                    _timer_add('eas_bn_stats', t0, nb_)
            else:
                mean = running_mean
                invstd = _eval_invstd(running_var, eps)
            sp = None
            if cat is not None:
                spikes = cat[0].narrow(2, cat[1], Cc)
                ctot = cat[0].shape[2]
                if want_sp:
                    assert len(cat) > 2 and cat[2] is not None and cat[1] % 8 == 0 and cat[2].is_contiguous()
                    sp = cat[2].narrow(2, cat[1] // 8, Cc // 8)
            elif want_sp:
                sp = new_planes(T, N, Cc, H, W, dev)
                spikes = ghost((T, N, Cc, H, W), dev)
                ctot = 0
            else:
                spikes = torch.empty((T, N, Cc, H, W), dtype=torch.float32, device=dev)
                ctot = 0
            v_out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev) if write_v else None
            _call('eas_bn_lif_fwd', (4 + (2 if sp is not None else 4)) * T * N * Cc * HW, L.eas_bn_lif_fwd_ex, yp, Ct, ptr(mean), ptr(invstd),
                  ptr(gamma), ptr(beta), ptr(v_in), ptr(v_out), ptr(w), k_const, v_th, v_reset, flags, None if sp is not None else ptr(spikes), None, T, N,
                  Cc, HW, 0, C.byref(pend) if pend is not None else None, None, ctot, ptr(sp), None, 0, stream())
            del keep
            outs += [spikes, v_out, sp]
            saved += [mean, invstd, gamma, beta, v_in, w]
            cfgs.append((k_const, v_th, v_reset, flags, sg_id, alpha, bool(use_batch_stats), Cc, c0))
            c0 += Cc
        ctx.save_for_backward(y12, *saved)
        ctx.cfgs = cfgs
        ctx.dims = (T, N, Ct, HW)
        for v in (outs[1], outs[2], outs[4], outs[5]):
            if v is not None:
                ctx.mark_non_differentiable(v)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_sa, g_va, _g_ua, g_sb, g_vb, _g_ub):
        y12, *saved = ctx.saved_tensors
        T, N, Ct, HW = ctx.dims
        L = _lib.lib()
        gy12 = torch.empty_like(y12)
        res = []
        for i, g_s in enumerate((g_sa, g_sb)):
            mean, invstd, gamma, beta, v_in, w = saved[6 * i:6 * i + 6]
            k_const, v_th, v_reset, flags, sg_id, alpha, batch_stats, Cc, c0 = ctx.cfgs[i]
            ggamma, gbeta = torch.empty_like(gamma), torch.empty_like(beta)
            gw = torch.empty_like(w) if w is not None else None
            if g_s is None:
                gy12.narrow(2, c0, Cc).zero_()
                ggamma.zero_(); gbeta.zero_()
                if gw is not None:
                    gw.zero_()
            else:
                ctot = _channel_slice_of(g_s, Cc) if g_s.dtype == torch.float32 else 0
                if ctot == 0:
                    g_s = _f32c(g_s)
                ws = torch.empty(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=y12.device)
                off = 4 * c0 * HW
                _call('eas_bn_lif_bwd', 12 * T * N * Cc * HW, L.eas_bn_lif_bwd_ex, ptr(g_s), ctot, None, y12.data_ptr() + off, Ct, ptr(mean),
                      ptr(invstd), ptr(gamma), ptr(beta), ptr(v_in), ptr(w), k_const, v_th, v_reset, flags, sg_id, alpha, int(batch_stats),
                      gy12.data_ptr() + off, ptr(ggamma), ptr(gbeta), ptr(gw), ptr(ws), T, N, Cc, HW, 0, stream())
            res += [ggamma, gbeta, gw]
        return (gy12,) + tuple(res) + (None, None)


def bn_lif_pair(y12, a, b):
    """a / b: (gamma, beta, w, cfg) of the two layers, cfg = (bn_state, v_in, k_const, v_th, v_reset, flags, surrogate id, alpha,
    write_v, cat (buffer, first channel[, planes of the buffer]) or None, channels[, output as spike planes]).  Returns (spikes_a, v_a,
    spikes_b, v_b); with planes the spike tensors are ghosts that carry them (``planes_of``)."""
    sa, va, pa, sb, vb, pb = _BNLIF2Fn.apply(y12, a[0], a[1], a[2], b[0], b[1], b[2], a[3], b[3])
    if pa is not None:
        sa._eas_sp = pa
    if pb is not None:
        sb._eas_sp = pb
    return sa, va, sb, vb


def conv2d_weight(x, weight, stride=1, small_int=None):
    """functional form of ``conv2d`` for a weight tensor that is not a module parameter (e.g. two concatenated 1x1 weights)"""
    if small_int is None:
        small_int = is_small_int(x)
    if planes_of(x) is not None and not (small_int and _planes_conv_ok(x, weight, stride)):
        x = dense(x)
    _verify_tags(x, small_int)
    return _ConvFn.apply(x, weight, None, stride, 1 if small_int else 3, None, planes_of(x) if small_int else None)


def bn_lif_supported(y_seq, T):
    return y_seq.dim() == 5 and T <= 8 and (y_seq.shape[-1] * y_seq.shape[-2]) % 4 == 0


def bn_lif_multistep(y_seq, gamma, beta, running_mean, running_var, use_batch_stats, momentum, eps, v_in, w, k_const,
                     v_th, v_reset, flags, surrogate, alpha, want_mean=False, write_v=None, t_bcast=0, residual=None, cat=None,
                     planes=False):
    """Fused BatchNorm(step_mode='m') + multi-step LIF on the conv output y_seq [T,N,C,H,W]
    (or one plane [N,C,H,W] shared by ``t_bcast`` identical steps).  cat = (buffer, first channel[, planes of the buffer]);
    planes: write the output as spike planes -- the returned spike tensor is then a ghost that carries them (``planes_of``); a ghost
    residual is read from its planes."""
    if write_v is None:
        write_v = _STATE_WRITEBACK
    state = (running_mean, running_var, bool(use_batch_stats), None if momentum is None else float(momentum), float(eps))
    sg_id = SURROGATE_IDS[surrogate] if isinstance(surrogate, str) else int(surrogate)
    C_ = y_seq.shape[-3]
    res_sp = planes_of(residual) if residual is not None else None
    cat_sp = cat is not None and len(cat) > 2 and cat[2] is not None
    if residual is not None and res_sp is None and (cat_sp or planes) and C_ % 8 == 0 and is_small_int(residual):
        res_sp = to_planes(residual)             # a shortcut that arrives as fp32 spikes next to an output kept as planes
    planes = bool(planes or cat_sp) and C_ % 8 == 0 and (residual is None or res_sp is not None) and (cat is None or cat_sp)
    if residual is not None and res_sp is not None and not planes:
        residual, res_sp = dense(residual), None              # fp32 output asked for: the shortcut as fp32 as well
    if cat_sp and not planes:
        raise _lib.EasHipError('a concatenation buffer kept as spike planes needs producers that write planes')
    sp_arg = None
    if planes:
        sp_arg = cat[2] if cat is not None else True
    spikes, v_out, mo, sp = _BNLIFFn.apply(y_seq, gamma, beta, state, v_in, w, float(k_const), float(v_th), float(v_reset), int(flags), sg_id,
                                           _alpha_arg(sg_id, alpha), bool(want_mean), bool(write_v), int(t_bcast), residual,
                                           cat[0] if cat is not None else None, int(cat[1]) if cat is not None else 0, sp_arg,
                                           res_sp.contiguous() if res_sp is not None else None)
    if sp is not None:
        spikes._eas_sp = sp
    return spikes, v_out, mo


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
FUSED_EVAL = {'1': 'all', 'all': 'all', '0': False, 'auto': 'auto'}.get(os.environ.get('EAS_FUSED_EVAL', 'auto'), 'auto')
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
    if not FUSED_EVAL or torch.is_grad_enabled() or _CONV_SINK is not None or torch.nn.modules.module._global_forward_hooks:
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
    if FUSED_EVAL == 'auto':
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
    return packs if (packs is not None and _PACK_SCOPE is not None and packs.get('gen') == _PACK_SCOPE) else None


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
        pk = pk[0] if (pk is not None and _PACK_SCOPE is not None and pk.get('gen') == _PACK_SCOPE) else conv_pack_weights(c.weight, 0)
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
    write_v = _STATE_WRITEBACK
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
        want_planes = bool(cat_sp or (planes[i] and cat is None)) and SPIKE_PLANES
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


# Number of identical copies the current batch stands for (set by SeqToANNContainer while it runs a stateless block
# ONCE for T identical time steps): batch-norm statistics are unchanged by replication except for the sample count
# used in the unbiased running variance.
_REPLICAS = 1


class replicated:
    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global _REPLICAS
        self.prev, _REPLICAS = _REPLICAS, self.n

    def __exit__(self, *exc):
        global _REPLICAS
        _REPLICAS = self.prev


def _channel_slice_of4(g, Cc):
    """total channel count if ``g`` [N,C,H,W] is a channel slice of a contiguous wider tensor (what the backward of an in-place
    concatenation hands out), Cc if it is contiguous itself, else 0."""
    if g.is_contiguous():
        return Cc
    N, C_, H, W = g.shape
    st = g.stride()
    if st[3] == 1 and st[2] == W and st[1] == H * W and st[0] % (H * W) == 0 and st[0] // (H * W) > C_ and (g.data_ptr() % 16) == 0:
        return st[0] // (H * W)
    return 0


class _BNSiLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, bn_state, cat_buf=None, cat_c0=0):
        """cat_buf [N,Ctot,H,W]: the result is written as channels cat_c0.. of it and returned as a view (concatenation in place)."""
        running_mean, running_var, use_batch_stats, momentum, eps, replicas = bn_state
        _dev(y, gamma, beta)
        L = _lib.lib()
        y = _f32c(y)
        N, Cc = y.shape[:2]
        HW = y.shape[-1] * y.shape[-2]
        dev = y.device
        if use_batch_stats:
            mean = torch.empty(Cc, dtype=torch.float32, device=dev)
            invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
            t0 = _timer_mark()
            pend, keep, nb_ = _pending_stats(L, y, N, Cc, HW, replicas, eps, momentum, running_mean, running_var, dev)
            if nb_:
                _timer_add('eas_bn_stats', t0, nb_)
        else:
            pend = keep = None
            mean = running_mean
            invstd = _eval_invstd(running_var, eps)
        if cat_buf is not None:
            assert cat_buf.is_contiguous() and cat_buf.dtype == torch.float32 and cat_buf.shape[0] == N and cat_buf.shape[2:] == y.shape[2:]
            out = cat_buf.narrow(1, cat_c0, Cc)
            ctot = cat_buf.shape[1]
        else:
            out = torch.empty_like(y)
            ctot = 0
        _call('eas_bn_silu_fwd', 8 * y.numel(), L.eas_bn_silu_fwd_ex, ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(out), N, Cc,
              HW, C.byref(pend) if pend is not None else None, ctot, 0, stream())
        del keep
        ctx.save_for_backward(y, mean, invstd, gamma, beta)
        ctx.cfg = (bool(use_batch_stats), N, Cc, HW)
        return out

    @staticmethod
    def backward(ctx, g):
        y, mean, invstd, gamma, beta = ctx.saved_tensors
        batch_stats, N, Cc, HW = ctx.cfg
        L = _lib.lib()
        # the gradient of an in-place concatenation arrives as a channel slice of the concatenation's gradient: read in place
        ctot = _channel_slice_of4(g, Cc) if (g.dim() == 4 and g.dtype == torch.float32) else 0
        if ctot == 0:
            g = _f32c(g)
        gy, ggamma, gbeta = torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(beta)
        ws = torch.empty(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=y.device)
        _call('eas_bn_silu_bwd', 12 * y.numel(), L.eas_bn_silu_bwd, ptr(g), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
              int(batch_stats), ptr(gy), ptr(ggamma), ptr(gbeta), ptr(ws), N, Cc, HW, ctot, 0, stream())
        return gy, ggamma, gbeta, None, None, None


class _BNSiLU2Fn(torch.autograd.Function):
    """Two BN + SiLU layers on the two channel ranges of ONE convolution output y12 [N,Ca+Cb,H,W] (two real-valued convolutions that read
    the same input computed as one, ``conv2d_dual``): each reads its channel slice in place, the backward writes both slices of ONE
    gradient tensor -- the convolution's input gradient needs no addition of two branch gradients."""

    @staticmethod
    def forward(ctx, y12, gamma_a, beta_a, gamma_b, beta_b, state_a, state_b, cat_a, cat_b):
        ctx.set_materialize_grads(False)      # a result nobody differentiates arrives as None in backward, not as a zero tensor
        L = _lib.lib()
        y12 = _f32c(y12)
        N, Ct, H, W = y12.shape
        HW = H * W
        dev = y12.device
        outs, saved, cfgs = [], [], []
        c0 = 0
        for gamma, beta, state, cat in ((gamma_a, beta_a, state_a, cat_a), (gamma_b, beta_b, state_b, cat_b)):
            running_mean, running_var, use_batch_stats, momentum, eps, replicas = state
            Cc = gamma.shape[0]
            _dev(gamma, beta)
            yp = y12.data_ptr() + 4 * c0 * HW
            pend = keep = None
            if use_batch_stats:
                mean = torch.empty(Cc, dtype=torch.float32, device=dev)
                invstd = torch.empty(Cc, dtype=torch.float32, device=dev)
                t0 = _timer_mark()
                pend, keep, nb_ = _pending_stats(L, y12, N, Cc, HW, replicas, eps, momentum, running_mean, running_var, dev, y_ptr=yp, y_ctot=Ct,
                                                 keep_slot=c0 == 0)
                if nb_:
                    _timer_add('eas_bn_stats', t0, nb_)
            else:
                mean = running_mean
                invstd = _eval_invstd(running_var, eps)
            if cat is not None:
                out = cat[0].narrow(1, cat[1], Cc)
                ctot = cat[0].shape[1]
            else:
                out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                ctot = 0
            _call('eas_bn_silu_fwd', 8 * N * Cc * HW, L.eas_bn_silu_fwd_ex, yp, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(out), N, Cc, HW,
                  C.byref(pend) if pend is not None else None, ctot, Ct, stream())
            del keep
            outs.append(out)
            saved += [mean, invstd, gamma, beta]
            cfgs.append((bool(use_batch_stats), Cc, c0))
            c0 += Cc
        assert c0 == Ct
        ctx.save_for_backward(y12, *saved)
        ctx.cfgs = cfgs
        ctx.dims = (N, Ct, HW)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_a, g_b):
        y12, *saved = ctx.saved_tensors
        N, Ct, HW = ctx.dims
        L = _lib.lib()
        gy12 = torch.empty_like(y12)
        res = []
        for i, g in enumerate((g_a, g_b)):
            mean, invstd, gamma, beta = saved[4 * i:4 * i + 4]
            batch_stats, Cc, c0 = ctx.cfgs[i]
            ggamma, gbeta = torch.empty_like(gamma), torch.empty_like(beta)
            if g is None:
                gy12.narrow(1, c0, Cc).zero_()
                ggamma.zero_(); gbeta.zero_()
            else:
                ctot = _channel_slice_of4(g, Cc) if (g.dim() == 4 and g.dtype == torch.float32) else 0
                if ctot == 0:
                    g = _f32c(g)
                ws = torch.empty(L.eas_bn_workspace_doubles(Cc), dtype=torch.float64, device=y12.device)
                off = 4 * c0 * HW
                _call('eas_bn_silu_bwd', 12 * N * Cc * HW, L.eas_bn_silu_bwd, ptr(g), y12.data_ptr() + off, ptr(mean), ptr(invstd), ptr(gamma),
                      ptr(beta), int(batch_stats), gy12.data_ptr() + off, ptr(ggamma), ptr(gbeta), ptr(ws), N, Cc, HW, ctot, Ct, stream())
            res += [ggamma, gbeta]
        return (gy12,) + tuple(res) + (None, None, None, None)


def _bn_state(bn):
    batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    update = batch and bn.training and bn.track_running_stats
    if update and bn.num_batches_tracked is not None:
        bump_counter(bn.num_batches_tracked)
    return (bn.running_mean if (update or not batch) else None, bn.running_var if (update or not batch) else None, bool(batch),
            float(bn.momentum) if update else None, float(bn.eps), _REPLICAS)


def bn_silu_pair(y12, bn_a, bn_b, cat_a=None, cat_b=None):
    """(silu(bn_a(y12[:, :Ca])), silu(bn_b(y12[:, Ca:]))) for the output y12 of ``conv2d_dual``; cat_a / cat_b = (buffer, first channel)
    as in ``bn_silu``."""
    return _BNSiLU2Fn.apply(y12, bn_a.weight, bn_a.bias, bn_b.weight, bn_b.bias, _bn_state(bn_a), _bn_state(bn_b), cat_a, cat_b)


def bn_silu_supported(y):
    return y.is_cuda and y.dim() == 4 and y.dtype == torch.float32 and (y.shape[-1] * y.shape[-2]) % 4 == 0


def bn_silu(y, bn, cat=None):
    """silu(batch_norm(y)) for a plain ``nn.BatchNorm2d`` module ``bn`` (running statistics updated like F.batch_norm).
    cat = (buffer [N,Ctot,H,W], first channel): the result is written into that channel range of the buffer and returned as a view
    (the caller joins the buffer with ``join_channels``)."""
    state = _bn_state(bn)
    if cat is not None:
        return _BNSiLUFn.apply(y, bn.weight, bn.bias, state, cat[0], cat[1])
    return _BNSiLUFn.apply(y, bn.weight, bn.bias, state)


# ------------------------------------------------------------------------------------------------ K3
def smallconv_pack(jobs):
    """Arrange sampler convolution weights for the vector-ALU kernels (eas_smallconv_pack_weights; up to 8 per launch).
    jobs: (w [Cout,Cin,k,k], mode, o_total, o_off, wr or None) -- mode 0 forward, mode 1 input gradient; several weights may share one
    packed tensor ``wr`` side by side along its output axis (o_total / o_off).  Returns the packed tensors, one per job."""
    L = _lib.lib()
    arr = (_lib.EasSmallconvPackJob * len(jobs))()
    outs = []
    for j, (w, mode, o_total, o_off, wr) in enumerate(jobs):
        w = _f32c(w)
        Cout, Cin, k = w.shape[0], w.shape[1], w.shape[-1]
        if wr is None:
            wr = torch.empty(L.eas_smallconv_packed_floats(Cout if mode else Cin, k, o_total), dtype=torch.float32, device=w.device)
        arr[j] = _lib.EasSmallconvPackJob(w.data_ptr(), wr.data_ptr(), Cin, Cout, k, int(mode), int(o_total), int(o_off))
        outs.append(wr)
    check(L.eas_smallconv_pack_weights(arr, len(jobs), stream()), 'eas_smallconv_pack_weights')
    return outs


def smallconv_fwd(x, w, b, relu=False, out=None, wr=None, x_tm=0):
    """Direct LDS-tiled conv (stride 1, 'same' padding) for the sampler's tiny-channel stacks.  ``out``: a contiguous
    [N,Cout,H,W] destination (e.g. one step's slice of a time-batched buffer) instead of a fresh tensor.  ``wr``: the weight already
    arranged by ``smallconv_pack`` (mode 0); else it is arranged here.  ``x_tm`` > 0: x is the collated micro-slice tensor
    [S, x_tm, Cin, H, W] and the result is time-major, newest slice first ([x_tm * S, Cout, H, W]; include/eas_hip.h)."""
    x = _f32c(x)
    if x_tm:
        assert x.dim() == 5 and x.shape[1] == x_tm
        x = x.view(-1, *x.shape[2:])
    N, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[-1]
    if wr is None:
        wr = smallconv_pack([(w, 0, Cout, 0, None)])[0]
    y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device) if out is None else out
    assert y.is_contiguous() and y.shape == (N, Cout, H, W)
    _call('eas_smallconv_fwd', 4 * (x.numel() + y.numel()), _lib.lib().eas_smallconv_fwd, ptr(x), ptr(wr), ptr(b), ptr(y), N, Cin, Cout,
          H, W, k, int(relu), int(x_tm), stream())
    return y


def smallconv_bwd_input(gy, w, relu_mask=None, out=None, wr=None):
    """``wr``: the weight arranged by ``smallconv_pack`` with mode 1"""
    gy = _f32c(gy)
    N, Cout, H, W = gy.shape
    Cin, k = w.shape[1], w.shape[-1]
    if wr is None:
        wr = smallconv_pack([(w, 1, Cin, 0, None)])[0]
    gx = torch.empty((N, Cin, H, W), dtype=torch.float32, device=gy.device) if out is None else out
    assert gx.is_contiguous() and gx.shape == (N, Cin, H, W)
    _call('eas_smallconv_bwd_input', 4 * (gy.numel() + gx.numel()), _lib.lib().eas_smallconv_bwd_input, ptr(gy), ptr(wr),
          ptr(relu_mask), ptr(gx), N, Cin, Cout, H, W, k, stream())
    return gx


def smallconv_bwd_input_dual(gy, wr8, k, mask_a, mask_b, out_a, out_b):
    """input gradients of two 4 -> 4 convolutions that received the same grad_y [N,4,H,W], in one pass (wr8: both weights packed with
    mode 1 side by side, o_total 8); each masked by the ReLU output in front of its convolution"""
    gy = _f32c(gy)
    N, C4, H, W = gy.shape
    assert C4 == 4 and out_a.is_contiguous() and out_b.is_contiguous() and out_a.shape == gy.shape == out_b.shape
    _call('eas_smallconv_bwd_input', 4 * 3 * gy.numel(), _lib.lib().eas_smallconv_bwd_input_dual, ptr(gy), ptr(wr8), ptr(mask_a), ptr(mask_b),
          ptr(out_a), ptr(out_b), N, H, W, int(k), stream())


def smallconv_bwd_weight(gy, x, w, x_tm=0):
    """``x_tm`` > 0: x is the collated micro-slice tensor [S, x_tm, Cin, H, W], gy time-major (see ``smallconv_fwd``)"""
    gy, x = _f32c(gy), _f32c(x)
    N, Cout, H, W = gy.shape
    assert x.numel() == N * w.shape[1] * H * W and (not x_tm or (x.dim() == 5 and x.shape[1] == x_tm))
    Cin, k = w.shape[1], w.shape[-1]
    L = _lib.lib()
    gw, gb = torch.empty_like(w), torch.empty(Cout, dtype=torch.float32, device=gy.device)
    ws = torch.empty(L.eas_smallconv_wgrad_workspace_floats(Cin, Cout, k), dtype=torch.float32, device=gy.device)
    _call('eas_smallconv_bwd_weight', 4 * (gy.numel() + x.numel()), L.eas_smallconv_bwd_weight, ptr(gy), ptr(x), ptr(gw), ptr(gb),
          ptr(ws), N, Cin, Cout, H, W, k, int(x_tm), stream())
    return gw, gb


def _conv_stack_fwd(x, params, k, mids=None, packs=None):
    """Conv(k, pad k//2) [+ ReLU + Conv]*: returns (out, inputs of every conv).  ReLU is fused into the producing conv.
    ``mids[i]``: destination of conv i's output for i < n-1 (the input of conv i+1); ``packs[i]``: conv i's arranged weight."""
    ins = []
    n = len(params) // 2
    for i in range(n):
        ins.append(x)
        x = smallconv_fwd(x, params[2 * i], params[2 * i + 1], relu=(i < n - 1), out=mids[i] if mids is not None and i < n - 1 else None,
                          wr=packs[i] if packs is not None else None)
    return x, ins


# K3 in one launch per micro-step (eas_arsnn_fused_step_fwd): the second convolutions of the input stack and of the gate stack run inside
# the step kernel, the two input gradients of those convolutions share one pass over the step's gradient (eas_smallconv_bwd_input_dual).
# EAS_ARSNN_FUSED=0: development switch, the separate launches.
ARSNN_FUSED = os.environ.get('EAS_ARSNN_FUSED', '1') == '1'


class _ARSNNFn(torch.autograd.Function):
    """Whole adaptive-sampler loop as ONE autograd node (embedding.py:141-226): conv stacks via eas_smallconv_*,
    the per-step integrate / fire / reset / segment-write via eas_arsnn_step_* (depth-2 stacks with four hidden channels: the second
    convolutions inside eas_arsnn_fused_step_fwd).

    ``running`` in the configuration selects the plain gated recurrence of the simpler embeddings instead
    (SpikingEmbedding "rsnn" embedding.py:229-316, LIFEmbedding "snn" :28-76): no segments, the output is the running sum
    of the pre-reset potentials ('sum') or the last potential ('last').  With an empty input stack ``ev`` already holds
    the [Tm,N,2*C2,H,W] gate|current planes; with an empty gate stack there is no recurrent convolution."""

    @staticmethod
    def forward(ctx, ev, cfg, *params):
        ctx.set_materialize_grads(False)      # a result nobody differentiates arrives as None in backward, not as a zero tensor
        _dev(ev, *params)
        L = _lib.lib()
        k, depth, Ts, readout, sat, wz, ab, thresh, v_reset, soft, record, running, d_in, d_gate = cfg[:14]
        ev = _f32c(ev)
        pin, pg = params[:2 * d_in], params[2 * d_in:]
        collated = len(cfg) > 14 and cfg[14]     # ev is [N, Tm, Cin, H, W] as the loader collates it; the kernels read it time-major, newest first
        v_record = len(cfg) > 15 and cfg[15]     # embedding.py:180: the potentials of the neurons that did not fire, step after step (debugging output)
        if collated:
            N, Tm, Cin, H, W = ev.shape
        else:
            Tm, N, Cin, H, W = ev.shape
        HW = H * W
        if HW % 4 != 0:
            raise _lib.EasHipError('sampler needs H*W divisible by 4')
        dev = ev.device
        need_grad = any(ctx.needs_input_grad[2:]) or ctx.needs_input_grad[0]
        st = stream()
        # every convolution weight arranged for the kernels by ONE launch
        jobs = [(pin[2 * i], 0, pin[2 * i].shape[0], 0, None) for i in range(d_in)] + [(pg[2 * i], 0, pg[2 * i].shape[0], 0, None) for i in range(d_gate)]
        pk = smallconv_pack(jobs) if jobs else []
        pk_in, pk_g = pk[:d_in], pk[d_in:]
        fused = bool(ARSNN_FUSED and d_in == 2 and d_gate in (0, 2) and W % 4 == 0 and pin[2].shape[:2] == (4, 4) and pin[0].shape[0] == 4
                     and (d_gate == 0 or (pg[2].shape[:2] == (4, 4) and pg[0].shape[0] == 4)) and _CONV_SINK is None and Tm > 0)
        ctx.in_collated = bool(collated)
        if collated and not (fused and not ctx.needs_input_grad[0]):
            # only the fused step reads the collated layout: everything else gets the flipped, time-major copy (embedding.py:147-156)
            ev = torch.stack([ev[:, Tm - 1 - t] for t in range(Tm)])
            collated = False
        if fused:
            # first convolution + ReLU of the input stack for all Tm steps at once; the second one runs inside the step kernel
            if collated:
                A_in = smallconv_fwd(ev, pin[0], pin[1], relu=True, wr=pk_in[0], x_tm=Tm).view(Tm, N, 4, H, W)
                in_ins = [ev, A_in.view(Tm * N, 4, H, W)]
            else:
                A_in = smallconv_fwd(ev.view(Tm * N, Cin, H, W), pin[0], pin[1], relu=True, wr=pk_in[0]).view(Tm, N, 4, H, W)
                in_ins = [ev.view(Tm * N, Cin, H, W), A_in.view(Tm * N, 4, H, W)]
            X = None
            C2 = 2
        elif d_in:
            X, in_ins = _conv_stack_fwd(ev.view(Tm * N, Cin, H, W), pin, k, packs=pk_in)
            X = X.view(Tm, N, X.shape[1], H, W)
            C2 = X.shape[2] // 2
        else:
            X, in_ins = ev, []
            C2 = X.shape[2] // 2
        shape = (N, C2, H, W)
        v = vsum = None          # first step: the kernel takes zero potentials / sums, seg = 0, t_last = -1 (no zero fills)
        if os.environ.get('EAS_ARSNN_ZERO_FILL') == '1':     # development: explicit zero state tensors
            v = torch.zeros(shape, device=dev)
            vsum = torch.zeros(shape, device=dev)
        # inputs of every gate conv for all Tm steps, written in place by the producing kernels: the batched weight
        # gradient reads them as one [Tm*N,...] tensor (no concatenation).  gate_in[0][t] = spike entering step t.
        keep = need_grad and d_gate
        # Step 0 of the gate stack.  The spike entering it is the constant 0 for every sample, so gate_conv(0) is ONE image (bias terms and
        # border effects) shared by the whole batch: computed on one zero image and broadcast, and its backward runs once on the gradient
        # summed over the batch (the stack is linear in its output gradient given the shared input) -- two N-image convolutions, one
        # N-image input gradient and a quarter of the batched weight-gradient work less.  EAS_ARSNN_STEP0=full: development switch.
        fast0 = bool(d_gate) and Tm > 0 and _CONV_SINK is None and os.environ.get('EAS_ARSNN_STEP0', 'shared') != 'full'
        if keep:
            gate_in = [torch.empty((Tm, N, pg[2 * i].shape[1], H, W), device=dev) for i in range(d_gate)]
            if not fast0:
                gate_in[0][0].zero_()
            spike = gate_in[0][0]
        else:
            gate_in = None
            spike = None if fast0 else torch.zeros(shape, device=dev)
        seg = torch.empty(shape, dtype=torch.int8, device=dev)        # segment counter 0..Ts and last-spike step -1..Tm-1: one byte each
        tl = torch.empty(shape, dtype=torch.int8, device=dev)
        if Tm == 0 or v is not None:
            seg.zero_(); tl.fill_(-1)
        agg = torch.zeros((1 if running else Ts,) + shape, device=dev)
        zero_rec = None if (d_gate or fused) else torch.zeros((N, 2 * C2, H, W), device=dev)
        saved = []
        t_rec = []
        v_rec = []
        recording = bool(record or v_record)     # the reference leaves its loop once every pixel has Ts segments (:200-201): nothing is recorded after
        for t in range(Tm):
            if _CONV_SINK is not None and d_gate:
                _CONV_SINK.sampler_spikes.append(spike)
            a_g = r_const = R = None
            g_ins = []
            if d_gate and t == 0 and fast0:
                R1, g_ins = _conv_stack_fwd(torch.zeros((1,) + shape[1:], device=dev), pg, k, packs=pk_g)         # one image
                if fused:
                    r_const = R1[0]
                else:
                    R = R1.expand(N, *R1.shape[1:]).contiguous()
            elif d_gate and fused:
                # first gate convolution + ReLU (kept for its weight gradient); the second one runs inside the step kernel
                a_g = smallconv_fwd(spike, pg[0], pg[1], relu=True, out=gate_in[1][t] if keep else None, wr=pk_g[0])
                g_ins = [spike, a_g]
            elif d_gate:
                R, g_ins = _conv_stack_fwd(spike, pg, k, [gate_in[i + 1][t] for i in range(d_gate - 1)] if keep else None, packs=pk_g)
            elif not fused:
                R = zero_rec
            v_n, vs_n = torch.empty(shape, device=dev), torch.empty(shape, device=dev)
            sp_n = gate_in[0][t + 1] if keep and t + 1 < Tm else torch.empty(shape, device=dev)
            if need_grad:
                gate, vn = torch.empty(shape, device=dev), torch.empty(shape, device=dev)
                seg_b, tl_b = torch.empty_like(seg), torch.empty_like(tl)
            else:
                gate = vn = seg_b = tl_b = None
            if fused:
                _call('eas_arsnn_step_fwd', 38 * v_n.numel(), L.eas_arsnn_fused_step_fwd, ptr(A_in[t]), ptr(pk_in[1]), ptr(pin[3]), ptr(a_g),
                      ptr(pk_g[1]) if d_gate else None, ptr(pg[3]) if d_gate else None, ptr(r_const), ptr(v), ptr(vsum), ptr(seg), ptr(tl),
                      ptr(agg), ptr(v_n), ptr(vs_n), ptr(sp_n), ptr(gate), ptr(vn), ptr(seg_b), ptr(tl_b), t, Ts, 3 if running else readout,
                      int(sat), thresh, v_reset, int(soft), N, H, W, k, st)
            else:
                _call('eas_arsnn_step_fwd', 38 * v_n.numel(), L.eas_arsnn_step_fwd, ptr(X[t]), ptr(R), ptr(v), ptr(vsum), ptr(seg), ptr(tl),
                      ptr(agg), ptr(v_n), ptr(vs_n), ptr(sp_n), ptr(gate), ptr(vn), ptr(seg_b), ptr(tl_b), t, Ts, 3 if running else readout,
                      int(sat), thresh, v_reset, int(soft), N, C2, HW, st)
            if need_grad:
                saved.append((g_ins, v, vsum, gate, vn, seg_b, tl_b))
            v, vsum, spike = v_n, vs_n, sp_n
            if recording:
                if record:
                    t_rec.append(tl.to(torch.int32))
                if v_record:
                    # where the neuron did not fire the potential after the reset IS the pre-reset potential (vn * 1 + v_reset * 0, or
                    # vn - thresh * 0): the state tensor the step kernel wrote serves as the reference's ``vmem_no_reset``
                    v_rec.append(v_n[sp_n == 0])
                if not running and int(seg.min()) >= Ts:          # (a host synchronisation: only these debugging outputs pay it)
                    recording = False
        pre_relu = None
        if running:
            out = vsum if running == 'sum' else v
        else:
            check(L.eas_arsnn_tail_fwd(ptr(v), ptr(vsum), ptr(spike), ptr(seg), ptr(tl), ptr(agg), Tm, Ts, readout, int(wz), N, C2,
                                       HW, st), 'eas_arsnn_tail_fwd')
            out = agg
        if ab:
            pre_relu = out
            out = torch.relu(out)
        ctx.cfg = cfg
        ctx.fast0 = fast0
        ctx.fused = fused
        ctx.dims = (Tm, N, Cin, C2, H, W)
        ctx.collated = bool(collated)
        # ``agg`` is this node's own output unless ``running``: keeping it on ctx would tie output -> grad_fn -> ctx -> output
        # into a reference cycle (Ts*N*C2*H*W floats held until the cyclic GC runs); the backward reads it in running mode only
        # (as a valid dummy pointer).  ``pre_relu`` likewise is ``out`` before the ReLU, a distinct tensor.  The tensors are
        # intermediates this node created itself (none is an input or an output of the node), so they need no version tracking.
        ctx.saved = (saved, in_ins, spike, seg, tl, pre_relu, agg if running else None, gate_in)
        ctx.params = params
        ctx.ev_needs_grad = ctx.needs_input_grad[0]
        rec = torch.stack(t_rec) if record else None
        vrec = torch.cat(v_rec) if v_record else None
        ctx.mark_non_differentiable(*[t_ for t_ in (rec, vrec) if t_ is not None])
        return out, rec, vrec

    @staticmethod
    def backward(ctx, g_out, _g_rec=None, _g_vrec=None):
        L = _lib.lib()
        k, depth, Ts, readout, sat, wz, ab, thresh, v_reset, soft, record, running, d_in, d_gate = ctx.cfg[:14]
        Tm, N, Cin, C2, H, W = ctx.dims
        saved, in_ins, spike_last, seg, tl, pre_relu, agg, gate_in = ctx.saved
        params = ctx.params
        fused = ctx.fused
        pin, pg = params[:2 * d_in], params[2 * d_in:]
        HW = H * W
        st = stream()
        if g_out is None:
            return (None,) * (2 + len(params))
        g_out = _f32c(g_out)
        if ab:
            g_out = g_out * (pre_relu > 0)
        dev = g_out.device
        shape = (N, C2, H, W)
        # the weights arranged for the input-gradient kernels by ONE launch
        jobs, slot = [], {}
        if fused and d_gate:
            wr8 = torch.empty(L.eas_smallconv_packed_floats(4, k, 8), dtype=torch.float32, device=dev)
            jobs += [(pin[2], 1, 8, 0, wr8), (pg[2], 1, 8, 4, wr8)]
        for name, w, need in (('in1', pin[2] if d_in > 1 else None, d_in > 1), ('in0', pin[0] if d_in else None, d_in and ctx.ev_needs_grad),
                              ('g1', pg[2] if d_gate > 1 else None, d_gate > 1), ('g0', pg[0] if d_gate else None, bool(d_gate))):
            if need:
                slot[name] = len(jobs)
                jobs.append((w, 1, w.shape[1], 0, None))
        pk = smallconv_pack(jobs) if jobs else []
        wr = {n: pk[i] for n, i in slot.items()}
        if running:
            g_agg = agg                                  # never read in running mode (no segment writes); a valid pointer
            zeros = torch.zeros(shape, device=dev)
            g_v, g_vs = (zeros, g_out.contiguous()) if running == 'sum' else (g_out.contiguous(), zeros)
        else:
            g_agg = g_out
            g_v = torch.empty(shape, device=dev)
            g_vs = torch.empty(shape, device=dev)
            check(L.eas_arsnn_tail_bwd(ptr(g_agg), ptr(spike_last), ptr(seg), ptr(tl), ptr(g_v), ptr(g_vs), Tm, Ts, readout, int(wz),
                                       N, C2, HW, st), 'eas_arsnn_tail_bwd')
        g_spike = None
        gX = torch.empty((Tm, N, 2 * C2, H, W), device=dev)
        # gradient reaching each conv of the gate stack at every step, written into time-batched buffers by the producing
        # kernels (batched weight-grad at the end); the last conv's is gX itself
        g_stage = [torch.empty((Tm, N, pg[2 * i].shape[0], H, W), device=dev) for i in range(d_gate - 1)] + ([gX] if d_gate else [])
        gA_in = torch.empty((Tm, N, 4, H, W), device=dev) if fused else None      # gradient at the input stack's hidden planes
        for t in range(Tm - 1, -1, -1):
            g_ins, v_prev, vs_prev, gate, vn, seg_b, tl_b = saved[t]
            g_vp, g_vsp = torch.empty_like(g_v), torch.empty_like(g_v)
            check(L.eas_arsnn_step_bwd(ptr(g_v), ptr(g_vs), ptr(g_spike), ptr(g_agg), ptr(v_prev), ptr(vs_prev), ptr(gate), ptr(vn),
                                       ptr(seg_b), ptr(tl_b), ptr(gX[t]), ptr(g_vp), ptr(g_vsp), t, Ts, 3 if running else readout,
                                       int(sat), thresh, v_reset, int(soft), 1.0, N, C2, HW, st), 'eas_arsnn_step_bwd')
            g_v, g_vs = g_vp, g_vsp
            if fused:
                a_in_t = in_ins[1].view(Tm, N, 4, H, W)[t]
                if d_gate and not (t == 0 and ctx.fast0) and t > 0:
                    # both second convolutions' input gradients from one pass over gX[t], each masked by its ReLU
                    smallconv_bwd_input_dual(gX[t], wr8, k, a_in_t, g_ins[1], gA_in[t], g_stage[0][t])
                    g_spike = smallconv_bwd_input(g_stage[0][t], pg[0], None, wr=wr['g0'])
                else:
                    smallconv_bwd_input(gX[t], pin[2], a_in_t, out=gA_in[t], wr=wr['in1'])
                    if d_gate and t == 0 and not ctx.fast0:
                        smallconv_bwd_input(gX[t], pg[2], g_ins[1], out=g_stage[0][t], wr=wr['g1'])
                    g_spike = None
                continue
            g = gX[t]
            for i in range(d_gate - 1, -1, -1):           # g = gradient at the output of gate conv i = g_stage[i][t]
                if t == 0 and (i == 0 or ctx.fast0):
                    break                      # spike input of step 0 is the constant 0 (fast0: the whole step-0 stack is done below)
                # ReLU in front of conv i fused as a mask
                g = smallconv_bwd_input(g, pg[2 * i], g_ins[i] if i > 0 else None, out=g_stage[i - 1][t] if i > 0 else None,
                                        wr=wr.get('g%d' % i))
            g_spike = g if (t > 0 and d_gate) else None
        grads_g = []
        t0 = 1 if ctx.fast0 else 0
        for i in range(d_gate):
            if Tm > t0:
                gw, gb = smallconv_bwd_weight(g_stage[i][t0:].flatten(0, 1), gate_in[i][t0:].flatten(0, 1), pg[2 * i])
            else:
                gw, gb = torch.zeros_like(pg[2 * i]), torch.zeros_like(pg[2 * i + 1])
            grads_g += [gw, gb]
        if ctx.fast0:
            # step 0: the stack's input is the same zero image for every sample, so its parameter gradients are those of ONE image
            # with the output gradient summed over the batch
            g1 = gX[0].sum(0, keepdim=True)
            g_ins0 = saved[0][0]
            for i in range(d_gate - 1, -1, -1):
                gw, gb = smallconv_bwd_weight(g1, g_ins0[i], pg[2 * i])
                grads_g[2 * i] = grads_g[2 * i] + gw
                grads_g[2 * i + 1] = grads_g[2 * i + 1] + gb
                if i > 0:
                    g1 = smallconv_bwd_input(g1, pg[2 * i], g_ins0[i], wr=wr['g1'])
        # input conv stack, all Tm steps at once
        grads_in = [None] * (2 * d_in)
        g = gX.view(Tm * N, 2 * C2, H, W)
        for i in range(d_in - 1, -1, -1):
            grads_in[2 * i], grads_in[2 * i + 1] = smallconv_bwd_weight(g, in_ins[i], pin[2 * i], x_tm=Tm if (i == 0 and ctx.collated) else 0)
            if fused and i == 1:
                g = gA_in.view(Tm * N, 4, H, W)              # the step loop already produced this input gradient
            elif i > 0 or ctx.ev_needs_grad:
                g = smallconv_bwd_input(g, pin[2 * i], in_ins[i] if i > 0 else None, wr=wr.get('in%d' % i))
            else:
                g = None
        g_ev = g.view(Tm, N, Cin, H, W) if ctx.ev_needs_grad else None
        if g_ev is not None and ctx.in_collated:
            g_ev = g_ev.flip(0).transpose(0, 1)          # back to the loader's [N, Tm, ...] in forward time order
        return (g_ev, None) + tuple(grads_in) + tuple(grads_g)


def arsnn_forward(ev_rev, input_params, gate_params, kernel_size, Ts, readout, spike_attach, write_zero, use_abs, thresh,
                  v_reset, record=False, collated=False, v_record=False):
    """ev_rev: [Tm, N, 2, H, W] micro-slices, newest first -- or, with ``collated``, the loader's [N, Tm, 2, H, W] in forward time order
    (the fused step's kernels then read it newest first themselves: no flipped copy of the input).  *_params: [w0, b0, (w1, b1, ...)]."""
    depth = len(input_params) // 2
    soft = v_reset is None
    cfg = (int(kernel_size), depth, int(Ts), READOUT_IDS[readout], bool(spike_attach), bool(write_zero), bool(use_abs),
           float(thresh), 0.0 if soft else float(v_reset), soft, bool(record), None, depth, len(gate_params) // 2, bool(collated), bool(v_record))
    out, rec, vrec = _ARSNNFn.apply(ev_rev, cfg, *input_params, *gate_params)
    return (out, rec, vrec) if v_record else (out, rec)


def gated_recurrence(ev_or_x, input_params, gate_params, kernel_size, readout, relu, thresh, v_reset):
    """The plain gated spiking recurrence of SpikingEmbedding / LIFEmbedding: vn = sigmoid(g)*v + c, fire (> thresh), reset;
    returns sum_t vn ('sum') or the last potential ('last').  ``input_params`` empty: ``ev_or_x`` is [Tm,N,2*C2,H,W]
    (gate pre-activations | currents); ``gate_params`` empty: no recurrent convolution."""
    if readout not in ('sum', 'last'):
        raise NotImplementedError(readout)
    soft = v_reset is None
    d_in, d_gate = len(input_params) // 2, len(gate_params) // 2
    cfg = (int(kernel_size), max(d_in, d_gate), 1, 0, False, False, bool(relu), float(thresh), 0.0 if soft else float(v_reset), soft,
           False, readout, d_in, d_gate)
    return _ARSNNFn.apply(ev_or_x, cfg, *input_params, *gate_params)[0]


class _SmallConvFn(torch.autograd.Function):
    """One tiny-channel convolution (+ fused ReLU) of the embeddings on the LDS-tiled direct kernels."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        _dev(x, w, b)
        y = smallconv_fwd(x, w, b, relu=relu)
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = _f32c(g)
        if y is not None:
            g = g * (y > 0)
        gx = smallconv_bwd_input(g, w) if ctx.needs_input_grad[0] else None
        gw, gb = smallconv_bwd_weight(g, x, w)
        return gx, gw, gb, None


def small_conv_stack(x, params):
    """Conv [+ ReLU + Conv]* with [w0, b0, w1, b1, ...] on x [N,C,H,W]."""
    n = len(params) // 2
    for i in range(n):
        x = _SmallConvFn.apply(x, params[2 * i], params[2 * i + 1], i < n - 1)
    return x

# ------------------------------------------------------------------------------------------------ K4 conv (MFMA)
_INEXACT = {}


def conv_inexact_flag(device):
    """Device int32 that the kernels OR with 1 when a tensor promised to hold spikes / small integers (one exact bf16 term: the fp32
    one-term convolutions, the conversion to spike planes) holds a value that is not exact in bf16.  One per device, persistent (a
    captured graph keeps writing the same word), sticky until ``clear_tag_violation``."""
    key = str(device)
    if key not in _INEXACT:
        _INEXACT[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _INEXACT[key]


def _tag_flag(x, x_terms):
    return ptr(conv_inexact_flag(x.device)) if x_terms == 1 else None


def tag_violation(device=None):
    """True when a kernel saw a mis-tagged tensor since the last ``clear_tag_violation`` (host synchronisation: outside timed regions)"""
    if device is not None:
        return str(device) in _INEXACT and bool(_INEXACT[str(device)].item())
    return any(bool(f.item()) for f in _INEXACT.values())


def clear_tag_violation():
    for f in _INEXACT.values():
        f.zero_()


def check_tags(what='this run'):
    """fail loudly when a tensor tagged "spikes / small integers" was not: its convolutions rounded their input to bf16, i.e. the results
    of ``what`` are NOT the fp32 results they claim to be.  Called by Trainer.save_ckpt, bench.py (after the timed region) and smoke()."""
    if tag_violation():
        raise _lib.EasHipError(f'{what}: a tensor tagged as spikes / small integers held values that are not exact in bf16 -- a one-term '
                               'convolution or a spike-plane conversion rounded them (mark_small_int on a real-valued tensor?)')


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
        if _WANT_CONV_STATS and bias is None:
            global _CONV_STATS_SLOT
            nb = _conv_stats_blocks(L, (NI, Cin, Cout, Hi, Wi, ksize, stride, 2))
            if 0 < nb <= CONV_STATS_MAX_BLOCKS:
                stats = torch.empty(Cout * nb * 2, dtype=torch.float64, device=y.device)
        _call('eas_conv_fwd', 2 * NI * Cin * Hi * Wi + 4 * y.numel(), L.eas_conv_fwd_planes, ptr(x_sp), ptr(packed), ptr(bias), ptr(y), NI, Cin, Cout,
              Hi, Wi, ksize, stride, ptr(stats), nb if stats is not None else 0, stream(), flops=fl, issue_flops=fl * 3)
        if stats is not None:
            _CONV_STATS_SLOT = (y, nb, stats, y._version)
        return y
    x = _f32c(x)
    if act:             # y = act(conv + bias) from the epilogue (eas_conv_fwd_act: an eval-mode BaseConv with its BatchNorm folded in)
        _call('eas_conv_fwd', 4 * (x.numel() + y.numel()), L.eas_conv_fwd_act, ptr(x), ptr(packed), ptr(bias), ptr(y), NI, Cin, Cout, Hi, Wi,
              ksize, stride, x_terms, int(act), _tag_flag(x, x_terms), stream(), flops=fl, issue_flops=fl * (3 if x_terms == 1 else 6))
        return y
    if _WANT_CONV_STATS and bias is None:
        nb = _conv_stats_blocks(L, (NI, Cin, Cout, Hi, Wi, ksize, stride, x_terms))
        if 0 < nb <= CONV_STATS_MAX_BLOCKS:
            stats = torch.empty(Cout * nb * 2, dtype=torch.float64, device=x.device)
            _call('eas_conv_fwd', 4 * (x.numel() + y.numel()), L.eas_conv_fwd_stats, ptr(x), ptr(packed), ptr(y), NI, Cin, Cout, Hi, Wi,
                  ksize, stride, x_terms, _tag_flag(x, x_terms), ptr(stats), nb, stream(), flops=fl, issue_flops=fl * (3 if x_terms == 1 else 6))
            _CONV_STATS_SLOT = (y, nb, stats, y._version)
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
DEFER_WGRAD_REDUCE = os.environ.get('EAS_DEFER_WGRAD_REDUCE', '0') == '1'
_PENDING_REDUCE = []          # (slab workspace kept alive, grad_w address, numel, slab count, [(parameter, address of its .grad, numel)])


class deferred_wgrad_reductions:
    """``with ops.deferred_wgrad_reductions():`` around forward + backward of a training step (see above)"""

    def __init__(self, flag=True):
        self.flag = bool(flag)

    def __enter__(self):
        global DEFER_WGRAD_REDUCE
        self.prev, DEFER_WGRAD_REDUCE = DEFER_WGRAD_REDUCE, self.flag

    def __exit__(self, exc_type, *exc):
        global DEFER_WGRAD_REDUCE, _PENDING_REDUCE
        DEFER_WGRAD_REDUCE = self.prev
        if exc_type is not None:
            # the backward pass did not finish: the gradient tensors of the pending jobs may already be gone (their addresses are all
            # that is kept) -- nothing is written, the jobs are dropped
            _PENDING_REDUCE = []
            _SIDE['pending'], _SIDE['keep'] = [], []
            return
        _flush_wgrad_reductions()


# Weight-gradient slab kernels on a side stream (round 5).  Nothing reads a deferred weight gradient before the end of the backward pass, so
# its slab kernel is off the pass's dependency chain: with WGRAD_SIDE_BATCH = n > 0 the launches are collected and, n at a time, issued on
# ONE side stream that forks from the main stream where the batch's last grad_y exists (one cross-stream edge per batch, not per launch:
# per-launch forks measured slower than no overlap) and joins in front of the batched reduction.  The chain's many small
# kernels (one or two waves per SIMD on the 8x10 / 16x20 maps) leave most of the chip idle; the slab kernels fill it.  Inputs and
# workspaces of the launches in flight are kept alive until the join, so the allocator cannot hand their memory to the main stream early.
WGRAD_SIDE_BATCH = int(os.environ.get('EAS_WGRAD_SIDE', '16'))       # 0: everything on the main stream; 16: swept on config 2 (12 / 15 / 17 / 20 / 24 lose 0.1-0.3 ms of its 0.37 ms)
WGRAD_SIDE_US = float(os.environ.get('EAS_WGRAD_SIDE_US', '0'))       # a batch also leaves once its estimated kernel time reaches this (0: count only)
WGRAD_SIDE_AT = tuple(int(v) for v in os.environ.get('EAS_WGRAD_SIDE_AT', '').split(',') if v.strip())   # or: after these launch counts of the pass
_SIDE = {'stream': None, 'pending': [], 'keep': [], 'dirty': False, 'us': 0.0, 'seen': 0}


def _wgrad_launch(job, keep, defer, issue_flops=0.0, nbytes=0.0):
    """run the slab-kernel launch ``job()`` now, or -- a deferred gradient with the side stream switched on -- with the next batch"""
    if not (defer and (WGRAD_SIDE_BATCH > 0 or WGRAD_SIDE_US > 0 or WGRAD_SIDE_AT) and DEFER_WGRAD_REDUCE):
        job()
        return
    _SIDE['pending'].append(job)
    _SIDE['keep'].extend(keep)
    _SIDE['us'] += max(issue_flops / 1.3e15, nbytes / 5e12) * 1e6
    _SIDE['seen'] += 1
    if ((WGRAD_SIDE_BATCH > 0 and len(_SIDE['pending']) >= WGRAD_SIDE_BATCH) or (WGRAD_SIDE_US > 0 and _SIDE['us'] >= WGRAD_SIDE_US)
            or _SIDE['seen'] in WGRAD_SIDE_AT):
        _side_flush()


def _side_flush():
    jobs, _SIDE['pending'] = _SIDE['pending'], []
    _SIDE['us'] = 0.0
    if not jobs:
        return
    if _SIDE['stream'] is None:
        _SIDE['stream'] = _lib.private_stream()          # joins graph captures: never a pooled stream (see there)
    side = _SIDE['stream']
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for job in jobs:
            job()
    _SIDE['dirty'] = True


def _side_join():
    _side_flush()
    if _SIDE['dirty']:
        torch.cuda.current_stream().wait_stream(_SIDE['stream'])
        _SIDE['dirty'] = False
    if os.environ.get('EAS_WGRAD_SIDE_LOG') and _SIDE['seen']:
        print(f"[eas] side stream: {_SIDE['seen']} slab launches in this pass", flush=True)
    _SIDE['keep'] = []
    _SIDE['seen'] = 0


def _flush_wgrad_reductions():
    """Reduce every pending slab set into its gradient tensor in ONE launch.  A job is only valid while the tensor handed to autograd is
    still the parameter's ``.grad`` (AccumulateGrad adopted it, nothing has been added to it in place): anything else means the
    deferral was switched on for a backward pass it is not sound for, and fails loudly instead of writing through a stale address."""
    global _PENDING_REDUCE
    _side_join()                 # slab kernels still pending or running on the side stream: the reduction reads their slabs
    jobs, _PENDING_REDUCE = _PENDING_REDUCE, []
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
    return (DEFER_WGRAD_REDUCE and w.is_leaf and w.grad is None and not torch.is_grad_enabled() and not w._backward_hooks
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
    if defer and owners and any(o[0] is p[0] for j in _PENDING_REDUCE for p in j[4] for o in owners):
        _flush_wgrad_reductions()           # a second contribution to the same weight: autograd adds the two as soon as this returns
        defer = False
    if defer and owners:
        if not _PENDING_REDUCE:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrad_reductions)
        _PENDING_REDUCE.append((ws, gw.data_ptr(), gw.numel(), int(nslabs), owners))
    else:
        arr = (_lib.EasWgradReduceJob * 1)(_lib.EasWgradReduceJob(ws.data_ptr(), gw.data_ptr(), gw.numel(), int(nslabs)))
        check(_lib.lib().eas_conv_wgrad_reduce_many(arr, 1, stream()), 'eas_conv_wgrad_reduce_many')


def _partial_call(name, nbytes, fn, *args, flops=0.0, issue_flops=0.0):
    """like _call for the *_partial entry points, whose non-negative return value is the slab count"""
    if _CALL_LOG is not None:
        _CALL_LOG.append((fn.__name__, args))
    if _TIMER is None:
        rc = fn(*args)
    else:
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        rc = fn(*args)
        e_.record()
        _TIMER.add(name, s_, e_, nbytes, flops, issue_flops)
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


# One backend: every dense 1x1 / 3x3 convolution of the step -- forward, input gradient, weight gradient -- runs on the
# hand-written matrix-core kernels (eas_conv_fwd / eas_conv_dgrad_s2 / eas_conv_wgrad).  The library convolution remains only for
# geometries those kernels do not cover at all (grouped / dilated / odd-width / other kernel sizes: none in the EAS-SNN models).


VERIFY_SMALL_INT = False     # tests switch this on: every tagged tensor is checked (host sync) before it is used


def mark_small_int(t):
    """Tag a tensor whose values are spikes or sums of a few spikes (exact in bf16): convolutions reading it use one
    bf16 term instead of three.  Untagged tensors always take the general three-term path."""
    t._eas_small_int = True
    return t


def is_small_int(t):
    return getattr(t, '_eas_small_int', False)


def _verify_tags(x, small_int):
    """test-suite check (VERIFY_SMALL_INT, host sync): a tagged tensor is exact in bf16; a ghost's planes hold small integers"""
    if not (small_int and VERIFY_SMALL_INT) or torch.cuda.is_current_stream_capturing():     # (a capture cannot read the device; the
        return                                                                                # eager iterations in front of it did)
    sp = planes_of(x)
    if sp is not None:
        v = sp.float()
        if not bool(((v == v.round()) & (v >= 0) & (v <= 255)).all()):
            raise _lib.EasHipError('the spike planes of a tensor do not hold small integers')
        return
    if not bool((x == x.to(torch.bfloat16).to(torch.float32)).all()):
        raise _lib.EasHipError('a tensor tagged as spikes / small integers is not exact in bf16')


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
SMALL_DGRAD = os.environ.get('EAS_SMALL_DGRAD', '1') != '0'


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
        ctx.tag = _TAG
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, x_sp = ctx.saved_tensors
        k, stride, x_terms, has_bias = ctx.cfg
        packs = ctx.packs
        gy = _f32c(gy)
        gx = gw = gb = None
        Cin = w.shape[1]
        if _TIMER is not None and ctx.tag is not None:
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
        small_d = (SMALL_DGRAD and ctx.needs_input_grad[0] and stride == 1 and k == 3 and w.dtype == torch.float32
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


_PACK_SCOPE = None       # the pack dictionaries' generation that is valid right now (inside ``packed_weights``), else None
_PACK_GEN = 0


_FROZEN = None          # {'model', 'gen', 'inv'} while a ``frozen_weights`` block is open, else None


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
        global _FROZEN
        self.prev = _FROZEN
        p = next(self.model.parameters(), None)
        if p is None or not p.is_cuda:
            return self
        _FROZEN = None                                   # (a nested block for another model: pack for real)
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
            _FROZEN = {'model': self.model, 'gen': gen, 'inv': inv}
        return self

    def __exit__(self, *exc):
        global _FROZEN
        _FROZEN = self.prev
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
    global _PACK_GEN
    if _FROZEN is not None and _FROZEN['model'] is model and not torch.is_grad_enabled() and not model.training:
        return _FROZEN['gen']                            # inside ``frozen_weights(model)``: packed when the block was entered
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
    _PACK_GEN += 1
    for c, d in zip(convs, plan['packs']):
        d['gen'] = _PACK_GEN
        object.__setattr__(c, '_eas_packs', d)
    for (owner, key, _, _), d in zip(duals, plan['dpacks']):
        d['gen'] = _PACK_GEN
        reg = getattr(owner, '_eas_dual_packs', None)
        if reg is None:
            reg = {}
            object.__setattr__(owner, '_eas_dual_packs', reg)
        reg[key] = d
    return _PACK_GEN


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
        if packs is not None and (_PACK_SCOPE is None or packs.get('gen') != _PACK_SCOPE):
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
        global _PACK_SCOPE, _PLANES_SCOPE
        self.prev = (_PACK_SCOPE, _PLANES_SCOPE)
        _PACK_SCOPE = prepack_conv_weights(self.model)
        mods = getattr(self.model, '_eas_modules', None)
        if mods is None:
            mods = list(self.model.modules())
            object.__setattr__(self.model, '_eas_modules', mods)
        # a forward hook anywhere in the model may look at a spike tensor: fp32 spikes throughout then (see "SPIKE PLANES")
        # (hooks that declare ``_eas_planes_safe`` -- the trainer's backward cut, which handles ghosts -- do not count)
        _PLANES_SCOPE = _PACK_SCOPE is not None and not any(_foreign_hooks(m) for m in mods)
        global _INVSTD_SCOPE
        self.prev_invstd = _INVSTD_SCOPE
        if not self.model.training:
            frozen = _FROZEN is not None and _FROZEN['model'] is self.model and not torch.is_grad_enabled()
            _INVSTD_SCOPE = _FROZEN['inv'] if frozen else _invstd_of_eval_model(mods)

    def __exit__(self, *exc):
        global _PACK_SCOPE, _PLANES_SCOPE, _INVSTD_SCOPE
        _INVSTD_SCOPE = self.prev_invstd
        _PACK_SCOPE, _PLANES_SCOPE = self.prev
        if self.prev[0] is None:
            clear_conv_stats()


# Statistics tap (eas_snn_amd/stats.py): when set, every convolution input of the model is shown to it before the
# convolution runs -- ``sink(conv_module, x, replicas)`` for the dense convolutions, and the sampler appends the spike
# tensor entering each micro-step to ``sink.sampler_spikes``.  The kernels that run are the same with or without it.
_CONV_SINK = None


def set_conv_sink(sink):
    global _CONV_SINK
    prev, _CONV_SINK = _CONV_SINK, sink
    return prev


def conv_sink():
    return _CONV_SINK


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
    if (act != 'silu' or torch.is_grad_enabled() or _CONV_SINK is not None or conv._forward_hooks or conv._forward_pre_hooks
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
    pk = packs[0] if (packs is not None and _PACK_SCOPE is not None and packs.get('gen') == _PACK_SCOPE) else conv_pack_weights(conv.weight, 0)
    return conv_fwd_packed(xd, pk, conv.bias, conv.out_channels, k, s_, xt, act=1)


# ------------------------------------------------------------------------------------------------ fused real-valued eval block (conv -> BN -> SiLU, ONE kernel)
# The eval-mode forward of an unconverted BaseConv -- the ANN PAFPN neck and head of use_spike=True models -- normalises with running
# statistics, so the BatchNorm and the activation can run on the convolution's accumulators: eas_conv_bn_act_eval, same arithmetic as
# eas_conv_fwd + eas_bn_silu_fwd_ex (bit-identical), one launch and 4 instead of 12 bytes per output element.  Unlike ``fuse_model`` nothing
# is folded into the weights, and the dual convolutions / in-place concatenations of the unfused model stay.  EAS_FUSED_ANN_EVAL=0: the
# two-kernel path (development).
FUSED_ANN_EVAL = os.environ.get('EAS_FUSED_ANN_EVAL', '1') == '1'


def _ann_eval_bn_ok(bn):
    return (type(bn) is torch.nn.BatchNorm2d and not bn.training and bn.affine and bn.running_mean is not None and bn.running_var is not None
            and bn.num_features % 8 == 0 and not bn._forward_hooks and not bn._forward_pre_hooks)


def fused_ann_eval_ok(x, conv, norms):
    """``conv`` (an nn.Conv2d, or a pair that reads the same input and runs as one) followed by ``norms`` on consecutive output-channel
    ranges and SiLU can run as eas_conv_bn_act_eval on ``x`` [NI,Cin,H,W].  norms[i]: the eval-mode BatchNorm2d behind a bias-free
    convolution, or -- for a block whose BatchNorm ``fuse_model`` folded into the convolution -- that convolution's bias tensor."""
    if not FUSED_ANN_EVAL or torch.is_grad_enabled() or _CONV_SINK is not None or torch.nn.modules.module._global_forward_hooks:
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
        pk = pk[0] if (pk is not None and _PACK_SCOPE is not None and pk.get('gen') == _PACK_SCOPE) else conv_pack_weights(c0.weight, 0)
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
    if _CONV_SINK is not None:
        x = dense(x)
        _CONV_SINK(conv, x, _REPLICAS)
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
    if packs is not None and (_PACK_SCOPE is None or packs.get('gen') != _PACK_SCOPE):
        packs = None            # not inside the forward that made this packing: pack the weight as it is now
    return _ConvFn.apply(x, conv.weight, conv.bias, conv.stride[0], 1 if small_int else 3, packs, planes_of(x) if small_int else None)


# ------------------------------------------------------------------------------------------------ BN step counters
_DEFERRED = None


class deferred_counters:
    """Inside this context ``bump_counter`` only records the ``num_batches_tracked`` tensors; they are all incremented by ONE
    foreach kernel on exit instead of one tiny launch per BatchNorm layer (74 per forward of SYOLOX-S)."""

    def __enter__(self):
        global _DEFERRED
        self.prev, _DEFERRED = _DEFERRED, []

    def __exit__(self, *exc):
        global _DEFERRED
        pending, _DEFERRED = _DEFERRED, self.prev
        if pending:
            torch._foreach_add_(pending, 1)


def bump_counter(t):
    if _DEFERRED is None:
        t.add_(1)
    else:
        _DEFERRED.append(t)


# the other kernel families of the operator layer live in modules of their own; ``ops.<name>`` stays the one public namespace
from .ops_events import *          # noqa: E402,F401,F403  K1 + event representations
from .ops_glue import *            # noqa: E402,F401,F403  SPP, upsample + concatenate, Focus
from .ops_det import *             # noqa: E402,F401,F403  post-processing, SimOTA, detection loss
