"""Operator layer, shared part: the call wrapper over the C ABI (status check, HIP-event timing, call trace), the spike-plane / ghost
tensor machinery, the "small integer" tags and their device-side check, the convolution-epilogue statistics slot, BatchNorm step counters.
The switches and scopes live in ``eas_snn_amd._ctx.ctx``.  ``eas_snn_amd.ops`` re-exports everything here (``ops.<name>``)."""
import ctypes as C
import os

import torch

from . import _lib
from ._ctx import ctx as opctx
from ._lib import check, ptr, stream

SURROGATE_IDS = {'atan': 0, 'sigmoid': 1, 'rect': 2, 'patan': 3}
SG_PATAN = 3      # learnable slope: ``alpha`` is a device tensor of one element and receives a gradient
FLAG_HARD_RESET, FLAG_DECAY_INPUT, FLAG_DETACH_RESET, FLAG_FIRE_STRICT = 1, 2, 4, 8
READOUT_IDS = {'sum': 0, 'last': 1, 'avg': 2}

# Write the final membrane potential back after every multi-step call (spikingjelly semantics: ``node.v``
# holds the state until ``reset_net``).  Training/eval loops that reset after every batch (the reference's
# do: yolox/core/trainer.py:115-117, yolox/evaluators/event_evaluator.py:196-198) can switch this off and
# save 4 B per neuron per layer; results are identical because the state is discarded by the reset.



def set_state_writeback(flag):
    opctx.state_writeback = bool(flag)


def state_writeback():
    return opctx.state_writeback


class state_writeback_scope:
    """``with ops.state_writeback_scope(flag):`` -- the setting for the duration of the block, the previous one restored on exit."""

    def __init__(self, flag):
        self.flag = bool(flag)

    def __enter__(self):
        self.prev, opctx.state_writeback = opctx.state_writeback, self.flag
        return self

    def __exit__(self, *exc):
        opctx.state_writeback = self.prev


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
        if opctx.tag is not None:                      # development: per-layer breakdown (scripts/layer_times.py)
            self.tagged.setdefault((opctx.tag, name), []).append((start, end, nbytes, flops, issue_flops))

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


def set_tag(tag):
    prev, opctx.tag = opctx.tag, tag
    return prev


def set_timer(timer):
    opctx.timer = timer


class kernel_trace:
    """``with ops.kernel_trace() as tr:`` -- afterwards ``tr.kernels`` holds the distinct device kernel symbols the library launched inside
    the block (the names rocprofv3 reports; eas_kernel_trace_begin / _dump) and ``tr.calls`` the C-ABI calls made through the operators
    with their arguments.  Test infrastructure: tests/test_gpu_bench_shapes.py replays every convolution geometry of a bench step against
    fp64 and proves that every kernel instance of the step is one an oracle-compared test has launched."""

    def __enter__(self):
        self.prev, opctx.call_log = opctx.call_log, []
        self.calls = opctx.call_log
        _lib.lib().eas_kernel_trace_begin()
        return self

    def __exit__(self, *exc):
        opctx.call_log = self.prev
        L = _lib.lib()
        need = L.eas_kernel_trace_dump(None, 0)
        buf = C.create_string_buffer(int(need))
        L.eas_kernel_trace_dump(buf, need)
        self.kernels = sorted(k for k in buf.value.decode().split('\n') if k)


def _call(name, nbytes, fn, *args, flops=0.0, issue_flops=0.0):
    """Invoke one C-ABI entry point (optionally bracketed by HIP events) and check its status."""
    if opctx.call_log is not None:
        opctx.call_log.append((fn.__name__, args))
    if opctx.timer is None:
        check(fn(*args), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    check(fn(*args), name)
    e.record()
    opctx.timer.add(name, s, e, nbytes, flops, issue_flops)


def _timer_mark():
    """start event of a timed region that is not a single _call (None when no timer is installed)"""
    if opctx.timer is None:
        return None
    s = torch.cuda.Event(enable_timing=True)
    s.record()
    return s


def _timer_add(name, start, nbytes):
    if start is not None and opctx.timer is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        opctx.timer.add(name, start, e, nbytes)


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

_GHOST_BASE = {}


def planes_enabled():
    """spike planes may be produced right now: switched on, inside a model's own forward (``packed_weights`` -- a block called on its own
    hands out fp32 spikes), and nothing that wants to see fp32 tensors is attached (module forward hooks, a statistics tap)"""
    return opctx.spike_planes and opctx.planes_scope and opctx.conv_sink is None and not torch.nn.modules.module._global_forward_hooks


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



def _eval_invstd(running_var, eps):
    if opctx.invstd_scope is not None:
        hit = opctx.invstd_scope.get(running_var.data_ptr())
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
_STATS_BLOCKS = {}


class conv_stats_scope:
    """``with ops.conv_stats_scope(bn_uses_batch_statistics):`` around the convolution whose output goes straight into a fused BN kernel"""

    def __init__(self, flag=True):
        self.flag = bool(flag) and opctx.conv_stats

    def __enter__(self):
        self.prev, opctx.want_conv_stats = opctx.want_conv_stats, self.flag

    def __exit__(self, *exc):
        opctx.want_conv_stats = self.prev


def _conv_stats_blocks(L, geom):
    nb = _STATS_BLOCKS.get(geom)
    if nb is None:
        nb = _STATS_BLOCKS[geom] = L.eas_conv_fwd_stats_blocks(*geom)
    return nb


def _take_conv_stats(base_ptr, count, Ctot, keep=False):
    """(stats tensor, nb) if the tensor at ``base_ptr`` ([.., Ctot, H, W] with ``count`` elements per channel) is the output of the last
    convolution that summed its tiles, else None; the slot is cleared unless ``keep`` (two consumers of one output)."""
    slot = opctx.conv_stats_slot
    if slot is None:
        return None
    y, nb, stats, ver = slot
    if not keep:
        opctx.conv_stats_slot = None
    if y.data_ptr() != base_ptr or y.shape[1] != Ctot or y.shape[0] * y.shape[2] * y.shape[3] != count or y._version != ver:
        return None                    # another tensor, or y was modified in place since the convolution summed it
    return stats, nb


def clear_conv_stats():
    """drop a convolution's tile sums nobody consumed (end of a model forward: the slot must not pin tensors across steps)"""
    opctx.conv_stats_slot = None


class replicated:
    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev, opctx.replicas = opctx.replicas, self.n

    def __exit__(self, *exc):
        opctx.replicas = self.prev

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


# One backend: every dense 1x1 / 3x3 convolution of the step -- forward, input gradient, weight gradient -- runs on the
# hand-written matrix-core kernels (eas_conv_fwd / eas_conv_dgrad_s2 / eas_conv_wgrad).  The library convolution remains only for
# geometries those kernels do not cover at all (grouped / dilated / odd-width / other kernel sizes: none in the EAS-SNN models).



def mark_small_int(t):
    """Tag a tensor whose values are spikes or sums of a few spikes (exact in bf16): convolutions reading it use one
    bf16 term instead of three.  Untagged tensors always take the general three-term path."""
    t._eas_small_int = True
    return t


def is_small_int(t):
    return getattr(t, '_eas_small_int', False)


def _verify_tags(x, small_int):
    """test-suite check (opctx.verify_small_int, host sync): a tagged tensor is exact in bf16; a ghost's planes hold small integers"""
    if not (small_int and opctx.verify_small_int) or torch.cuda.is_current_stream_capturing():     # (a capture cannot read the device; the
        return                                                                                # eager iterations in front of it did)
    sp = planes_of(x)
    if sp is not None:
        v = sp.float()
        if not bool(((v == v.round()) & (v >= 0) & (v <= 255)).all()):
            raise _lib.EasHipError('the spike planes of a tensor do not hold small integers')
        return
    if not bool((x == x.to(torch.bfloat16).to(torch.float32)).all()):
        raise _lib.EasHipError('a tensor tagged as spikes / small integers is not exact in bf16')


class deferred_counters:
    """Inside this context ``bump_counter`` only records the ``num_batches_tracked`` tensors; they are all incremented by ONE
    foreach kernel on exit instead of one tiny launch per BatchNorm layer (74 per forward of SYOLOX-S)."""

    def __enter__(self):
        self.prev, opctx.deferred_counters = opctx.deferred_counters, []

    def __exit__(self, *exc):
        pending, opctx.deferred_counters = opctx.deferred_counters, self.prev
        if pending:
            torch._foreach_add_(pending, 1)


def bump_counter(t):
    if opctx.deferred_counters is None:
        t.add_(1)
    else:
        opctx.deferred_counters.append(t)
