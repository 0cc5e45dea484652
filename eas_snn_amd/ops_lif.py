"""Operator layer, K2: the stand-alone multi-step neuron (spikingjelly call sites: LIFNode / ParametricLIFNode / IFNode on [T,N,...]
tensors) and the mean over T.  ``eas_snn_amd.ops`` re-exports everything here (``ops.<name>``)."""
import ctypes as C
import os

import torch

from . import _lib
from ._ctx import ctx as opctx
from ._lib import check, ptr, stream
from .ops_core import FLAG_DECAY_INPUT, SG_PATAN, SURROGATE_IDS, _alpha_arg, _call, _dev, _f32c, dense

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
        write_v = opctx.state_writeback
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
