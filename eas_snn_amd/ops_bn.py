"""Operator layer, BatchNorm-fused kernels: conv output -> BatchNorm2d('m') -> (P)LIF over T (eas_bn_lif_*), the CSPLayer branch
pair, channel joins, BatchNorm + SiLU of the real-valued blocks (eas_bn_silu_*).  Reference arithmetic: yolox/models/network_blocks.py:52-53
behind yolox/utils/utils_snn.py:16-58.  ``eas_snn_amd.ops`` re-exports everything here (``ops.<name>``)."""
import ctypes as C
import os

import torch

from . import _lib
from ._ctx import ctx as opctx
from ._lib import check, ptr, stream
from .ops_core import SG_PATAN, SURROGATE_IDS, _alpha_arg, _call, _dev, _eval_invstd, _f32c, _take_conv_stats, _timer_add, _timer_mark, bump_counter, dense, ghost, is_small_int, mark_small_int, new_planes, planes_of, to_planes

# ------------------------------------------------------------------------------------------------ K4 (BN + LIF)
# Convolution -> BatchNorm hand-over (eas_conv_fwd_stats, north_star's fused conv -> BN -> LIF step): inside ``conv_stats_scope`` a
# matrix-core convolution also leaves the per-channel sums of its output tile by tile, and the BN kernel that consumes exactly that
# tensor next adds them up instead of reading y once more (no eas_bn_stats_partial launch).  The slot holds the convolution output
# itself, so its address cannot be reused while the slot is valid; a BN call on anything else falls back to the statistics pass.



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
    if opctx.call_log is not None:
        opctx.call_log.append(('eas_bn_stats_partial', (None, y_ctot, TN, Cc, HW, int(replicas))))
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
        write_v = opctx.state_writeback
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


# Number of identical copies the current batch stands for (set by SeqToANNContainer while it runs a stateless block
# ONCE for T identical time steps): batch-norm statistics are unchanged by replication except for the sample count
# used in the unbiased running variance.



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
            float(bn.momentum) if update else None, float(bn.eps), opctx.replicas)


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


# ------------------------------------------------------------------------------------------------ BN step counters
