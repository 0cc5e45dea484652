"""Glue kernels of the network: SPP pooling block, fused upsample + concatenate of the PAFPN, Focus
(reference: yolox/models/network_blocks.py:143-147, 198-213; yolo_pafpn.py:88-116).

Part of the operator layer of ``eas_snn_amd.ops`` (split by kernel family; ``ops`` re-exports everything here, so ``ops.<name>`` keeps working)."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, ptr, stream
from .ops_core import _call, _dev, _f32c, _timer_add, _timer_mark, dense, ghost, is_small_int, mark_small_int, planes_enabled, planes_of


# ------------------------------------------------------------------------------------------------ SPP pooling block
class _SPPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ks):
        _dev(x)
        x = _f32c(x)
        lead, (Cc, H, W) = x.shape[:-3], x.shape[-3:]
        N = x.numel() // (Cc * H * W)
        out = torch.empty(lead + (4 * Cc, H, W), dtype=torch.float32, device=x.device)
        _call('eas_spp_pool_fwd', 4 * 5 * x.numel(), _lib.lib().eas_spp_pool_fwd, ptr(x), ptr(out), N, Cc, H, W, ks[0], ks[1], ks[2], stream())
        ctx.save_for_backward(x)
        ctx.ks = ks
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _f32c(g)
        Cc, H, W = x.shape[-3:]
        N = x.numel() // (Cc * H * W)
        gx = torch.empty_like(x)
        ks = ctx.ks
        _call('eas_spp_pool_bwd', 4 * 6 * x.numel(), _lib.lib().eas_spp_pool_bwd, ptr(x), ptr(g), ptr(gx), N, Cc, H, W, ks[0], ks[1], ks[2],
              stream())
        return gx, None


class _SPPPlanesFn(torch.autograd.Function):
    """the SPP block on spike planes: ghost in (its planes), ghost out -- no fp32 copy of the spikes exists (eas_spp_pool_planes_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, x, sp, ks):
        lead, (Cc, H, W) = x.shape[:-3], x.shape[-3:]
        N = 1
        for d in lead:
            N *= d
        out_sp = torch.empty(tuple(lead) + (4 * Cc // 8, H * W, 8), dtype=torch.bfloat16, device=sp.device)
        _call('eas_spp_pool_fwd', 2 * 5 * N * Cc * H * W, _lib.lib().eas_spp_pool_planes_fwd, ptr(sp), ptr(out_sp), N, Cc, H, W, ks[0], ks[1], ks[2],
              stream())
        ctx.save_for_backward(sp)
        ctx.cfg = (ks, tuple(x.shape), N)
        ctx.mark_non_differentiable(out_sp)
        return ghost(tuple(lead) + (4 * Cc, H, W), sp.device), out_sp

    @staticmethod
    def backward(ctx, g, _g_sp):
        (sp,) = ctx.saved_tensors
        ks, shape, N = ctx.cfg
        Cc, H, W = shape[-3:]
        g = _f32c(g)
        gx = torch.empty(shape, dtype=torch.float32, device=g.device)
        _call('eas_spp_pool_bwd', 4 * 5 * gx.numel() + 2 * gx.numel(), _lib.lib().eas_spp_pool_planes_bwd, ptr(sp), ptr(g), ptr(gx), N, Cc, H, W,
              ks[0], ks[1], ks[2], stream())
        return gx, None, None


def spp_pool_supported(x, ks):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() in (4, 5) and len(ks) == 3 and all(k % 2 == 1 for k in ks) and (
        x.shape[-1] * x.shape[-2] <= 1024)


def spp_pool_cat(x, ks):
    """cat[x, maxpool_k(x) for k in ks] along the channel axis of x [..., C, H, W] in one kernel (and one for the backward).  A ghost (spike
    planes) stays one: packed 16-bit maxima over 16-byte pixels, and the backward reads x from the planes."""
    sp = planes_of(x)
    if sp is not None and x.shape[-3] % 8 == 0 and os.environ.get('EAS_SPP_PLANES', '1') == '1':
        out, out_sp = _SPPPlanesFn.apply(x, sp.contiguous(), tuple(int(k) for k in ks))
        out._eas_sp = out_sp
        out._eas_small_int = True
        return out
    x = dense(x)
    out = _SPPFn.apply(x, tuple(int(k) for k in ks))
    if is_small_int(x):
        mark_small_int(out)
    return out


# ------------------------------------------------------------------------------------------------ neck glue
class _UpcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, up):
        _dev(a, b)
        a, b = _f32c(a), _f32c(b)
        lead, (Ca, H, W) = a.shape[:-3], a.shape[-3:]
        Cb = b.shape[-3]
        M = a.numel() // (Ca * H * W)
        out = torch.empty(lead + (Ca + Cb, H * up, W * up), dtype=torch.float32, device=a.device)
        check(_lib.lib().eas_upcat_fwd(ptr(a), ptr(b), ptr(out), M, Ca, Cb, H, W, up, stream()), 'eas_upcat_fwd')
        ctx.cfg = (a.shape, b.shape, M, Ca, Cb, H, W, up)
        return out

    @staticmethod
    def backward(ctx, g):
        ashape, bshape, M, Ca, Cb, H, W, up = ctx.cfg
        g = _f32c(g)
        ga = torch.empty(ashape, dtype=torch.float32, device=g.device)
        gb = torch.empty(bshape, dtype=torch.float32, device=g.device)
        check(_lib.lib().eas_upcat_bwd(ptr(g), ptr(ga), ptr(gb), M, Ca, Cb, H, W, up, stream()), 'eas_upcat_bwd')
        return ga, gb, None


class _UpcatPlanesFn(torch.autograd.Function):
    """upsample + concatenate on spike planes: ghosts in, a ghost out (eas_upcat_planes_fwd); the gradient is fp32 (eas_upcat_bwd)"""

    @staticmethod
    def forward(ctx, a, b, a_sp, b_sp, up):
        ctx.set_materialize_grads(False)
        lead, (Ca, H, W) = tuple(a.shape[:-3]), a.shape[-3:]
        Cb = b.shape[-3]
        M = 1
        for d in lead:
            M *= d
        Ho, Wo = H * up, W * up
        sp = torch.empty(lead + ((Ca + Cb) // 8, Ho * Wo, 8), dtype=torch.bfloat16, device=a_sp.device)
        _call('eas_upcat', 2 * 2 * sp.numel(), _lib.lib().eas_upcat_planes_fwd, ptr(a_sp), ptr(b_sp), ptr(sp), M, Ca, Cb, H, W, up, stream())
        ctx.cfg = (tuple(a.shape), tuple(b.shape), M, Ca, Cb, H, W, up)
        ctx.mark_non_differentiable(sp)
        return ghost(lead + (Ca + Cb, Ho, Wo), a_sp.device), sp

    @staticmethod
    def backward(ctx, g, _g_sp):
        if g is None:
            return None, None, None, None, None
        ashape, bshape, M, Ca, Cb, H, W, up = ctx.cfg
        g = _f32c(g)
        ga = torch.empty(ashape, dtype=torch.float32, device=g.device)
        gb = torch.empty(bshape, dtype=torch.float32, device=g.device)
        check(_lib.lib().eas_upcat_bwd(ptr(g), ptr(ga), ptr(gb), M, Ca, Cb, H, W, up, stream()), 'eas_upcat_bwd')
        return ga, gb, None, None, None


def upcat_supported(a, b, up):
    if os.environ.get('EAS_NO_UPCAT'):         # development switch
        return False
    return (a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float32 and a.dim() == b.dim() and a.dim() >= 4
            and a.shape[:-3] == b.shape[:-3] and b.shape[-2] == a.shape[-2] * up and b.shape[-1] == a.shape[-1] * up
            and (a.shape[-1] * up) % 4 == 0 and a.shape[-1] % 2 == 0)


def upsample_cat(a, b, up=2):
    """cat[nearest-upsample(a, x up), b] along channels in one kernel (one more for the backward); up = 1: plain concatenation"""
    a_sp, b_sp = planes_of(a), planes_of(b)
    if a_sp is not None and b_sp is not None and planes_enabled():
        out, sp = _UpcatPlanesFn.apply(a, b, a_sp.contiguous(), b_sp.contiguous(), int(up))
        out._eas_sp = sp
        return mark_small_int(out)
    a, b = dense(a), dense(b)
    out = _UpcatFn.apply(a, b, int(up))
    if is_small_int(a) and is_small_int(b):       # copies of spikes / small integers (nearest upsampling repeats values)
        mark_small_int(out)
    return out


class _FocusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _f32c(x)
        M, Cc, H, W = x.shape
        out = torch.empty((M, 4 * Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
        check(_lib.lib().eas_focus(ptr(x), ptr(out), M, Cc, H // 2, W // 2, 0, stream()), 'eas_focus')
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        M, C4, Ho, Wo = g.shape
        gx = torch.empty((M, C4 // 4, 2 * Ho, 2 * Wo), dtype=torch.float32, device=g.device)
        check(_lib.lib().eas_focus(ptr(g), ptr(gx), M, C4 // 4, Ho, Wo, 1, stream()), 'eas_focus')
        return gx


def focus_supported(x):
    if os.environ.get('EAS_NO_FOCUS'):         # development switch
        return False
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-2] % 2 == 0 and x.shape[-1] % 4 == 0


def focus(x):
    """space to depth of Focus.forward: cat(x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]) in one kernel"""
    return _FocusFn.apply(x)
