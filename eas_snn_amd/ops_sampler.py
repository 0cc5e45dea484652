"""Operator layer, K3: the adaptive sampler (yolox/models/embedding.py:141-226) as one autograd node over eas_smallconv_* /
eas_arsnn_*, and the simpler embeddings on the same kernels.  ``eas_snn_amd.ops`` re-exports everything here (``ops.<name>``)."""
import ctypes as C
import os

import torch

from . import _lib
from ._ctx import ctx as opctx
from ._lib import check, ptr, stream
from .ops_core import READOUT_IDS, _call, _dev, _f32c

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
        fused = bool(opctx.arsnn_fused and d_in == 2 and d_gate in (0, 2) and W % 4 == 0 and pin[2].shape[:2] == (4, 4) and pin[0].shape[0] == 4
                     and (d_gate == 0 or (pg[2].shape[:2] == (4, 4) and pg[0].shape[0] == 4)) and opctx.conv_sink is None and Tm > 0)
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
        fast0 = bool(d_gate) and Tm > 0 and opctx.conv_sink is None and os.environ.get('EAS_ARSNN_STEP0', 'shared') != 'full'
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
            if opctx.conv_sink is not None and d_gate:
                opctx.conv_sink.sampler_spikes.append(spike)
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
