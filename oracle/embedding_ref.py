"""TEST INFRASTRUCTURE -- torch-CPU restatement of the adaptive event-sampling
module and its simpler siblings.

Pinned: tests/golden/arsnn_*.npz etc. hold inputs, weights, outputs and
gradients produced by the real reference classes
(/root/reference/yolox/models/embedding.py) -- see oracle/gen_golden.py.

The reference walks the fired positions with ``nonzero`` + advanced indexing
(embedding.py:181-197); the restatement below is the equivalent dense masked
update: every element adds into ``agg[seg]`` at most once per step, in step
order, so the fp32 sums are formed in the same order and the result is
bit-identical.  The global early exit (embedding.py:200-201) only skips steps
whose writes would all be masked by ``seg < Ts``, except that it freezes
``spike_last`` for the tail write -- reproduced by the same ``break``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class RectangleFn(torch.autograd.Function):
    """activation.py:17-30: fwd (x > 0); bwd g * [|x| < 0.5/alpha] * alpha, alpha = 1."""
    alpha = 1.0

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.gt(0).float()

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return g * ((x.abs() < 0.5 / RectangleFn.alpha).float() * RectangleFn.alpha)


def build_conv_stack(cin, cout, k, depth):
    """embedding.py:106-111: Conv(cin->cout,k,p=k//2) [+ ReLU + Conv(cout->cout)]*(depth-1)."""
    mods = [nn.Conv2d(cin, cout, k, padding=k // 2)]
    for _ in range(depth - 1):
        mods.append(nn.ReLU(inplace=True))
        mods.append(nn.Conv2d(cout, cout, k, padding=k // 2))
    return nn.Sequential(*mods)


class _TimeFlat(nn.Module):
    """tdLayer (yolox/models/layer.py:122-132): fold [T, N] into the batch axis around ``layer``
    (child name ``layer`` kept so state_dict keys match: ``input_conv.layer.0.weight``)."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        T = x.shape[0]
        y = self.layer(x.contiguous().view(-1, *x.shape[2:]))
        return y.view(T, -1, *y.shape[1:])


class _DecayCell(nn.Module):
    """Holder for LIFCell's learnable ``decay`` (cell.py:26-35) so the key is ``cell.decay``."""

    def __init__(self, decay):
        super().__init__()
        self.decay = nn.Parameter(torch.tensor(math.log(decay / (1 - decay))))   # warp_decay, util.py:278-280


def _to_time_major_reversed(events):
    """[B,Tl,Tm,2,H,W] or [B,Tm,2,H,W] -> [Tm, N, 2, H, W], newest micro-slice first
    (embedding.py:147-156)."""
    if events.dim() > 5:
        events = events.flatten(end_dim=-5)
    events = events.transpose(0, 1)
    return torch.flip(events, dims=[0])


class AdaptiveRSNNEmbeddingRef(nn.Module):
    """embedding.py:79-226 ("arsnn")."""

    def __init__(self, kernel_size, in_channel=2, out_channel=2, Ts=1, spike_attach=False, write_zero=False,
                 abs=False, depth=1, readout='sum', Tm=4, thresh=1.0, vreset=0.0):
        super().__init__()
        self.Ts, self.abs, self.readout = Ts, abs, readout
        self.spike_attach, self.write_zero = spike_attach, write_zero
        self.nb_steps, self.thresh, self.vreset = Tm, thresh, vreset
        self.depth = int(depth)
        self.gate_conv = build_conv_stack(out_channel, out_channel * 2, kernel_size, self.depth)
        self.input_conv = build_conv_stack(in_channel, out_channel * 2, kernel_size, self.depth)
        # init (embedding.py:121-130)
        for m in self.input_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain('relu'))
        for m in self.gate_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, nonlinearity='sigmoid')

    def forward(self, events, record=False, v_record=False):
        if events.dim() < 5:  # parameter-registration passthrough (embedding.py:144-146)
            ev, _ = torch.broadcast_tensors(events, torch.zeros((self.Ts,) + events.shape))
            return ev
        ev = _to_time_major_reversed(events)
        v_list = []
        Tm = self.nb_steps
        spike = torch.zeros_like(ev[0])
        v = torch.zeros_like(ev[0])
        vsum = torch.zeros_like(ev[0])
        seg = torch.zeros_like(ev[0], dtype=torch.long)
        t_last = torch.full_like(seg, -1)
        agg = [torch.zeros_like(ev[0]) for _ in range(self.Ts)]
        t_record = []
        for t in range(Tm):
            g_rec, c_rec = self.gate_conv(spike).chunk(2, dim=-3)
            g_in, c_in = self.input_conv(ev[t]).chunk(2, dim=-3)
            gate = torch.sigmoid(g_in + g_rec)
            cur = c_in + c_rec
            vn = gate * v + cur                                    # :133
            spike = RectangleFn.apply(vn - self.thresh)            # :134 (strict >)
            if self.vreset is None:
                v = vn - self.thresh * spike
            else:
                v = vn * (1 - spike) + self.vreset * spike         # :135-138
            vsum = vsum + vn                                       # :179
            v_list.append(vn[(1 - spike).bool()])                  # :180 (returned with v_record=True)
            fired = spike.bool()
            if self.readout == 'sum':
                val = vsum
            elif self.readout == 'last':
                val = v
            elif self.readout == 'avg':
                val = vsum / (t - t_last).clamp(min=1).to(vsum.dtype)
            else:
                raise NotImplementedError(self.readout)
            if self.spike_attach:
                val = val * spike                                  # :192-193 (SAT)
            for k in range(self.Ts):
                m = fired & (seg == k)
                agg[k] = agg[k] + torch.where(m, val, torch.zeros_like(val))   # :194
            live = fired & (seg < self.Ts)                         # spike_pos is filtered by seg < Ts first (:183-184)
            seg = seg + live.long()                                # :195
            t_last = torch.where(live, torch.full_like(t_last, t), t_last)    # :196
            vsum = torch.where(fired, torch.zeros_like(vsum), vsum)           # :197 (all fired positions)
            if record:
                t_record.append(t_last.clone())
            if int(seg.min()) >= self.Ts:                          # :200-201
                break
        # tail (embedding.py:203-217)
        quiet = ~spike.bool()
        if self.readout == 'sum':
            val = vsum
        elif self.readout == 'last':
            val = v
        else:
            val = vsum / (Tm - 1 - t_last).clamp(min=1).to(vsum.dtype)   # masked-out lanes stay finite
        if self.write_zero:
            val = val * 0                                          # RPD
        for k in range(self.Ts):
            m = quiet & (seg == k)
            agg[k] = agg[k] + torch.where(m, val, torch.zeros_like(val))
        out = torch.stack(agg)
        if self.abs:
            out = F.relu(out)
        if record:
            return out, torch.stack(t_record)
        return (out, torch.cat(v_list)) if v_record else out       # :221-226


class SpikeCountEmbeddingRef(nn.Module):
    """embedding.py:9-24 ("count"): sum over the micro-slice axis."""

    def __init__(self, nb_steps):
        super().__init__()
        self.nb_steps = nb_steps

    def forward(self, events):
        if events.dim() < 5:
            events, _ = torch.broadcast_tensors(events, torch.zeros((self.nb_steps,) + events.shape))
        elif events.dim() > 5:
            events = events.flatten(end_dim=-5).transpose(0, 1)
        else:
            events = events.transpose(0, 1)
        return events.sum(axis=0)


class SpikingEmbeddingRef(nn.Module):
    """embedding.py:229-316 ("rsnn"): same gate/update as arsnn, readout = sum of pre-reset v or last v."""

    def __init__(self, kernel_size, in_channel=2, out_channel=2, readout='sum', relu=False, depth=1,
                 Tm=4, thresh=1.0, vreset=0.0):
        super().__init__()
        self.readout, self.relu, self.nb_steps = readout, relu, Tm
        self.thresh, self.vreset = thresh, vreset
        self.input_conv = _TimeFlat(build_conv_stack(in_channel, out_channel * 2, kernel_size, depth))
        self.gate_conv = build_conv_stack(out_channel, out_channel * 2, kernel_size, depth)
        for m in self.input_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain('relu'))
        for m in self.gate_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, nonlinearity='sigmoid')

    def forward(self, events):
        lead = None
        if events.dim() < 5:
            events, _ = torch.broadcast_tensors(events, torch.zeros((self.nb_steps,) + events.shape))
            ev = torch.flip(events, dims=[0])
        else:
            if events.dim() > 5:
                lead = events.shape[:-4]                           # (B, Tl)
            ev = _to_time_major_reversed(events)
        gs_in, cs_in = self.input_conv(ev).chunk(2, dim=-3)
        spike = torch.zeros_like(gs_in[0])
        v = torch.zeros_like(gs_in[0])
        vsum = 0
        for t in range(self.nb_steps):
            g_rec, c_rec = self.gate_conv(spike).chunk(2, dim=-3)
            gate = torch.sigmoid(gs_in[t] + g_rec)
            cur = cs_in[t] + c_rec
            vn = gate * v + cur
            spike = RectangleFn.apply(vn - self.thresh)
            if self.vreset is None:
                v = vn - self.thresh * spike
            else:
                v = vn * (1 - spike) + self.vreset * spike
            vsum = vsum + vn
        if self.readout not in ('sum', 'last'):
            raise NotImplementedError(self.readout)
        if self.readout == 'sum':
            out = vsum
        else:                                                      # embedding.py:304-306: 'last' is un-flattened
            out = v.view(lead + v.shape[1:]).transpose(0, 1) if lead is not None else v
        return F.relu(out) if self.relu else out


class LIFEmbeddingRef(nn.Module):
    """embedding.py:28-76 ("snn") + LIFCell.forward (cell.py:37-65):
    v = sigmoid(decay) * v + psp; fire Rectangle(v - thresh); reset; readout sum of pre-reset v | last v."""

    def __init__(self, kernel_size, in_channel=2, out_channel=2, readout='sum', depth=1,
                 Tm=4, thresh=1.0, vreset=0.0, decay=0.5):
        super().__init__()
        self.readout, self.nb_steps = readout, Tm
        self.thresh, self.vreset = thresh, vreset
        self.embedding_conv = _TimeFlat(build_conv_stack(in_channel, out_channel, kernel_size, depth))
        self.cell = _DecayCell(decay)
        for m in self.embedding_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain('relu'))

    def forward(self, events):
        if events.dim() < 5:
            events, _ = torch.broadcast_tensors(events, torch.zeros((self.nb_steps,) + events.shape))
            ev = torch.flip(events, dims=[0])
        else:
            ev = _to_time_major_reversed(events)
        psp = self.embedding_conv(ev)
        v = 0
        vsum = 0
        for t in range(self.nb_steps):
            vn = torch.sigmoid(self.cell.decay) * v + psp[t]
            spike = RectangleFn.apply(vn - self.thresh)
            if self.vreset is None:
                v = vn - self.thresh * spike
            else:
                v = vn * (1 - spike) + self.vreset * spike
            vsum = vsum + vn
        if self.readout == 'sum':
            return vsum
        if self.readout == 'last':
            return v
        raise NotImplementedError(self.readout)
