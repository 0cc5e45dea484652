"""Event -> frame embeddings in front of the network (reference: yolox/models/embedding.py).

``AdaptiveRSNNEmbedding`` ("arsnn") is the adaptive sampling module: a recurrent conv-SNN over the Tm
micro-slices whose spikes decide, per pixel, where one adaptive frame ends and the next begins.  Here the
whole loop is one autograd node over HIP kernels (eas_snn_amd.ops.arsnn_forward); parameter names equal the
reference's (``gate_conv.0.weight``, ``input_conv.2.bias``, ...).  The simpler embeddings ("count", "rsnn",
"snn") are thin tensor programs around the same kernels."""
import copy

import torch
import torch.nn as nn

from eas_snn_amd import ops

from .activation import Rectangle


def _conv_stack(cin, cout, k, depth):
    mods = [nn.Conv2d(cin, cout, k, padding=k // 2)]
    for _ in range(int(depth) - 1):
        mods += [nn.ReLU(inplace=True), nn.Conv2d(cout, cout, k, padding=k // 2)]
    return nn.Sequential(*mods)


def _stack_params(seq):
    out = []
    for m in seq:
        if isinstance(m, nn.Conv2d):
            out += [m.weight, m.bias]
    return out


def _time_major(events):
    """[B,Tl,Tm,2,H,W] | [B,Tm,2,H,W] -> [Tm, N, 2, H, W] with the newest micro-slice first (embedding.py:147-156)."""
    lead = None
    if events.dim() > 5:
        lead = events.shape[:-4]
        events = events.flatten(end_dim=-5)
    # one pass: gather the micro-slices newest first into a contiguous time-major tensor (flip + contiguous would be two)
    Tm = events.shape[1]
    return torch.stack([events[:, Tm - 1 - t] for t in range(Tm)]), lead


def _check_spike_fn(kwargs_spikes):
    fn = kwargs_spikes.get('spike_fn', Rectangle)
    if fn is not Rectangle:
        raise NotImplementedError('the HIP sampler implements the Rectangle spike function '
                                  '(the only one EventExp.get_kwargs_spikes passes)')


class SpikeCountEmbedding(nn.Module):
    def __init__(self, nb_steps):
        super().__init__()
        self.nb_steps = nb_steps

    def forward(self, events):
        if events.dim() < 5:
            return events.unsqueeze(0).expand(self.nb_steps, *events.shape).sum(0)
        ev, _ = _time_major(events)
        return ev.sum(axis=0)


class AdaptiveRSNNEmbedding(nn.Module):
    def __init__(self, kernel_size, in_channel=2, out_channel=2, Ts=1, split=False, spike_attach=False, write_zero=False,
                 abs=False, depth=1, readout='sum', **kwargs_spikes):
        super().__init__()
        _check_spike_fn(kwargs_spikes)
        self.kernel_size, self.Ts, self.abs, self.split, self.readout = kernel_size, Ts, abs, split, readout
        self.write_zero, self.spike_attach = write_zero, spike_attach
        self.kwargs_spikes = kwargs_spikes
        self.nb_steps = kwargs_spikes['Tm'] if 'Tm' in kwargs_spikes else kwargs_spikes['nb_steps']
        self.thresh = kwargs_spikes['thresh']
        self.vreset = copy.deepcopy(kwargs_spikes['vreset'])
        self.depth = int(depth)
        self.gate_conv = _conv_stack(out_channel, out_channel * 2, kernel_size, self.depth)
        self.input_conv = _conv_stack(in_channel, out_channel * 2, kernel_size, self.depth)
        if split:
            # the reference's forward never calls these two (embedding.py:100-102, 141-226): they exist so that parameter counts,
            # optimizer groups and checkpoints written with split=True match
            self.gate_conv_agg = nn.Conv2d(out_channel, out_channel * 2, kernel_size, padding=kernel_size // 2)
            self.input_conv_agg = nn.Conv2d(in_channel, out_channel * 2, kernel_size, padding=kernel_size // 2)
        self._init_weight()

    def _init_weight(self):
        for m in self.input_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain('relu'))
        for m in self.gate_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, nonlinearity='sigmoid')
        if self.split:
            nn.init.kaiming_uniform_(self.input_conv_agg.weight, nonlinearity='sigmoid')
            nn.init.orthogonal_(self.gate_conv_agg.weight, gain=nn.init.calculate_gain('relu'))

    def forward(self, events, record=False, v_record=False):
        if events.dim() < 5:        # parameter-registration passthrough used by get_model_info (embedding.py:144-146)
            return events.unsqueeze(0).expand(self.Ts, *events.shape)
        # the loader's [B(,Tl),Tm,2,H,W] goes to the operator as it is: the kernels of the fused step read it time-major, newest slice first
        # (embedding.py:147-156), other configurations make that copy inside
        ev = events.flatten(end_dim=-5) if events.dim() > 5 else events
        if ev.shape[1] != self.nb_steps:
            raise ValueError(f'expected {self.nb_steps} micro-slices, got {ev.shape[1]}')
        agg, rec, *vrec = ops.arsnn_forward(ev, _stack_params(self.input_conv), _stack_params(self.gate_conv), self.kernel_size,
                                            self.Ts, self.readout, self.spike_attach, self.write_zero, self.abs,
                                            float(self.thresh), None if self.vreset is None else float(self.vreset), record=record, collated=True,
                                            v_record=bool(v_record) and not record)
        if record:
            return agg, rec.long()
        return (agg, vrec[0]) if v_record else agg      # embedding.py:221-226: ``record`` wins over ``v_record``


class _TimeFlat(nn.Module):
    """``tdLayer`` of the reference (yolox/models/layer.py:122-132): the wrapped module is applied with [T, N] folded into
    the batch axis; child name ``layer`` so the keys read ``input_conv.layer.0.weight`` like the reference's."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        T = x.shape[0]
        y = ops.small_conv_stack(x.contiguous().view(-1, *x.shape[2:]), _stack_params(self.layer))
        return y.view(T, -1, *y.shape[1:])


class _DecayCell(nn.Module):
    """Holder of LIFCell's learnable decay logit (yolox/models/cell.py:26-35): key ``cell.decay``."""

    def __init__(self, decay):
        super().__init__()
        self.decay = decay if isinstance(decay, nn.Parameter) else nn.Parameter(torch.as_tensor(float(decay)))


def _steps_newest_first(events, nb_steps):
    """[B,Tl,Tm,2,H,W] | [B,Tm,2,H,W] -> ([Tm,N,2,H,W] newest first, lead dims); a 4-D batch is repeated for every step."""
    if events.dim() < 5:
        return events.unsqueeze(0).expand(nb_steps, *events.shape).contiguous(), None
    return _time_major(events)


class SpikingEmbedding(nn.Module):
    """ "rsnn" (yolox/models/embedding.py:229-316): the sampler's gated recurrence without segments; readout = sum of the
    pre-reset potentials, or the last potential.  One autograd node over the HIP kernels (ops.gated_recurrence)."""

    def __init__(self, kernel_size, in_channel=2, out_channel=2, readout='sum', relu=False, depth=1, **kwargs_spikes):
        super().__init__()
        _check_spike_fn(kwargs_spikes)
        self.kernel_size, self.readout, self.relu, self.depth = kernel_size, readout, relu, int(depth)
        self.kwargs_spikes = kwargs_spikes
        self.nb_steps = kwargs_spikes['Tm'] if 'Tm' in kwargs_spikes else kwargs_spikes['nb_steps']
        self.thresh = kwargs_spikes['thresh']
        self.vreset = copy.deepcopy(kwargs_spikes['vreset'])
        self.input_conv = _TimeFlat(_conv_stack(in_channel, out_channel * 2, kernel_size, self.depth))
        self.gate_conv = _conv_stack(out_channel, out_channel * 2, kernel_size, self.depth)
        for m in self.input_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain('relu'))
        for m in self.gate_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, nonlinearity='sigmoid')

    def forward(self, events):
        ev, lead = _steps_newest_first(events, self.nb_steps)
        out = ops.gated_recurrence(ev, _stack_params(self.input_conv.layer), _stack_params(self.gate_conv), self.kernel_size,
                                   self.readout, self.relu, float(self.thresh), None if self.vreset is None else float(self.vreset))
        if self.readout == 'last' and lead is not None:          # embedding.py:304-306: 'last' is un-flattened
            out = out.view(tuple(lead) + tuple(out.shape[1:])).transpose(0, 1)
        return out


class LIFEmbedding(nn.Module):
    """ "snn" (yolox/models/embedding.py:28-76 + LIFCell.forward, cell.py:37-65): psp = conv stack per micro-slice;
    v = sigmoid(decay)*v + psp, fire (> thresh), reset; readout = sum of the pre-reset potentials or the last potential.
    The recurrence runs on the sampler's step kernels with a constant gate plane and no recurrent convolution."""

    def __init__(self, kernel_size, in_channel=2, out_channel=2, readout='sum', depth=1, **kwargs_spikes):
        super().__init__()
        _check_spike_fn(kwargs_spikes)
        self.kernel_size, self.readout, self.depth = kernel_size, readout, int(depth)
        self.nb_steps = kwargs_spikes['Tm'] if 'Tm' in kwargs_spikes else kwargs_spikes['nb_steps']
        self.thresh = kwargs_spikes['thresh']
        self.vreset = copy.deepcopy(kwargs_spikes['vreset'])
        self.embedding_conv = _TimeFlat(_conv_stack(in_channel, out_channel, kernel_size, self.depth))
        self.cell = _DecayCell(kwargs_spikes['decay'])
        for m in self.embedding_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.orthogonal_(m.weight, gain=nn.init.calculate_gain('relu'))

    def forward(self, events):
        ev, _ = _steps_newest_first(events, self.nb_steps)
        psp = self.embedding_conv(ev)                                        # [Tm, N, C2, H, W]
        x = torch.cat([self.cell.decay.expand_as(psp), psp], dim=2)          # gate pre-activation | current
        return ops.gated_recurrence(x, [], [], self.kernel_size, self.readout, False, float(self.thresh),
                                    None if self.vreset is None else float(self.vreset))
