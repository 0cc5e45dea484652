"""Spiking neurons with the spikingjelly multi-step API, running on libeas_hip.so.

Replaces ``spikingjelly.activation_based.neuron.{BaseNode, IFNode, LIFNode, ParametricLIFNode}`` as built
at yolox/utils/utils_snn.py:44-53.  ``backend='torch'`` (what the reference passes) and ``backend='hip'``
both select the HIP kernels: there is no other implementation in this package and CPU tensors raise.

Semantics kept: charge -> fire (Heaviside ``>=``) -> reset per step; ``v`` starts
as a float, becomes a tensor lazily, persists across calls until ``reset()``; ``w`` is the only parameter
of ParametricLIFNode (``state_dict`` key ``...act.w``); memories are not in ``state_dict``.
The membrane state carried between calls is a constant for autograd (the reference resets after every
iteration, so no gradient ever flows through it).
"""
import math

import torch
import torch.nn as nn

from eas_snn_amd import ops

from . import base, surrogate


class BaseNode(base.MemoryModule):
    def __init__(self, v_threshold: float = 1., v_reset: float = 0., surrogate_function=None, detach_reset: bool = False,
                 step_mode='s', backend='torch', store_v_seq: bool = False):
        assert isinstance(v_reset, float) or v_reset is None
        assert isinstance(v_threshold, float)
        assert isinstance(detach_reset, bool)
        super().__init__()
        self.register_memory('v', 0. if v_reset is None else v_reset)
        self.v_threshold = v_threshold
        self.v_reset = v_reset
        self.detach_reset = detach_reset
        self.surrogate_function = surrogate_function if surrogate_function is not None else surrogate.Sigmoid()
        self.step_mode = step_mode
        self.backend = backend
        self.store_v_seq = store_v_seq
        if store_v_seq:
            raise NotImplementedError('store_v_seq is not provided by the HIP neuron kernels')

    # ---- what the kernels need to know about this neuron
    def _decay(self):
        """-> (w_logit tensor | None, k_const, decay_input)"""
        raise NotImplementedError

    def _flags(self, decay_input):
        f = 0
        if self.v_reset is not None:
            f |= ops.FLAG_HARD_RESET
        if decay_input:
            f |= ops.FLAG_DECAY_INPUT
        if self.detach_reset:
            f |= ops.FLAG_DETACH_RESET
        return f

    def _surrogate(self):
        sf = self.surrogate_function
        sid = getattr(sf, 'hip_id', None)
        if sid is None:
            raise NotImplementedError(f'surrogate {type(sf).__name__} has no HIP backward; use surrogate.ATan / Sigmoid / Rect')
        if sid == 'patan':
            return sid, sf.alpha              # learnable device scalar: handed to the kernels as a pointer, gradient comes back
        return sid, float(sf.alpha)

    def _v_in(self, x0):
        """Tensor state carried from a previous call, or None for the reset value (no HBM read)."""
        v = self.v
        if isinstance(v, torch.Tensor):
            if v.shape != x0.shape:
                raise ValueError(f'membrane state {tuple(v.shape)} does not match input {tuple(x0.shape)}; call reset()')
            return v.detach()
        rv = 0. if self.v_reset is None else self.v_reset
        return None if v == rv else torch.full_like(x0, v)

    def lif_args(self):
        w, k, di = self._decay()
        sid, alpha = self._surrogate()
        return dict(w=w, k_const=k, v_th=self.v_threshold, v_reset=0. if self.v_reset is None else self.v_reset,
                    flags=self._flags(di), surrogate=sid, alpha=alpha)

    def v_float_to_tensor(self, x):
        if isinstance(self.v, float):
            self.v = torch.full_like(x.data, self.v)

    def multi_step_forward(self, x_seq, want_mean=False):
        a = self.lif_args()
        spikes, v_out, mean = ops.lif_multistep(x_seq, self._v_in(x_seq[0]), a['w'], a['k_const'], a['v_th'], a['v_reset'],
                                                a['flags'], a['surrogate'], a['alpha'], want_mean=want_mean)
        if v_out is not None:
            self.v = v_out
        ops.mark_small_int(spikes)          # 0/1 by construction (convolutions reading it use one exact bf16 term)
        return (spikes, mean) if want_mean else spikes

    def single_step_forward(self, x):
        return self.multi_step_forward(x.unsqueeze(0))[0]

    def extra_repr(self):
        return (f'v_threshold={self.v_threshold}, v_reset={self.v_reset}, detach_reset={self.detach_reset}, '
                f'step_mode={self.step_mode}, backend={self.backend}')


class IFNode(BaseNode):
    def _decay(self):
        return None, 0.0, False       # h = v * (1 - 0) + x


class LIFNode(BaseNode):
    def __init__(self, tau: float = 2., decay_input: bool = True, v_threshold: float = 1., v_reset: float = 0.,
                 surrogate_function=None, detach_reset: bool = False, step_mode='s', backend='torch', store_v_seq: bool = False):
        assert isinstance(tau, float) and tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.tau = tau
        self.decay_input = decay_input

    def _decay(self):
        return None, 1. / self.tau, self.decay_input

    def extra_repr(self):
        return super().extra_repr() + f', tau={self.tau}'


class ParametricLIFNode(BaseNode):
    def __init__(self, init_tau: float = 2.0, decay_input: bool = True, v_threshold: float = 1., v_reset: float = 0.,
                 surrogate_function=None, detach_reset: bool = False, step_mode='s', backend='torch', store_v_seq: bool = False):
        assert isinstance(init_tau, float) and init_tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.decay_input = decay_input
        self.w = nn.Parameter(torch.as_tensor(-math.log(init_tau - 1.)))

    def _decay(self):
        return self.w, 0.0, self.decay_input

    def extra_repr(self):
        with torch.no_grad():
            tau = 1. / self.w.sigmoid()
        return super().extra_repr() + f', tau={tau}'
