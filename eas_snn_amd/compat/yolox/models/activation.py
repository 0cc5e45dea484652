"""Spike functions named by the experiment config (interface of the reference's yolox/models/activation.py).

``Rectangle`` is the sampler's fixed spike function (yolox/exp/event_yolox_base.py:156): Heaviside with strict ``>`` forward,
box-shaped surrogate of width 1/alpha and height alpha backward (class attribute ``alpha`` = 1, activation.py:17-30).  The HIP
sampler evaluates it inside eas_arsnn_step_fwd/bwd; this tensor-op form is what callers that import the name get.

``EfficientNoisySpikeII(InvArcTanh(alpha), p=0)`` is the 'patan' spike function (event_yolox_base.py:145-150,
activation.py:121-130,181-205): Heaviside forward, gradient of sigma(u) = atan(pi/2 |alpha| u)/pi + 1/2 backward, with a
LEARNABLE alpha (parameter ``...act.surrogate_function.inv_sg.alpha`` of shape [1], same state_dict key as the reference).
Inside a neuron the HIP kernels evaluate it (surrogate id EAS_SG_PATAN: eas_lif_bwd_patan / eas_bn_lif_bwd_patan read alpha from
the device and reduce dL/dalpha in a fixed order); called directly, the modules compute the same with tensor ops."""
import math

import torch
import torch.nn as nn


class Rectangle(torch.autograd.Function):
    alpha = 1.0

    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return (u > 0).to(u.dtype)

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        a = Rectangle.alpha
        return torch.where(u.abs() < 0.5 / a, g * a, torch.zeros_like(g))


class InvArcTanh(nn.Module):
    def __init__(self, alpha: float = 1.0, learnable=True):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor([float(alpha)])) if learnable else alpha

    def get_temperature(self):
        return self.alpha.detach().clone()

    def forward(self, x, gates=None):
        return torch.atan(math.pi / 2.0 * torch.abs(self.alpha) * x) / math.pi + 0.5


class EfficientNoisySpikeII(nn.Module):
    """Noise probability ``p`` other than 0 (random per-element mixing of the spike and its surrogate in the FORWARD value) is not
    a configuration the experiment files reach ('patan' fixes p = 0) and has no HIP kernel: the constructor refuses it."""
    hip_id = 'patan'

    def __init__(self, inv_sg=None, p=0.5, spike=True):
        super().__init__()
        if p != 0 or not spike:
            raise NotImplementedError('EfficientNoisySpikeII: only p = 0, spike = True (the "patan" spike function) is provided')
        self.inv_sg = InvArcTanh() if inv_sg is None else inv_sg
        if not isinstance(self.inv_sg, InvArcTanh):
            raise NotImplementedError(f'EfficientNoisySpikeII({type(self.inv_sg).__name__}) has no HIP backward (InvArcTanh only)')
        self.p, self.spike = p, spike
        self.mask = None

    @property
    def alpha(self):
        """what the neuron hands to the kernels: the learnable device scalar itself (never read on the host)"""
        return self.inv_sg.alpha

    def reset_mask(self):
        self.mask = None

    def forward(self, x, gates=None):
        step = (x >= 0).to(x.dtype)
        if not self.training:
            return step
        soft = self.inv_sg(x, gates)
        return soft + (step - soft).detach()          # p = 0: the mask is all ones
