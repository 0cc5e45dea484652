"""Surrogate-gradient spike functions (spikingjelly ``surrogate.ATan`` / ``surrogate.Sigmoid``).

Inside a neuron the surrogate is NOT called: the fused HIP kernels read ``hip_id`` and ``alpha`` from the
module and evaluate the gradient in the backward kernel (eas_common.h: eas_surrogate_grad).  Calling the
module directly (API completeness, not on the hot path) evaluates the same formulas with tensor ops.
"""
import math

import torch
import torch.nn as nn


def heaviside(x):
    return (x >= 0).to(x)


class _ATanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return heaviside(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return ctx.alpha / 2 / (1 + (math.pi / 2 * ctx.alpha * x).pow(2)) * g, None


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return heaviside(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        s = (x * ctx.alpha).sigmoid()
        return g * (1. - s) * s * ctx.alpha, None


class SurrogateFunctionBase(nn.Module):
    hip_id = None            # id understood by libeas_hip.so (include/eas_hip.h: EAS_SG_*)

    def __init__(self, alpha, spiking=True):
        super().__init__()
        self.spiking = spiking
        self.alpha = alpha

    def set_spiking_mode(self, spiking):
        self.spiking = spiking

    def extra_repr(self):
        return f'alpha={self.alpha}, spiking={self.spiking}'

    def forward(self, x):
        if self.spiking:
            return self.spiking_function(x, self.alpha)
        return self.primitive_function(x, self.alpha)


class ATan(SurrogateFunctionBase):
    """g'(x) = alpha / 2 / (1 + (pi/2 * alpha * x)^2)"""
    hip_id = 'atan'

    def __init__(self, alpha=2.0, spiking=True):
        super().__init__(alpha, spiking)

    @staticmethod
    def spiking_function(x, alpha):
        return _ATanFn.apply(x, alpha)

    @staticmethod
    def primitive_function(x, alpha):
        return (math.pi / 2 * alpha * x).atan_() / math.pi + 0.5


class Sigmoid(SurrogateFunctionBase):
    """g'(x) = alpha * sigmoid(alpha x) * (1 - sigmoid(alpha x))"""
    hip_id = 'sigmoid'

    def __init__(self, alpha=4.0, spiking=True):
        super().__init__(alpha, spiking)

    @staticmethod
    def spiking_function(x, alpha):
        return _SigmoidFn.apply(x, alpha)

    @staticmethod
    def primitive_function(x, alpha):
        return (x * alpha).sigmoid()


class Rect(SurrogateFunctionBase):
    """g'(x) = alpha * [|x| < 0.5/alpha]  (the reference's in-repo ``Rectangle``, yolox/models/activation.py:17-30,
    with spikingjelly's ``>=`` firing rule)."""
    hip_id = 'rect'

    def __init__(self, alpha=1.0, spiking=True):
        super().__init__(alpha, spiking)

    @staticmethod
    def spiking_function(x, alpha):
        soft = torch.clamp(alpha * x + 0.5, 0, 1)
        return soft + (heaviside(x) - soft).detach()

    primitive_function = spiking_function
