"""Spike functions named by the experiment config (interface of the reference's yolox/models/activation.py).

``Rectangle`` is the sampler's fixed spike function (yolox/exp/event_yolox_base.py:156): Heaviside with strict ``>`` forward,
box-shaped surrogate of width 1/alpha and height alpha backward (class attribute ``alpha`` = 1, activation.py:17-30).  The HIP
sampler evaluates it inside eas_arsnn_step_fwd/bwd; this tensor-op form is what callers that import the name get.

``InvArcTanh`` / ``EfficientNoisySpikeII`` ('patan', activation.py:121-130,181-205) live in
``spikingjelly.activation_based.surrogate`` of this package as ``PATan`` (learnable alpha, evaluated by eas_lif_bwd /
eas_bn_lif_bwd with surrogate id 3)."""
import torch


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
