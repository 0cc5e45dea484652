"""Spike functions named by the experiment config (reference: yolox/models/activation.py).
``Rectangle`` is the sampler's fixed spike function (yolox/exp/event_yolox_base.py:156): forward ``x > 0``,
backward ``g * [|x| < 0.5/alpha] * alpha`` with class attribute ``alpha = 1``.  Inside the HIP sampler
it is evaluated by eas_arsnn_step_bwd; the autograd.Function below exists for API parity."""
import torch


class Rectangle(torch.autograd.Function):
    alpha = 1.0

    @staticmethod
    def forward(ctx, inpt):
        ctx.save_for_backward(inpt)
        return inpt.gt(0).float()

    @staticmethod
    def backward(ctx, grad_output):
        inpt, = ctx.saved_tensors
        return grad_output * ((inpt.abs() < 0.5 / Rectangle.alpha).float() * Rectangle.alpha)
