"""Model helpers the tools import (reference: yolox/utils/model_utils.py)."""
import torch.nn as nn


def get_model_info(model: nn.Module, tsize=None) -> str:
    """Parameter count (the reference also runs thop on a dummy input; thop is optional and not used here)."""
    params = sum(p.numel() for p in model.parameters()) / 1e6
    return 'Params: {:.2f}M'.format(params)


def fuse_conv_and_bn(conv, bn):
    import torch
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))
    fused.weight.copy_(torch.mm(w_bn, conv.weight.clone().view(conv.out_channels, -1)).view(fused.weight.shape))
    b_conv = torch.zeros(conv.weight.size(0), device=conv.weight.device) if conv.bias is None else conv.bias
    b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
    fused.bias.copy_(torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
    return fused


def fuse_model(model: nn.Module) -> nn.Module:
    """Fold BN into the conv of every ANN BaseConv (spiking blocks keep their BN: it is fused with the neuron)."""
    from yolox.models.network_blocks import BaseConv
    for m in model.modules():
        if type(m) is BaseConv and hasattr(m, 'bn') and isinstance(m.conv, nn.Conv2d) and type(m.bn) is nn.BatchNorm2d:
            m.conv = fuse_conv_and_bn(m.conv, m.bn)
            delattr(m, 'bn')
            m.forward = m.fuseforward
    return model


def adjust_status(module: nn.Module, training: bool = False):
    """context manager restoring each submodule's train/eval flag (used around evaluation)."""
    from contextlib import contextmanager

    @contextmanager
    def _ctx():
        status = {m: m.training for m in module.modules()}
        module.train(training)
        try:
            yield module
        finally:
            for m, t in status.items():
                m.training = t
    return _ctx()
