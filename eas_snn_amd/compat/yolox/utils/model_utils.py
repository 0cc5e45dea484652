"""Model helpers the tools import (reference: yolox/utils/model_utils.py)."""
import torch.nn as nn


def get_model_info(model: nn.Module, tsize=None) -> str:
    """Parameter count (the reference also runs thop on a dummy input; thop is optional and not used here)."""
    params = sum(p.numel() for p in model.parameters()) / 1e6
    return 'Params: {:.2f}M'.format(params)


def fuse_conv_and_bn(conv, bn):
    """conv followed by eval-mode BatchNorm as one biased convolution: every output channel is scaled by
    gamma / sqrt(var + eps) and shifted (reference: yolox/utils/model_utils.py fuse_conv_and_bn)."""
    import torch
    with torch.no_grad():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        bias = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, groups=conv.groups,
                        bias=True).to(conv.weight.device)
        out.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
        out.bias.copy_((bias - bn.running_mean) * scale + bn.bias)
    return out.requires_grad_(False)


def fuse_model(model: nn.Module) -> nn.Module:
    """Fold BN into the conv of every ANN BaseConv (spiking blocks keep their BN: it is fused with the neuron)."""
    from yolox.models.network_blocks import BaseConv
    for m in model.modules():
        if type(m) is BaseConv and hasattr(m, 'bn') and isinstance(m.conv, nn.Conv2d) and type(m.bn) is nn.BatchNorm2d:
            m.conv = fuse_conv_and_bn(m.conv, m.bn)
            delattr(m, 'bn')
            m.forward = m.fuseforward
    # the module tree changed: the cached list of convolutions whose weights one launch packs per forward (ops.prepack_conv_weights) is stale
    for holder in model.modules():
        for attr in ('_eas_pack_plan', '_eas_pack_plan_fwd', '_eas_modules'):
            holder.__dict__.pop(attr, None)
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
