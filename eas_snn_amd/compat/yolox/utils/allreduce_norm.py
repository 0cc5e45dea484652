"""All-reduce of BatchNorm statistics before evaluation (reference: yolox/utils/allreduce_norm.py:59-103):
ranks train with per-rank running stats (DDP ``broadcast_buffers=False``) and average them here."""
import torch
from torch import distributed as dist
from torch import nn

from .dist import get_world_size

NORMS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)


def get_async_norm_states(module):
    states = {}
    for name, child in module.named_modules():
        if isinstance(child, NORMS):
            for k, v in child.state_dict().items():
                states['.'.join([name, k])] = v
    return states


def all_reduce_norm(module):
    """Average every norm layer's weight/bias/running stats over ranks with ONE flattened RCCL all-reduce."""
    world = get_world_size()
    if world == 1:
        return
    states = get_async_norm_states(module)
    if not states:
        return
    keys = sorted(states)                                  # same module tree on every rank -> same order
    flat = torch.cat([states[k].detach().reshape(-1).float() for k in keys])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    out = {}
    for k in keys:
        n = states[k].numel()
        out[k] = flat[off:off + n].view(states[k].shape).to(states[k].dtype)
        off += n
    module.load_state_dict(out, strict=False)
