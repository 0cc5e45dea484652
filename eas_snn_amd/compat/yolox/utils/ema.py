"""Exponential moving average of the weights (interface of the reference's yolox/utils/ema.py: ``ModelEMA(model, decay,
updates)``, ``.ema``, ``.updates``, ``.decay(n)``, ``.update(model)``; ``is_parallel``).

The average covers every floating-point entry of the state dict (parameters and BatchNorm running statistics); integer
entries (``num_batches_tracked``) keep the value they were copied with.  The update is two multi-tensor launches for the
whole model instead of two launches per tensor."""
import copy
import math

import torch
from torch.nn.parallel import DataParallel, DistributedDataParallel

_WRAPPERS = (DataParallel, DistributedDataParallel)
_RAMP = 2000.0          # updates over which the effective decay ramps from 0 to its nominal value


def is_parallel(model):
    return isinstance(model, _WRAPPERS)


def _bare(model):
    return model.module if isinstance(model, _WRAPPERS) else model


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        self.nominal_decay = float(decay)
        self.updates = updates
        self.ema = copy.deepcopy(_bare(model))
        self.ema.eval()
        self.ema.requires_grad_(False)

    def decay(self, n):
        """effective decay after n updates: nominal * (1 - exp(-n / 2000)) -- a short warm-up so early averages follow the model"""
        return self.nominal_decay * (1.0 - math.exp(-n / _RAMP))

    @torch.no_grad()
    def update(self, model):
        self.updates += 1
        keep = self.decay(self.updates)
        live = _bare(model).state_dict()
        avg, cur = [], []
        for name, tensor in self.ema.state_dict().items():
            if tensor.is_floating_point():
                avg.append(tensor)
                cur.append(live[name].detach().to(tensor.dtype))
        if avg:
            torch._foreach_mul_(avg, keep)
            torch._foreach_add_(avg, cur, alpha=1.0 - keep)
