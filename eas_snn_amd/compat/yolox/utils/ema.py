"""Exponential moving average of the weights (interface of the reference's yolox/utils/ema.py: ``ModelEMA(model, decay,
updates)``, ``.ema``, ``.updates``, ``.decay(n)``, ``.update(model)``; ``is_parallel``).

The average covers every floating-point entry of the state dict (parameters and BatchNorm running statistics); integer
entries (``num_batches_tracked``) keep the value they were copied with.  The update rounds like the reference's three tensor
operators per entry (``v *= d``; ``(1 - d) * m``; ``v += ...``, ema.py:57-60) but as multi-tensor launches for the whole model;
on the GPU ``eas_snn_amd.optim.FusedAdam.attach_ema`` folds it into the optimizer's one launch (csrc/adam.hip), after which
``update`` only counts: the decay ramp then lives in a device counter and the update survives HIP-graph replay."""
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
        self.ramp = _RAMP
        self.updates = updates
        self._fused_in = None           # the optimizer whose step makes this update (FusedAdam.attach_ema)
        self.ema = copy.deepcopy(_bare(model))
        self.ema.eval()
        self.ema.requires_grad_(False)

    def decay(self, n):
        """effective decay after n updates: nominal * (1 - exp(-n / 2000)) -- a short warm-up so early averages follow the model"""
        return self.nominal_decay * (1.0 - math.exp(-n / _RAMP))

    @torch.no_grad()
    def update(self, model):
        self.updates += 1
        if self._fused_in is not None:
            return                      # made by the optimizer step that has just run (one launch for Adam + average)
        keep = self.decay(self.updates)
        live = _bare(model).state_dict()
        avg, cur = [], []
        for name, tensor in self.ema.state_dict().items():
            if tensor.is_floating_point():
                avg.append(tensor)
                cur.append(live[name].detach().to(tensor.dtype))
        if avg:
            torch._foreach_mul_(avg, keep)
            torch._foreach_add_(avg, torch._foreach_mul(cur, 1.0 - keep))
