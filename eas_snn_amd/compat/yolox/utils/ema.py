"""Exponential moving average of the weights (reference: yolox/utils/ema.py)."""
import math
from copy import deepcopy

import torch
import torch.nn as nn


def is_parallel(model):
    return isinstance(model, (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel))


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(model.module if is_parallel(model) else model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            msd = (model.module if is_parallel(model) else model).state_dict()
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1.0 - d) * msd[k].detach()
