import math

import torch


def warp_decay(decay):
    """logit, so that sigmoid(warp_decay(d)) == d (reference: yolox/utils/util.py:278-280)."""
    return torch.tensor(math.log(decay / (1 - decay)))
