"""Learning-rate schedules (reference: yolox/utils/lr_scheduler.py); only the two the event recipes use."""
import math
from functools import partial


def _yolox_warm_cos(lr, min_lr_ratio, total_iters, warmup_total_iters, warmup_lr_start, no_aug_iter, iters):
    min_lr = lr * min_lr_ratio
    if iters <= warmup_total_iters:
        return (lr - warmup_lr_start) * pow(iters / float(max(warmup_total_iters, 1)), 2) + warmup_lr_start
    if iters >= total_iters - no_aug_iter:
        return min_lr
    return min_lr + 0.5 * (lr - min_lr) * (1.0 + math.cos(
        math.pi * (iters - warmup_total_iters) / (total_iters - warmup_total_iters - no_aug_iter)))


def _cos(lr, total_iters, iters):
    return lr * 0.5 * (1.0 + math.cos(math.pi * iters / total_iters))


class LRScheduler:
    def __init__(self, name, lr, iters_per_epoch, total_epochs, **kwargs):
        self.lr, self.iters_per_epoch, self.total_epochs = lr, iters_per_epoch, total_epochs
        self.total_iters = iters_per_epoch * total_epochs
        self.__dict__.update(kwargs)
        if name == 'cos':
            self.lr_func = partial(_cos, lr, self.total_iters)
        elif name == 'yoloxwarmcos':
            self.lr_func = partial(_yolox_warm_cos, lr, getattr(self, 'min_lr_ratio', 0.2), self.total_iters,
                                   iters_per_epoch * self.warmup_epochs, getattr(self, 'warmup_lr_start', 0),
                                   iters_per_epoch * self.no_aug_epochs)
        else:
            raise ValueError('Scheduler version {} not supported.'.format(name))

    def update_lr(self, iters):
        return self.lr_func(iters)
