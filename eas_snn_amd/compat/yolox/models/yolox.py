"""Model wrappers: sampler -> (T-broadcast) -> backbone -> head (reference: yolox/models/yolox.py:10-60,
spiking_yolox.py:16-76)."""
import contextlib

import torch
import torch.nn as nn

from eas_snn_amd import ops

from .yolo_head import SpikingYOLOXHead, YOLOXHead
from .yolo_pafpn import YOLOPAFPN
from .spiking_yolo_pafpn import SpikingYOLOPAFPN

_LOSS_KEYS = ('total_loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')


def _run_embedding(embedding, x, squeeze_above):
    if isinstance(embedding, nn.ModuleList):
        x = embedding[0](x)
        if x.dim() > 4:
            x = x[0]
        if len(embedding) > 1:
            x = embedding[1](x)
        return x
    x = embedding(x)
    if x.dim() > squeeze_above:
        x = x[0]
    return x


def _head_outputs(model, fpn_outs, targets, x):
    if not model.training:
        return model.head(fpn_outs)
    assert targets is not None
    loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg = model.head(fpn_outs, targets, x)
    return {'total_loss': loss, 'iou_loss': iou_loss, 'l1_loss': l1_loss, 'conf_loss': conf_loss, 'cls_loss': cls_loss,
            'num_fg': num_fg}


def _scope(model, x):
    """all convolution weights packed for the matrix-core kernels by one launch, valid for this forward only"""
    return ops.packed_weights(model) if x.is_cuda else contextlib.nullcontext()


class YOLOX(nn.Module):
    def __init__(self, backbone=None, head=None, embedding=None):
        super().__init__()
        self.embedding = embedding
        self.backbone = YOLOPAFPN() if backbone is None else backbone
        self.head = YOLOXHead(80) if head is None else head

    def forward(self, x, targets=None):
        with _scope(self, x), ops.deferred_counters():
            x = _run_embedding(self.embedding, x, 4)
            return _head_outputs(self, self.backbone(x), targets, x)


class SpikingYOLOX(nn.Module):
    def __init__(self, backbone=None, head=None, embedding=None, T=4):
        super().__init__()
        self.nb_steps = T
        self.embedding = embedding
        self.backbone = SpikingYOLOPAFPN() if backbone is None else backbone
        self.head = SpikingYOLOXHead(80) if head is None else head

    def forward(self, x, targets=None):
        with _scope(self, x), ops.deferred_counters():
            return self._forward(x, targets)

    def _forward(self, x, targets=None):
        x = _run_embedding(self.embedding, x, 5)
        # one adaptive frame is repeated for all T steps; Ts > 1 must equal T (spiking_yolox.py:52-57)
        if x.dim() == 4 or x.shape[0] == 1:
            base = x if x.dim() == 4 else x[0]
            x = base.unsqueeze(0).expand(self.nb_steps, *base.shape)
            x._eas_base = base          # lets stateless layers in front of the first neuron run once instead of T times
        else:
            assert x.shape[0] == self.nb_steps, 'the timestep of SNN is not matched with that of input'
        return _head_outputs(self, self.backbone(x), targets, x)
