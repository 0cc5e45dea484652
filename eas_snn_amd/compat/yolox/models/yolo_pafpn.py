"""PAFPN neck (reference: yolox/models/yolo_pafpn.py:12-116, spiking_yolo_pafpn.py:13-120)."""
import torch
import torch.nn as nn

from eas_snn_amd import ops

from .darknet import CSPDarknet
from .network_blocks import BaseConv, CSPLayer, DWConv


def _cat2(a, b):
    """torch.cat([a, b], channel axis) with a one-kernel backward (two contiguous gradients instead of two slice copies)"""
    if ops.upcat_supported(a, b, 1):
        return ops.upsample_cat(a, b, 1)
    out = torch.cat([ops.dense(a), ops.dense(b)], -3)
    return ops.mark_small_int(out) if ops.is_small_int(a) and ops.is_small_int(b) else out


class YOLOPAFPN(nn.Module):
    def __init__(self, depth=1.0, width=1.0, in_features=('dark3', 'dark4', 'dark5'), in_channels=[256, 512, 1024],
                 depthwise=False, in_dim=3, act='silu'):
        super().__init__()
        self.backbone = CSPDarknet(depth, width, depthwise=depthwise, in_dim=in_dim, act=act)
        self.backbone.planes_to_owner = True      # converted (full_spike) model: the neck below takes the stages' spikes as planes
        self.in_features = in_features
        self.in_channels = in_channels
        self._build_neck(depth, width, in_channels, depthwise, act)

    def _build_neck(self, depth, width, in_channels, depthwise, act):
        Conv = DWConv if depthwise else BaseConv
        c0, c1, c2 = (int(c * width) for c in in_channels)
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode='nearest')
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = Conv(c0, c0, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = Conv(c1, c1, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise, act=act)

    def _features(self, x):
        feats = self.backbone(x)
        return [feats[f] for f in self.in_features]

    def _up_cat(self, low, skip):
        """torch.cat([self.upsample(low), skip], channel axis): one kernel (and one for its backward) for a plain nearest x2"""
        up = self.upsample[0] if isinstance(self.upsample, nn.Sequential) and len(self.upsample) == 1 else self.upsample
        if (type(up) is nn.Upsample and up.mode == 'nearest' and up.scale_factor in (2, 2.0, (2, 2), (2.0, 2.0)) and up.size is None
                and ops.upcat_supported(low, skip, 2)):
            return ops.upsample_cat(low, skip, 2)
        return torch.cat([self.upsample(low), ops.dense(skip)], -3)

    def forward(self, x):
        x2, x1, x0 = self._features(x)
        fpn_out0 = self.lateral_conv0(x0)
        f_out0 = self.C3_p4(self._up_cat(fpn_out0, x1))
        fpn_out1 = self.reduce_conv1(f_out0)
        pan_out2 = self.C3_p3(self._up_cat(fpn_out1, x2))
        pan_out1 = self.C3_n3(_cat2(self.bu_conv2(pan_out2), fpn_out1))
        pan_out0 = self.C3_n4(_cat2(self.bu_conv1(pan_out1), fpn_out0))
        return pan_out2, pan_out1, pan_out0
