"""Spiking backbone + ANN neck (reference: yolox/models/spiking_yolo_pafpn.py:13-120).  The firing-rate
readout ``out_features[f].mean(axis=0)`` (:98) comes out of the fused BN+LIF kernel of the last conv of
dark3/dark4/dark5 (no extra pass over the [T,N,C,H,W] spikes)."""
from eas_snn_amd import ops
from yolox.utils.utils_snn import convert_to_spiking

from .darknet import CSPDarknet
from .network_blocks import CSPLayer, enable_spike_planes
from .yolo_pafpn import YOLOPAFPN


class SpikingYOLOPAFPN(YOLOPAFPN):
    def __init__(self, depth=1.0, width=1.0, in_features=('dark3', 'dark4', 'dark5'), in_channels=[256, 512, 1024],
                 depthwise=False, in_dim=3, act='silu', spike_fn=None):
        super(YOLOPAFPN, self).__init__()
        self.backbone = convert_to_spiking(CSPDarknet(depth, width, depthwise=depthwise, in_dim=in_dim, act=act), spike_fn)
        self.in_features = in_features
        self.in_channels = in_channels
        self._build_neck(depth, width, in_channels, depthwise, act)
        enable_spike_planes(self.backbone)          # spikes between the backbone's layers as bf16 planes (its readers are its own blocks)
        for f in in_features:                       # last block of each output stage emits its firing rate
            last = getattr(self.backbone, f)[-1]
            if isinstance(last, CSPLayer):
                last.conv3.emit_rate = True

    def _features(self, x):
        feats = self.backbone(x)
        out = []
        for f in self.in_features:
            v = feats[f]
            out.append(v[1] if isinstance(v, tuple) else ops.time_mean(v))
        return out
