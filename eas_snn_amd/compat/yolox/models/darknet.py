"""CSPDarknet backbone (reference: yolox/models/darknet.py:97-180); same attribute names -> same checkpoint keys."""
from torch import nn

from eas_snn_amd import ops

from .network_blocks import BaseConv, CSPLayer, DWConv, Focus, SPPBottleneck


class CSPDarknet(nn.Module):
    def __init__(self, dep_mul, wid_mul, out_features=('dark3', 'dark4', 'dark5'), depthwise=False, act='silu', in_dim=3):
        super().__init__()
        assert out_features, 'please provide output features of Darknet'
        self.out_features = out_features
        Conv = DWConv if depthwise else BaseConv
        c = int(wid_mul * 64)
        d = max(round(dep_mul * 3), 1)
        self.stem = Focus(in_dim, c, ksize=3, act=act)
        self.dark2 = nn.Sequential(Conv(c, c * 2, 3, 2, act=act),
                                   CSPLayer(c * 2, c * 2, n=d, depthwise=depthwise, act=act))
        self.dark3 = nn.Sequential(Conv(c * 2, c * 4, 3, 2, act=act),
                                   CSPLayer(c * 4, c * 4, n=d * 3, depthwise=depthwise, act=act))
        self.dark4 = nn.Sequential(Conv(c * 4, c * 8, 3, 2, act=act),
                                   CSPLayer(c * 8, c * 8, n=d * 3, depthwise=depthwise, act=act))
        self.dark5 = nn.Sequential(Conv(c * 8, c * 16, 3, 2, act=act), SPPBottleneck(c * 16, c * 16, activation=act),
                                   CSPLayer(c * 16, c * 16, n=d, shortcut=False, depthwise=depthwise, act=act))

    def forward(self, x):
        feats = {}
        x = self.stem(x)
        feats['stem'] = x
        for name in ('dark2', 'dark3', 'dark4', 'dark5'):
            x = getattr(self, name)(x)
            feats[name] = x                # (spikes, firing_rate) when the stage's last conv emits its rate
            if isinstance(x, tuple):
                x = x[0]
        # what leaves the backbone is a real tensor (a stage may have handed its spikes on as planes, see ops.dense) -- unless the module
        # that owns this backbone reads planes itself and says so (YOLOPAFPN: its neck is convolutions and ops.upsample_cat)
        keep = getattr(self, 'planes_to_owner', False)
        return {k: (v if (isinstance(v, tuple) or keep) else ops.dense(v)) for k, v in feats.items() if k in self.out_features}
