"""CSPDarknet backbone (reference: yolox/models/darknet.py:97-180); same attribute names -> same checkpoint keys."""
from torch import nn

from eas_snn_amd import ops

from .network_blocks import BaseConv, CSPLayer, DWConv, Focus, SPPBottleneck


# stage name, width as a multiple of the stem's, CSPLayer depth as a multiple of the base depth, last stage (SPP block in front of a
# CSPLayer without shortcuts) -- darknet.py:118-167 of the reference spells the four stages out one by one
_STAGES = (('dark2', 2, 1, False), ('dark3', 4, 3, False), ('dark4', 8, 3, False), ('dark5', 16, 1, True))


class CSPDarknet(nn.Module):
    def __init__(self, dep_mul, wid_mul, out_features=('dark3', 'dark4', 'dark5'), depthwise=False, act='silu', in_dim=3):
        super().__init__()
        assert out_features, 'please provide output features of Darknet'
        self.out_features = out_features
        down = DWConv if depthwise else BaseConv           # the stride-2 convolution that opens every stage
        width, depth = int(wid_mul * 64), max(round(dep_mul * 3), 1)
        self.stem = Focus(in_dim, width, ksize=3, act=act)
        cin = width
        for name, wmul, dmul, last in _STAGES:
            cout = width * wmul
            layers = [down(cin, cout, 3, 2, act=act)]
            if last:
                layers.append(SPPBottleneck(cout, cout, activation=act))
            layers.append(CSPLayer(cout, cout, n=depth * dmul, shortcut=not last, depthwise=depthwise, act=act))
            setattr(self, name, nn.Sequential(*layers))     # (registration order = the reference's: same state_dict key order)
            cin = cout

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
