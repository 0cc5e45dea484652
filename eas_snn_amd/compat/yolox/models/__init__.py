from .darknet import CSPDarknet
from .embedding import AdaptiveRSNNEmbedding, SpikeCountEmbedding
from .losses import IOUloss
from .yolo_head import SpikingYOLOXHead, YOLOXHead
from .yolo_pafpn import YOLOPAFPN
from .spiking_yolo_pafpn import SpikingYOLOPAFPN
from .yolox import YOLOX, SpikingYOLOX
