from .yolox import SpikingYOLOX  # noqa: F401  (reference module path: yolox/models/spiking_yolox.py)
