from .yolo_head import SpikingYOLOXHead  # noqa: F401  (reference module path: yolox/models/spiking_yolo_head.py)
