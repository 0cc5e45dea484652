"""SYOLOX-S: depth/width multipliers of the reference's exps/default/e_yolox_s.py."""
from yolox.exp.event_yolox_base import sized_exp

Exp = sized_exp(__file__, depth=0.33, width=0.50, max_epoch=60)
