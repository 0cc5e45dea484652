"""SYOLOX-M: depth/width multipliers of the reference's exps/default/e_yolox_m.py."""
from yolox.exp.event_yolox_base import sized_exp

Exp = sized_exp(__file__, depth=0.67, width=0.75, max_epoch=300)
