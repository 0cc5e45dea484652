"""SYOLOX-L: depth/width multipliers of the reference's exps/default/e_yolox_l.py."""
from yolox.exp.event_yolox_base import sized_exp

Exp = sized_exp(__file__, depth=1.0, width=1.0, max_epoch=300)
