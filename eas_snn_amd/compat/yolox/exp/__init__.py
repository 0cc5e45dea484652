from .base_exp import BaseExp
from .build import get_exp
from .event_yolox_base import EventExp, check_exp_value

Exp = EventExp   # upstream exports the COCO ``Exp`` here (yolox/exp/yolox_base.py, out of scope); the tools only import the name
