from .launch import launch
from .trainer import Trainer
