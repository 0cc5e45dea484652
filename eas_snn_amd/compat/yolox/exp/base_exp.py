"""Experiment base class and the "k v k v ..." override mechanism (reference: yolox/exp/base_exp.py:16-90).

``merge`` keeps the reference's coercion rule exactly -- the override takes the TYPE of the default value --
including its quirks: ``bool("False") is True``, and ``use_spike`` is a *string* option."""
import ast
import pprint
from abc import ABCMeta, abstractmethod


class BaseExp(metaclass=ABCMeta):
    def __init__(self):
        self.seed = None
        self.output_dir = './YOLOX_outputs'
        self.print_interval = 100
        self.eval_interval = 10
        self.dataset = None

    @abstractmethod
    def get_model(self):
        pass

    def __repr__(self):
        rows = [(str(k), pprint.pformat(v)) for k, v in vars(self).items() if not k.startswith('_')]
        try:
            from tabulate import tabulate
            return tabulate(rows, headers=['keys', 'values'], tablefmt='fancy_grid')
        except ImportError:
            return '\n'.join(f'{k}: {v}' for k, v in rows)

    def merge(self, cfg_list):
        assert len(cfg_list) % 2 == 0, f'length must be even, check value here: {cfg_list}'
        for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
            if not hasattr(self, k):
                continue                                   # unknown keys are ignored, as upstream
            src = getattr(self, k)
            if isinstance(src, (list, tuple)):
                v = [t.strip() for t in v.strip('[]()').split(',')]
                if len(src) > 0:
                    v = [type(src[0])(t) for t in v]
            if src is not None and type(src) != type(v):
                try:
                    v = type(src)(v)
                except Exception:
                    v = ast.literal_eval(v)
            setattr(self, k, v)
