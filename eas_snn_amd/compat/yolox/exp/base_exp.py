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
        """``[key, text, key, text, ...]``: every key the experiment already has takes ``text`` coerced to the TYPE of its current value
        (base_exp.py:67-90 of the reference); keys it does not have are skipped silently"""
        assert len(cfg_list) % 2 == 0, f'length must be even, check value here: {cfg_list}'
        for i in range(0, len(cfg_list), 2):
            key, text = cfg_list[i], cfg_list[i + 1]
            if hasattr(self, key):
                setattr(self, key, _coerce(getattr(self, key), text))


def _coerce(default, text):
    """the override ``text`` in the type of ``default``.  Sequences: brackets stripped, split at commas, items in the type of the default's
    first item (an empty default keeps strings), then the default's own sequence type.  Everything else: ``type(default)(text)`` -- which
    makes ``bool('False')`` True and keeps ``use_spike`` a string, quirks the reference's recipes rely on -- with ``ast.literal_eval`` as the
    fallback when that constructor refuses; a ``None`` default keeps the text."""
    value = text
    if isinstance(default, (list, tuple)):
        items = [part.strip() for part in text.strip('[]()').split(',')]
        value = [type(default[0])(item) for item in items] if len(default) else items
    if default is None or type(default) is type(value):
        return value
    try:
        return type(default)(value)
    except Exception:
        return ast.literal_eval(value)
