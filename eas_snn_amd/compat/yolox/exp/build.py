"""Experiment lookup by file or by name (reference: yolox/exp/build.py:11-42)."""
import importlib
import os
import sys


def get_exp_by_file(exp_file):
    try:
        sys.path.append(os.path.dirname(exp_file))
        module = importlib.import_module(os.path.basename(exp_file).split('.')[0])
        return module.Exp()
    except Exception:
        raise ImportError("{} doesn't contains class named 'Exp'".format(exp_file))


def get_exp_by_name(exp_name):
    module_name = '.'.join(['yolox', 'exp', 'default', exp_name.replace('-', '_')])   # "e-yolox-s" -> e_yolox_s
    return importlib.import_module(module_name).Exp()


def get_exp(exp_file=None, exp_name=None):
    assert exp_file is not None or exp_name is not None, 'plz provide exp file or exp name.'
    return get_exp_by_file(exp_file) if exp_file is not None else get_exp_by_name(exp_name)
