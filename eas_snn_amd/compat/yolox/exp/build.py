"""Experiment lookup (interface of the reference's yolox/exp/build.py:11-42): ``get_exp(exp_file, exp_name)`` returns an
instance of the ``Exp`` class of an experiment file, or of the packaged experiment ``yolox.exp.default.<name>`` ("e-yolox-s"
names the module ``e_yolox_s``).  A file wins over a name when both are given; a file that cannot be imported or has no
``Exp`` raises ImportError, like the reference."""
import importlib
import importlib.util
import os
import sys

_PACKAGED = 'yolox.exp.default'


def get_exp_by_file(exp_file):
    folder, leaf = os.path.split(exp_file)
    stem = leaf.split('.')[0]
    if folder not in sys.path:
        sys.path.append(folder)          # experiment files import their siblings by bare name
    try:
        return importlib.import_module(stem).Exp()
    except Exception as err:
        raise ImportError("{} doesn't contains class named 'Exp'".format(exp_file)) from err


def get_exp_by_name(exp_name):
    return importlib.import_module(_PACKAGED + '.' + exp_name.replace('-', '_')).Exp()


def get_exp(exp_file=None, exp_name=None):
    if exp_file is None and exp_name is None:
        raise AssertionError('plz provide exp file or exp name.')
    return get_exp_by_name(exp_name) if exp_file is None else get_exp_by_file(exp_file)
