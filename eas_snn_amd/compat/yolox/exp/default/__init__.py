"""``yolox.exp.default.<name>`` is the experiment file ``exps/default/<name>.py`` next to the ``yolox`` package (what the
reference achieves with a meta-path finder, yolox/exp/default/__init__.py:11-28).  Here the experiment directory simply joins
this package's search path, so the ordinary import system finds the files."""
import os

_exps = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..', 'exps', 'default'))
if os.path.isdir(_exps):
    __path__.append(_exps)
