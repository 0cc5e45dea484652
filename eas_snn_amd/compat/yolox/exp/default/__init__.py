"""Resolves ``yolox.exp.default.<name>`` to ``<root>/exps/default/<name>.py`` (reference:
yolox/exp/default/__init__.py:11-28)."""
import sys
from importlib import abc, util
from pathlib import Path

_EXP_PATH = Path(__file__).resolve().parent.parent.parent.parent / 'exps' / 'default'

if _EXP_PATH.is_dir():
    class _ExpFinder(abc.MetaPathFinder):
        def find_spec(self, name, path, target=None):
            if not name.startswith('yolox.exp.default'):
                return None
            target_file = _EXP_PATH / (name.split('.')[-1] + '.py')
            if not target_file.is_file():
                return None
            return util.spec_from_file_location(name, target_file)

    sys.meta_path.append(_ExpFinder())
