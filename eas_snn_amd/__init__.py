"""eas_snn_amd -- MI355X (gfx950) native hot path of EAS-SNN.

Layout:
  csrc/            HIP kernels + the C ABI (include/eas_hip.h)  -> libeas_hip.so
  _lib.py, ops.py  ctypes binding and autograd operators over the C ABI
  compat/          host-side mirror of the reference's operator interface:
                   ``spikingjelly.activation_based`` and the ``yolox`` names the EAS-SNN tools import
  data.py          synthetic event streams -> GPU event histogram (K1) -> model input; augmentation draws + box transform
  stats.py         spike-count / SOP statistics and the reference's energy estimate on the device
  parallel.py      flat gradient all-reduce over RCCL for one process per GPU

Importing this package puts ``compat/`` at the front of ``sys.path`` so that ``import spikingjelly`` /
``import yolox`` resolve to the HIP-backed implementations (set EAS_SNN_NO_COMPAT=1 to skip).
"""
import os
import sys

__version__ = '0.1.0'

COMPAT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'compat')


def install():
    for name in ('spikingjelly', 'yolox'):
        mod = sys.modules.get(name)
        if mod is not None and not os.path.abspath(getattr(mod, '__file__', '') or '').startswith(COMPAT_DIR):
            raise ImportError(f"another '{name}' package is already imported ({getattr(mod, '__file__', None)}); "
                              'import eas_snn_amd first or put eas_snn_amd/compat on PYTHONPATH')
    if COMPAT_DIR not in sys.path:
        sys.path.insert(0, COMPAT_DIR)


if os.environ.get('EAS_SNN_NO_COMPAT', '0') != '1':
    install()

from . import _lib  # noqa: E402
from ._lib import EasHipError, build  # noqa: E402,F401


def hip_library():
    """Loaded libeas_hip.so (raises EasHipError when it has not been built)."""
    return _lib.lib()
