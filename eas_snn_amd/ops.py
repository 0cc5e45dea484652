"""Host-side operators over the C ABI (include/eas_hip.h): thin ``torch.autograd.Function`` wrappers that
hand raw device pointers and the current HIP stream to libeas_hip.so.  PyTorch is plumbing here (memory,
streams, autograd bookkeeping); all arithmetic of the hot path runs in the HIP kernels.

No CPU path: CPU tensors raise.

``ops`` is the one public namespace of the operator layer; the code lives in modules by kernel family:

  ops_core     the call wrapper (status check, HIP-event timing, call trace), spike planes / ghost tensors, "small integer" tags,
               the convolution-epilogue statistics slot, BatchNorm step counters
  ops_lif      K2: stand-alone multi-step neuron, mean over T
  ops_bn       conv output -> BatchNorm -> (P)LIF over T, the CSPLayer branch pair, BatchNorm + SiLU
  ops_sampler  K3: the adaptive sampler and the simpler embeddings
  ops_conv     dense convolutions on the matrix cores, weight packing and its scopes, weight-gradient side stream, fused eval blocks
  ops_events   K1 and the other event representations          ops_glue   SPP, upsample + concatenate, Focus
  ops_det      post-processing, SimOTA, detection loss         ops_group  grouped (multi-problem) launches (imported on its own)

Every switch and scope the operators consult is a field of ONE object, ``eas_snn_amd._ctx.ctx`` (``ops.ctx``).  The module-level names
of earlier rounds (``ops.WGRAD_SIDE_BATCH``, ``ops.VERIFY_SMALL_INT``, ``ops._PLANES_SCOPE`` ...) still read and ASSIGN: this module
forwards them to the context (``_ALIASES``)."""
import sys
import types

from . import ops_bn, ops_conv, ops_core, ops_lif, ops_sampler
from ._ctx import ctx

_PLUMBING = {'C', 'os', 'torch', '_lib', 'check', 'ptr', 'stream', 'opctx'}
for _m in (ops_core, ops_lif, ops_bn, ops_sampler, ops_conv):
    for _k, _v in vars(_m).items():
        if not _k.startswith('__') and _k not in _PLUMBING:
            globals()[_k] = _v
from ._lib import check, ptr, stream          # noqa: E402,F401  (callers say ops.stream())

# the other kernel families of the operator layer; ``ops.<name>`` stays the one public namespace
from .ops_events import *          # noqa: E402,F401,F403  K1 + event representations
from .ops_glue import *            # noqa: E402,F401,F403  SPP, upsample + concatenate, Focus
from .ops_det import *             # noqa: E402,F401,F403  post-processing, SimOTA, detection loss

# old module-level name -> context field
_ALIASES = {
    '_STATE_WRITEBACK': 'state_writeback', '_TIMER': 'timer', '_TAG': 'tag', '_CALL_LOG': 'call_log', 'SPIKE_PLANES': 'spike_planes',
    '_PLANES_SCOPE': 'planes_scope', '_INVSTD_SCOPE': 'invstd_scope', 'CONV_STATS': 'conv_stats', 'CONV_STATS_MAX_BLOCKS': 'conv_stats_max_blocks',
    '_WANT_CONV_STATS': 'want_conv_stats', '_CONV_STATS_SLOT': 'conv_stats_slot', 'FUSED_EVAL': 'fused_eval', '_REPLICAS': 'replicas',
    'ARSNN_FUSED': 'arsnn_fused', 'DEFER_WGRAD_REDUCE': 'defer_wgrad_reduce', '_PENDING_REDUCE': 'pending_reduce',
    'WGRAD_SIDE_BATCH': 'wgrad_side_batch', 'WGRAD_SIDE_US': 'wgrad_side_us', 'WGRAD_SIDE_AT': 'wgrad_side_at', '_SIDE': 'side',
    'VERIFY_SMALL_INT': 'verify_small_int', 'SMALL_DGRAD': 'small_dgrad', '_PACK_SCOPE': 'pack_scope', '_PACK_GEN': 'pack_gen', '_FROZEN': 'frozen',
    '_CONV_SINK': 'conv_sink', 'FUSED_ANN_EVAL': 'fused_ann_eval', '_DEFERRED': 'deferred_counters'}


class _OpsModule(types.ModuleType):
    """this module's class: the aliases above as properties, so that ``ops.X`` and ``ops.X = v`` (also ``monkeypatch.setattr(ops, 'X', v)``)
    reach the context object"""


def _alias(field):
    return property(lambda self: getattr(ctx, field), lambda self, value: setattr(ctx, field, value))


for _old, _new in _ALIASES.items():
    assert hasattr(ctx, _new), _new
    setattr(_OpsModule, _old, _alias(_new))
sys.modules[__name__].__class__ = _OpsModule
del _old, _new, _m, _k, _v
