"""The one context object of the operator layer (``eas_snn_amd.ops`` and its family modules ``ops_core`` / ``ops_lif`` / ``ops_bn`` /
``ops_sampler`` / ``ops_conv``): every switch and every scope the operators consult, in one place instead of module-level variables
spread over the files.  ``ops.<OLD_NAME>`` keeps working for reading and assignment -- ``ops`` forwards those names here
(see ``ops._ALIASES``) -- but new code says ``ctx.<field>``.

Three kinds of fields:
  switches   read from the environment once (development A/B switches; the defaults are what is measured and tested)
  scopes     set for the duration of a ``with`` block by the operators' own context managers (``packed_weights``, ``conv_stats_scope``,
             ``kernel_trace``, ``deferred_wgrad_reductions``, ...) and restored on exit
  taps       objects attached from outside (timer of bench.py, statistics sink of eas_snn_amd.stats)
One process = one context: the reference's modules are not re-entrant either (neuron state is per-module Python state; one forward at
a time per replica)."""
import os


def _env_flag(name, default='1'):
    return os.environ.get(name, default) == '1'


class Context:
    def __init__(self):
        # ---- switches
        self.spike_planes = _env_flag('EAS_SPIKE_PLANES')              # 0: fp32 spikes everywhere (development; both forms are tested bit-identical)
        self.conv_stats = _env_flag('EAS_CONV_STATS')                  # BatchNorm partial sums in the convolution epilogue
        self.conv_stats_max_blocks = int(os.environ.get('EAS_CONV_STATS_MAX_BLOCKS', '4096'))     # partials per channel the consumers still add cheaply
        self.fused_eval = {'1': 'all', 'all': 'all', '0': False, 'auto': 'auto'}.get(os.environ.get('EAS_FUSED_EVAL', 'auto'), 'auto')
        self.fused_ann_eval = _env_flag('EAS_FUSED_ANN_EVAL')
        self.arsnn_fused = _env_flag('EAS_ARSNN_FUSED')
        self.small_dgrad = os.environ.get('EAS_SMALL_DGRAD', '1') != '0'
        self.defer_wgrad_reduce = _env_flag('EAS_DEFER_WGRAD_REDUCE', '0')
        self.wgrad_side_batch = int(os.environ.get('EAS_WGRAD_SIDE', '16'))   # 0: everything on the main stream; 16: swept on config 2 (12 / 15 / 17 / 20 / 24 lose 0.1-0.3 ms of its 0.37 ms)
        self.wgrad_side_us = float(os.environ.get('EAS_WGRAD_SIDE_US', '0'))  # a batch also leaves once its estimated kernel time reaches this (0: count only)
        self.wgrad_side_at = tuple(int(v) for v in os.environ.get('EAS_WGRAD_SIDE_AT', '').split(',') if v.strip())   # or: after these launch counts of the pass
        self.verify_small_int = False       # tests switch this on: every tagged tensor is checked (host sync) before it is used
        # ---- scopes
        self.state_writeback = True         # final membrane potentials are written back after a multi-step call (ops.no_state_writeback)
        self.planes_scope = False           # inside the forward of a whole model (``packed_weights``) none of whose modules carries a forward hook
        self.invstd_scope = None            # {address of running_var: (eps, invstd)} or None
        self.want_conv_stats = False        # inside ``conv_stats_scope``: the next convolution leaves BatchNorm partial sums
        self.conv_stats_slot = None         # (y [NI,Cout,Ho,Wo], nb, stats [Cout*nb*2] fp64, y._version)
        self.replicas = 1                   # ``replicated``: N samples stand for T*N (T identical frames)
        self.pack_scope = None              # the pack dictionaries' generation that is valid right now (inside ``packed_weights``), else None
        self.pack_gen = 0
        self.frozen = None                  # {'model', 'gen', 'inv'} while a ``frozen_weights`` block is open, else None
        self.pending_reduce = []            # (slab workspace kept alive, grad_w address, numel, slab count, [(parameter, address of its .grad, numel)])
        self.side = {'stream': None, 'pending': [], 'keep': [], 'dirty': False, 'us': 0.0, 'seen': 0}     # weight-gradient side stream
        self.deferred_counters = None       # inside ``deferred_counters``: the num_batches_tracked tensors to bump at exit
        self.call_log = None                # test infrastructure (``kernel_trace``): list of (C-ABI symbol, argument tuple) of every call made through _call
        # ---- taps
        self.timer = None                   # ops.KernelTimer of bench.py / scripts
        self.tag = None                     # development: label (layer name, phase) attached to the timed calls, see KernelTimer.tagged
        self.conv_sink = None               # statistics tap (eas_snn_amd/stats.py)


ctx = Context()
