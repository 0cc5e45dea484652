"""ctypes binding of libeas_hip.so (the C ABI declared in include/eas_hip.h).

There is NO CPU fallback: if the library is missing every op raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C eas_snn_amd/csrc``.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EAS_LIB') or os.path.join(_HERE, 'libeas_hip.so')      # EAS_LIB: development (a variant build, scripts/build_variant.sh)
CSRC = os.path.join(_HERE, 'csrc')

_lib = None

_f32p, _i32p, _u32p, _u16p, _u8p, _i64p, _f64p = (C.c_void_p,) * 7   # raw device addresses
_P = C.c_void_p

class EasWgradReduceJob(C.Structure):
    """include/eas_hip.h: EasWgradReduceJob"""
    _fields_ = [('slabs', C.c_void_p), ('grad_w', C.c_void_p), ('n', C.c_int), ('slabs_count', C.c_int)]


class EasSmallconvPackJob(C.Structure):
    """include/eas_hip.h: EasSmallconvPackJob"""
    _fields_ = [('w', C.c_void_p), ('wr', C.c_void_p), ('Cin', C.c_int), ('Cout', C.c_int), ('k', C.c_int), ('mode', C.c_int),
                ('o_total', C.c_int), ('o_off', C.c_int)]


class EasBnPending(C.Structure):
    """include/eas_hip.h EasBnPending: statistics whose finalize happens inside the consuming kernel."""
    _fields_ = [('partial', C.c_void_p), ('chunks', C.c_int), ('replicas', C.c_int), ('count', C.c_double), ('eps', C.c_float),
                ('momentum', C.c_float), ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('pitch', C.c_int)]


class EasLifRange(C.Structure):
    """include/eas_hip.h EasLifRange: one neuron layer (a range of output channels) of the fused eval step"""
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p), ('w_logit', C.c_void_p),
                ('k_const', C.c_float), ('v_th', C.c_float), ('planes', C.c_void_p), ('out_f32', C.c_void_p), ('out_ctot', C.c_int),
                ('out_c0', C.c_int), ('res_planes', C.c_void_p), ('res_f32', C.c_void_p), ('res_ctot', C.c_int), ('rate', C.c_void_p),
                ('v_in', C.c_void_p), ('v_out', C.c_void_p)]


class EasConvBnLifEval(C.Structure):
    """include/eas_hip.h EasConvBnLifEval"""
    _fields_ = [('x', C.c_void_p), ('packed_w', C.c_void_p), ('x_terms', C.c_int), ('x_shared', C.c_int), ('T', C.c_int), ('N', C.c_int),
                ('Cin', C.c_int), ('Cout', C.c_int), ('Hi', C.c_int), ('Wi', C.c_int), ('ksize', C.c_int), ('stride', C.c_int),
                ('csplit', C.c_int), ('range', EasLifRange * 2)]


class EasBnActRange(C.Structure):
    """include/eas_hip.h EasBnActRange: one BatchNorm (a range of output channels) of the fused real-valued eval block"""
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p), ('out', C.c_void_p),
                ('out_ctot', C.c_int), ('out_c0', C.c_int)]


class EasConvBnActEval(C.Structure):
    """include/eas_hip.h EasConvBnActEval"""
    _fields_ = [('x', C.c_void_p), ('packed_w', C.c_void_p), ('x_terms', C.c_int), ('NI', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int),
                ('Hi', C.c_int), ('Wi', C.c_int), ('ksize', C.c_int), ('stride', C.c_int), ('act', C.c_int), ('csplit', C.c_int),
                ('range', EasBnActRange * 2)]


class EasConvProblem(C.Structure):
    """include/eas_hip.h EasConvProblem: one convolution of a grouped launch"""
    _fields_ = [('x', C.c_void_p), ('packed_w', C.c_void_p), ('bias', C.c_void_p), ('y', C.c_void_p), ('stats', C.c_void_p),
                ('NI', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('Hi', C.c_int), ('Wi', C.c_int), ('accumulate', C.c_int)]


class EasBnSiluFwdProblem(C.Structure):
    """include/eas_hip.h EasBnSiluFwdProblem"""
    _fields_ = [('y', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('out', C.c_void_p), ('N', C.c_int), ('C', C.c_int), ('HW', C.c_int), ('out_ctot', C.c_int), ('y_ctot', C.c_int),
                ('pending', EasBnPending)]


class EasBnSiluBwdProblem(C.Structure):
    """include/eas_hip.h EasBnSiluBwdProblem"""
    _fields_ = [('grad_out', C.c_void_p), ('y', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p), ('gamma', C.c_void_p),
                ('beta', C.c_void_p), ('grad_y', C.c_void_p), ('grad_gamma', C.c_void_p), ('grad_beta', C.c_void_p),
                ('workspace', C.c_void_p), ('batch_stats', C.c_int), ('N', C.c_int), ('C', C.c_int), ('HW', C.c_int),
                ('grad_out_ctot', C.c_int), ('y_ctot', C.c_int)]


class EasWgradProblem(C.Structure):
    """include/eas_hip.h EasWgradProblem"""
    _fields_ = [('x', C.c_void_p), ('grad_y', C.c_void_p), ('workspace', C.c_void_p), ('NI', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int),
                ('Hi', C.c_int), ('Wi', C.c_int)]


class EasChannelSumProblem(C.Structure):
    """include/eas_hip.h EasChannelSumProblem"""
    _fields_ = [('g', C.c_void_p), ('out', C.c_void_p), ('N', C.c_int), ('C', C.c_int), ('HW', C.c_int)]


class EasPredDgradProblem(C.Structure):
    """include/eas_hip.h EasPredDgradProblem"""
    _fields_ = [('gy_a', C.c_void_p), ('w_a', C.c_void_p), ('Ka', C.c_int), ('gy_b', C.c_void_p), ('w_b', C.c_void_p), ('Kb', C.c_int),
                ('gx', C.c_void_p), ('N', C.c_int), ('C', C.c_int), ('HW', C.c_int)]


class EasAdamHyper(C.Structure):
    """include/eas_hip.h EasAdamHyper"""
    _fields_ = [('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double), ('group_lr', C.c_double * 16), ('ema_updates', C.c_void_p),
                ('ema_decay', C.c_double), ('ema_ramp', C.c_double)]


# name -> (restype, argtypes) ; one line per prototype of include/eas_hip.h
ABI_VERSION = 9

PROTOTYPES = {
    'eas_abi_version': (C.c_int, []),
    'eas_kernel_trace_begin': (None, []),
    'eas_kernel_trace_dump': (C.c_int64, [C.c_char_p, C.c_int64]),
    'eas_launch_counter': (C.c_int64, []),
    'eas_status_string': (C.c_char_p, [C.c_int]),
    'eas_conv_bn_lif_eval': (C.c_int, [C.POINTER(EasConvBnLifEval), _P]),
    'eas_conv_bn_lif_eval_supported': (C.c_int, [C.c_int] * 10),
    'eas_conv_bn_act_eval': (C.c_int, [C.POINTER(EasConvBnActEval), _P, _P]),
    'eas_conv_bn_act_eval_group': (C.c_int, [C.POINTER(EasConvBnActEval), C.c_int, _P]),
    'eas_event_histogram': (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    'eas_event_histogram_dat': (C.c_int, [_P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    'eas_counts_to_canvas': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'eas_event_voxel_cube': (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'eas_event_time_surface': (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P]),
    'eas_spike_sop_workspace_doubles': (C.c_int64, []),
    'eas_spike_sop': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    'eas_counts_letterbox': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'eas_postprocess_workspace_bytes': (C.c_int64, [C.c_int, C.c_int]),
    'eas_postprocess': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, _P, _P]),
    'eas_event_frames': (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    'eas_simota_assign': (C.c_int, [_P] * 8 + [C.c_int] * 4 + [_P] * 4),
    'eas_simota_assign_rows': (C.c_int, [_P] * 6 + [C.c_int] * 4 + [_P] * 4),
    'eas_det_decode': (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    'eas_det_decode_eval': (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    'eas_det_labels': (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    'eas_det_loss_workspace_doubles': (C.c_int64, []),
    'eas_det_loss': (C.c_int, [C.c_int] + [_P] * 8 + [C.c_int, C.c_int] + [_P] * 3 + [C.c_int] + [_P] * 4 + [C.c_int, _P, _P, _P]),
    'eas_upcat_fwd': (C.c_int, [_P, _P, _P, C.c_int64] + [C.c_int] * 5 + [_P]),
    'eas_upcat_bwd': (C.c_int, [_P, _P, _P, C.c_int64] + [C.c_int] * 5 + [_P]),
    'eas_upcat_planes_fwd': (C.c_int, [_P, _P, _P, C.c_int64] + [C.c_int] * 5 + [_P]),
    'eas_focus': (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'eas_event_voxel_grid': (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'eas_event_window_search': (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P]),
    'eas_event_histogram_dat_ranges': (C.c_int, [_P, _P] + [C.c_int] * 4 + [_P, _P, _P]),
    'eas_stacked_hist_event_sum': (C.c_int, [_P, _P] + [C.c_int] * 7 + [_P, _P]),
    'eas_lif_fwd': (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, _P, _P, _P, C.c_int, C.c_int64, _P]),
    'eas_lif_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float,
                              _P, _P, _P, C.c_int, C.c_int64, _P]),
    'eas_lif_bwd_patan': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, _P, _P,
                                    _P, _P, _P, C.c_int, C.c_int64, _P]),
    'eas_reduce_workspace_floats': (C.c_int64, [C.c_int64]),
    'eas_time_mean': (C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    'eas_channel_sum': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    'eas_bn_stats': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P]),
    'eas_bn_workspace_doubles': (C.c_int64, [C.c_int]),
    'eas_bn_lif_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, _P, _P,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'eas_bn_lif_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                 C.c_float, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'eas_bn_silu_fwd': (C.c_int, [_P] * 6 + [C.c_int] * 3 + [_P]),
    'eas_bn_silu_fwd_ex': (C.c_int, [_P] * 6 + [C.c_int] * 3 + [C.POINTER(EasBnPending), C.c_int, C.c_int, _P]),
    'eas_bn_stats_partial': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'eas_bn_lif_fwd_ex': (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, _P, _P,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(EasBnPending), _P, C.c_int, _P, _P, C.c_int, _P]),
    'eas_bn_lif_bwd_ex': (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                    C.c_float, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'eas_bn_lif_bwd_patan': (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, _P,
                                       _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'eas_bn_silu_bwd': (C.c_int, [_P] * 6 + [C.c_int] + [_P] * 4 + [C.c_int] * 5 + [_P]),
    'eas_arsnn_step_fwd': (C.c_int, [_P] * 14 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, _P]),
    'eas_arsnn_step_bwd': (C.c_int, [_P] * 13 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                                 C.c_float, C.c_int, C.c_int, C.c_int, _P]),
    'eas_arsnn_tail_fwd': (C.c_int, [_P] * 6 + [C.c_int] * 7 + [_P]),
    'eas_arsnn_tail_bwd': (C.c_int, [_P] * 6 + [C.c_int] * 7 + [_P]),
    'eas_arsnn_fused_step_fwd': (C.c_int, [_P] * 19 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'eas_smallconv_packed_floats': (C.c_int64, [C.c_int] * 3),
    'eas_smallconv_pack_weights': (C.c_int, [_P, C.c_int, _P]),
    'eas_smallconv_bwd_input_dual': (C.c_int, [_P] * 6 + [C.c_int] * 4 + [_P]),
    'eas_smallconv_fwd': (C.c_int, [_P] * 4 + [C.c_int] * 8 + [_P]),
    'eas_smallconv_bwd_input': (C.c_int, [_P] * 4 + [C.c_int] * 6 + [_P]),
    'eas_smallconv_bwd_weight': (C.c_int, [_P] * 5 + [C.c_int] * 7 + [_P]),
    'eas_smallconv_wgrad_workspace_floats': (C.c_int64, [C.c_int] * 3),
    'eas_conv_packed_weight_bytes': (C.c_int64, [C.c_int] * 4),
    'eas_conv_pack_weights': (C.c_int, [_P, _P] + [C.c_int] * 4 + [_P]),
    'eas_conv_pack_weights_many': (C.c_int, [_P, C.c_int, _P]),
    'eas_conv_fwd_supported': (C.c_int, [C.c_int] * 8),
    'eas_conv_wgrad_parts': (C.c_int, [C.c_int] * 7),
    'eas_conv_fwd': (C.c_int, [_P] * 4 + [C.c_int] * 8 + [_P, _P]),
    'eas_conv_fwd_act': (C.c_int, [_P] * 4 + [C.c_int] * 9 + [_P, _P]),
    'eas_conv_fwd_planes': (C.c_int, [_P] * 4 + [C.c_int] * 7 + [_P, C.c_int, _P]),
    'eas_spike_planes_from_f32': (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P]),
    'eas_spike_planes_to_f32': (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P]),
    'eas_conv_fwd_stats': (C.c_int, [_P] * 3 + [C.c_int] * 8 + [_P, _P, C.c_int, _P]),
    'eas_conv_fwd_stats_blocks': (C.c_int, [C.c_int] * 8),
    'eas_conv_dgrad_s2': (C.c_int, [_P] * 3 + [C.c_int] * 5 + [_P]),
    'eas_conv_dgrad_small_supported': (C.c_int, [C.c_int] * 5),
    'eas_conv_dgrad_small': (C.c_int, [_P] * 3 + [C.c_int] * 5 + [_P]),
    'eas_conv_wgrad_workspace_floats': (C.c_int64, [C.c_int] * 8),
    'eas_conv_wgrad': (C.c_int, [_P] * 4 + [C.c_int] * 8 + [_P]),
    'eas_conv_wgrad_partial': (C.c_int, [_P] * 3 + [C.c_int] * 8 + [_P]),
    'eas_conv_wgrad_planes_partial': (C.c_int, [_P] * 3 + [C.c_int] * 7 + [_P]),
    'eas_conv_wgrad_reduce_many': (C.c_int, [_P, C.c_int, _P]),
    'eas_spp_pool_fwd': (C.c_int, [_P, _P, C.c_int64] + [C.c_int] * 6 + [_P]),
    'eas_spp_pool_bwd': (C.c_int, [_P, _P, _P, C.c_int64] + [C.c_int] * 6 + [_P]),
    'eas_spp_pool_planes_fwd': (C.c_int, [_P, _P, C.c_int64] + [C.c_int] * 6 + [_P]),
    'eas_spp_pool_planes_bwd': (C.c_int, [_P, _P, _P, C.c_int64] + [C.c_int] * 6 + [_P]),
    'eas_conv_fwd_group_plan': (C.c_int, [C.POINTER(EasConvProblem), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'eas_conv_fwd_group': (C.c_int, [C.POINTER(EasConvProblem), C.c_int, C.c_int, C.c_int, _P]),
    'eas_bn_silu_fwd_group': (C.c_int, [C.POINTER(EasBnSiluFwdProblem), C.c_int, _P]),
    'eas_bn_silu_bwd_group': (C.c_int, [C.POINTER(EasBnSiluBwdProblem), C.c_int, _P]),
    'eas_conv_wgrad_group_plan': (C.c_int, [C.POINTER(EasWgradProblem), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'eas_conv_wgrad_group_partial': (C.c_int, [C.POINTER(EasWgradProblem), C.c_int, C.c_int, C.c_int, _P]),
    'eas_channel_sum_group': (C.c_int, [C.POINTER(EasChannelSumProblem), C.c_int, _P]),
    'eas_pred_dgrad_group': (C.c_int, [C.POINTER(EasPredDgradProblem), C.c_int, _P]),
    'eas_adam_table_entry_bytes': (C.c_int, []),
    'eas_adam_chunk': (C.c_int, []),
    'eas_adam_step': (C.c_int, [_P, C.c_int, C.c_longlong, C.c_double, C.c_double, C.c_double, _P]),
    'eas_adam_advance_steps': (C.c_int, [_P, C.c_int, _P]),
    'eas_adam_step_ex': (C.c_int, [_P, C.c_int, C.c_longlong, C.POINTER(EasAdamHyper), _P]),
    'eas_adam_advance_steps_ex': (C.c_int, [_P, C.c_int, _P, _P]),
}


class EasHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into eas_snn_amd/libeas_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise EasHipError('building libeas_hip.so failed')
    return LIB_PATH


def _bind_host_hip_runtime():
    """libeas_hip.so is linked without a HIP runtime (csrc/Makefile): make the runtime PyTorch uses visible to it.
    Loading torch first and promoting its bundled libamdhip64 to the global symbol scope keeps ONE runtime in
    the process, so device pointers and streams handed over from torch are valid inside the kernels' launches."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if not os.path.exists(cand):
        cand = 'libamdhip64.so'
    C.CDLL(cand, mode=C.RTLD_GLOBAL)


_PRIVATE_STREAMS = []


def private_stream():
    """A torch stream on a HIP stream of its own (hipStreamCreateWithFlags), NOT one of the 32 pooled streams ``torch.cuda.Stream()`` hands out
    round-robin.  Every stream that may end up inside a graph capture is made here: ProcessGroupNCCL takes its internal stream from the same
    pool, and when a pooled stream that happens to be THAT one is put into capture, the process group's watchdog thread -- which keeps
    querying the end events of collectives recorded on it -- fails with hipErrorCapturedEvent ("operation not permitted on an event last
    recorded in a capturing stream") and aborts the process.  Seen in roughly one of ten runs of the one-rank RCCL bench."""
    import torch
    _bind_host_hip_runtime()
    rt = C.CDLL(None)                    # the process-wide symbol scope: the HIP runtime torch loaded
    handle = C.c_void_p()
    # (default priority on purpose: a stream of another priority -- higher or lower, the chain's or the slab kernels' -- inside a recorded
    # step costs +7.7 ms per replay on this runtime, HISTORY.md D round 6)
    rc = rt.hipStreamCreateWithFlags(C.byref(handle), C.c_uint(1))          # hipStreamNonBlocking
    if rc != 0 or not handle.value:
        raise EasHipError(f'hipStreamCreateWithFlags failed ({rc})')
    st = torch.cuda.ExternalStream(handle.value)
    _PRIVATE_STREAMS.append(st)          # lives as long as the process (graphs recorded on it are replayed until the end)
    return st


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EasHipError(
                f'{LIB_PATH} is missing: the HIP extension has not been built and eas_snn_amd has no CPU fallback. '
                'Run `python -c "import __graft_entry__ as g; g.build()"`.')
        _bind_host_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)      # AttributeError = ABI mismatch, also loud
            fn.restype = res
            fn.argtypes = args
        if handle.eas_abi_version() != ABI_VERSION:
            raise EasHipError('libeas_hip.so ABI version mismatch; rebuild')
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().eas_status_string(status).decode()
        raise EasHipError(f'{what}: status {status} ({msg})')


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_RAW_STREAM = None


def stream():
    """raw hipStream_t of torch's current stream on the current device (called once per kernel launch: the raw getter avoids
    building a torch.cuda.Stream object every time)"""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        if raw is not None:
            _RAW_STREAM = lambda: raw(torch.cuda.current_device())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return _RAW_STREAM()
