/*
 * eas_hip.h -- C ABI of libeas_hip.so: the MI355X (gfx950) hot path of EAS-SNN.
 *
 * The reference (Windere/EAS-SNN) has NO native code on this path: everything below
 * replaces chains of PyTorch/numpy calls.  Each entry point cites the reference
 * code it replaces (paths relative to the reference repository root).
 *
 * Conventions:
 *   - every function is asynchronous on the caller-supplied HIP stream (hipStream_t passed as void*);
 *   - all pointers are DEVICE pointers to contiguous row-major buffers owned by the caller
 *     (the PyTorch allocator); nothing is allocated, freed or retained inside;
 *   - return value: 0 = ok, negative = error (EAS_ERR_*), never throws, never exits;
 *   - no host synchronisation inside (graph-capturable).
 */
#ifndef EAS_HIP_H
#define EAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* eas_stream_t; /* hipStream_t */

#define EAS_OK 0
#define EAS_ERR_INVALID_ARG (-1)
#define EAS_ERR_UNSUPPORTED (-2)
#define EAS_ERR_LAUNCH (-3)

/* surrogate gradient ids (spikingjelly surrogate.ATan / surrogate.Sigmoid, chosen at
 * yolox/exp/event_yolox_base.py:144-151; Rectangle: yolox/models/activation.py:17-30) */
#define EAS_SG_ATAN 0
#define EAS_SG_SIGMOID 1
#define EAS_SG_RECT 2
/* 'patan' = EfficientNoisySpikeII(InvArcTanh(alpha), p=0) (yolox/exp/event_yolox_base.py:145-150,
 * yolox/models/activation.py:121-130,181-205): Heaviside forward, backward through
 * sigma(u) = atan(pi/2 |alpha| u)/pi + 1/2 with a LEARNABLE alpha that lives on the device.  Only through
 * eas_lif_bwd_patan / eas_bn_lif_bwd_patan (alpha pointer + grad_alpha output). */
#define EAS_SG_PATAN 3

/* neuron flags */
#define EAS_LIF_HARD_RESET 1   /* v = (1-s)*h + s*v_reset ; else soft: v = h - s*v_th        */
#define EAS_LIF_DECAY_INPUT 2  /* h = v + (x-(v-v_reset))*k ; else h = v - (v-v_reset)*k + x */
#define EAS_LIF_DETACH_RESET 4 /* spike is a constant in the reset term during backward      */
#define EAS_LIF_FIRE_STRICT 8  /* fire on h - v_th > 0 (in-repo LIFCell); else >= 0 (spikingjelly heaviside) */

int eas_abi_version(void);
/* Kernel-instance trace (test infrastructure): between eas_kernel_trace_begin() and eas_kernel_trace_dump() every launch of the library
 * records the (demangled) symbol of the device kernel it starts -- the names rocprofv3 --kernel-trace reports.  dump switches the
 * trace off and writes the distinct symbols, newline separated, into buf (cap bytes incl. the terminator); returns the bytes needed. */
void eas_kernel_trace_begin(void);
int64_t eas_kernel_trace_dump(char* buf, int64_t cap);
/* number of device kernels the library has launched since it was loaded (a host-side count: captured launches count when they are
 * captured, not when the graph is replayed) */
int64_t eas_launch_counter(void);
const char* eas_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * K1  event -> count frames.  Replaces GEN1Dataset.slice_events + agrregate('sum'|'micro_sum')
 *     (yolox/data/datasets/gen1.py:313-328, 333-349, 355-360; same code in ncaltech.py:227-237,
 *     368-378, rvt_gen4.py:411-426, gen4.py:445-460).
 * Events of sample b are [sample_offsets[b], sample_offsets[b+1]) (t ascending inside a sample).
 * out[b][k][c][y][x] (int32, zeroed by the call) counts events of micro-slice k and polarity
 * channel c (c = 0 for p == 0, 1 otherwise) with
 *     window = (t_last - t_first) / Tm  (integer floor),  k = (t - t_first) / window,
 * dropped when window == 0 or k >= Tm.  Events with x >= W or y >= H are dropped and counted
 * in *oob_count (may be NULL).  Integer atomics: bit-exact. */
int eas_event_histogram(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                        const int64_t* sample_offsets, int B, int Tm, int H, int W, int32_t* out,
                        uint32_t* oob_count, eas_stream_t stream);

/* The same count frames straight from Prophesee .dat event records (8 bytes each: u32 t, u32 packed with x = bits 0..13,
 * y = bits 14..27, p = bit 28; load_td_data, yolox/utils/psee_loader/io/dat_events_tools.py:29-54): decode, micro-slice
 * window and histogram in one pass over the raw file bytes after the header.  `records` is 8-byte aligned. */
int eas_event_histogram_dat(const void* records, int64_t nev, const int64_t* sample_offsets, int B, int Tm, int H, int W,
                            int32_t* out, uint32_t* oob_count, eas_stream_t stream);

/* int32 counts [F][H][W] -> fp32 canvas [F][Hc][Wc], zero padded bottom/right (top-left placement,
 * as gen1.py:447-455 does with scale 1).  Replaces np.stack + astype + pad + trainer.py:99 cast. */
int eas_counts_to_canvas(const int32_t* counts, int64_t F, int H, int W, int Hc, int Wc, float* out,
                         eas_stream_t stream);

/* eas_event_histogram + eas_counts_to_canvas in one call: fp32 frames [B][Tm][2][Hc][Wc] (zero padded bottom/right).  Dense
 * streams are binned in LDS and written to the canvas directly (the int32 counts never reach HBM); otherwise the counts go
 * through scratch_counts [B][Tm][2][H][W].  Values are identical either way. */
int eas_event_frames(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                     const int64_t* sample_offsets, int B, int Tm, int H, int W, int Hc, int Wc, float* frames,
                     int32_t* scratch_counts, uint32_t* oob_count, eas_stream_t stream);

/* Letterbox / jitter augmentation of the count frames on the device (GEN1Dataset.get_random_data,
 * gen1.py:433-521: batch_resize with cv2.INTER_LINEAR :423-431, paste into a zero canvas, left-right flip) fused with the
 * fp32 cast of trainer.py:99.  counts [B][F][H][W] int32; params [B][5] int32 = (nw, nh, dx, dy, flip) per sample (the
 * host draws them exactly as the reference does, eas_snn_amd/data.py); out [B][F][Hc][Wc] fp32.  The paste rectangle must lie
 * inside the canvas.  cv2 itself is absent from this image: the resize follows OpenCV's published algorithm. */
int eas_counts_letterbox(const int32_t* counts, const int32_t* params, int B, int F, int H, int W, int Hc, int Wc, float* out,
                         eas_stream_t stream);

/* Bilinear-in-time voxel grid, to_voxel_grid_numpy (yolox/utils/event_reps.py:30-89).
 * out[b][bin][y][x] float64 (zeroed by the call); polarity 0 counts as -1. */
int eas_event_voxel_grid(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                         const int64_t* sample_offsets, int B, int n_bins, int H, int W, double* out,
                         eas_stream_t stream);

/* Voxel cube, to_voxel_cube_numpy (yolox/utils/event_reps.py:92-138; call sites gen1.py:371-372, ncaltech.py:259):
 * out[b][slice][channel][y][x] int32 counts (zeroed by the call), 2*tbins channels, channel = (p+1)*(tbin+1)-1 as the
 * reference computes it.  The reference returns these counts as float64. */
int eas_event_voxel_cube(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                         const int64_t* sample_offsets, int B, int num_slices, int tbins, int H, int W, int32_t* out,
                         eas_stream_t stream);

/* Time surfaces at the end of each micro-slice, GEN1Dataset.agrregate(method='timesurface') (gen1.py:362-369) =
 * slice_events + to_timesurface_numpy (yolox/utils/event_reps.py:141-160): out[b][slice][p][y][x] float64 =
 * exp(-((slice+1)*window + t0 - latest_timestamp) / tau).  workspace: B*num_slices*2*H*W uint32 (latest timestamps). */
int eas_event_time_surface(const uint32_t* t, const uint16_t* x, const uint16_t* y, const uint8_t* p, int64_t nev,
                           const int64_t* sample_offsets, int B, int num_slices, int H, int W, double tau,
                           uint32_t* workspace, double* out, eas_stream_t stream);

/* Timestamp-window search on the device: replaces GEN1Dataset.search_events (yolox/data/datasets/gen1.py:217-232) and the
 * PSEELoader.seek_time / load_delta_t calls under it (yolox/utils/psee_loader/io/psee_loader.py:128-238).  `records` is the
 * record area of one or more .dat recordings in HBM (8 bytes per event, timestamps ascending inside a recording), recording f
 * being records [file_offsets[f], file_offsets[f+1]).  For label b (recording file_id[b] -- NULL = recording 0 --, label time
 * label_t[b] in us) ranges[2b], ranges[2b+1] receive the first / one-past-last RECORD index (into `records`) of the events of
 * [label_t + window_lo, + (window_hi - window_lo)); an empty window steps back by its own length, num_slice + 2 attempts at most
 * (the reference's zero_trigger rule).  The reader's behaviour is reproduced exactly, including seek_time's reset for targets
 * <= 0 and its bisection probes on recordings of more than 100 000 events (a probe that equals the target leaves the reader one
 * event past it).  Index work: bit-exact. */
int eas_event_window_search(const void* records, const int64_t* file_offsets, int F, const int32_t* file_id, const int64_t* label_t, int B,
                            int64_t window_lo, int64_t window_hi, int num_slice, int64_t* ranges, eas_stream_t stream);
/* eas_event_histogram_dat for samples given as device-resident record ranges (the output of eas_event_window_search; the ranges
 * of neighbouring labels overlap): out int32 [B][Tm][2][H][W], zeroed by the call.  No host read of the ranges. */
int eas_event_histogram_dat_ranges(const void* records, const int64_t* ranges, int B, int Tm, int H, int W, int32_t* out,
                                   uint32_t* oob_count, eas_stream_t stream);

/* Config-4 input (BASELINE configs[3]): RVT stacked histogram -> per-polarity counts on the model canvas.  Replaces
 * RVTGEN4Dataset.generate_slices(..., method='event_sum') (yolox/data/datasets/rvt_gen4.py:109-125: reshape
 * [n][2*nbins][H][W] -> [n][2][nbins][H][W], sum over the bins, zero slices in FRONT when fewer than num_slice
 * representations exist) followed by the validation letterbox at scale 1 = zero padding to the canvas (rvt_gen4.py:516-533).
 * hist: u8 [B][Tm][2*nbins][H][W] (channel = polarity*nbins + bin); n_valid (nullable, int32 [B]): sample b supplies
 * only its first n_valid[b] slices, which become the LAST n_valid[b] output slices.  out: fp32 [B][Tm][2][Hc][Wc]
 * (the model input [B][Tl=1][Tm][2][Hc][Wc]), Hc >= H, Wc >= W, Wc % 16 == 0.  Integer sums (<= 255*nbins): bit-exact. */
int eas_stacked_hist_event_sum(const uint8_t* hist, const int32_t* n_valid, int B, int Tm, int nbins, int H, int W, int Hc,
                               int Wc, float* out, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  multi-step (P)LIF neuron.  Replaces spikingjelly ParametricLIFNode / LIFNode
 *     multi_step_forward (call site yolox/utils/utils_snn.py:44-53): per step
 *         h = charge(v, x_t);  s_t = H(h - v_th);  v = reset(h, s_t)
 * x, spikes: [T][M] fp32.  v_in: [M] initial membrane potential (NULL = v_reset, the state after
 * reset_net); v_out: [M] final membrane potential (NULL = not needed; may alias v_in).  decay: k = sigmoid(*w_logit) if w_logit != NULL (PLIF),
 * else k = k_const (LIF: 1/tau).  h_save (nullable): [T][M] pre-fire potential kept for backward.
 * mean_out (nullable): [M] mean of the spikes over T (firing-rate readout,
 * yolox/models/spiking_yolo_pafpn.py:98).  */
int eas_lif_fwd(const float* x, const float* v_in, float* v_out, const float* w_logit, float k_const, float v_th,
                float v_reset, int flags, float* spikes, float* h_save, float* mean_out, int T, int64_t M,
                eas_stream_t stream);

/* Backward of eas_lif_fwd.  grad_s: [T][M] (nullable), grad_mean: [M] (nullable, adds grad_mean/T
 * to every step), h_save from forward, v_init: [M] initial state used in forward (NULL = v_reset),
 * x: only read when EAS_LIF_DECAY_INPUT and w_logit != NULL.  Writes grad_x [T][M] and, when
 * grad_w != NULL, *grad_w = dL/dw_logit (deterministic two-stage reduction through
 * workspace, which must hold eas_reduce_workspace_floats(M) floats). */
int eas_lif_bwd(const float* grad_s, const float* grad_mean, const float* h_save, const float* v_init,
                const float* x, const float* w_logit, float k_const, float v_th, float v_reset, int flags,
                int surrogate, float alpha, float* grad_x, float* grad_w, float* workspace, int T, int64_t M,
                eas_stream_t stream);

int64_t eas_reduce_workspace_floats(int64_t M);

/* eas_lif_bwd for the learnable arctan surrogate (EAS_SG_PATAN): alpha is a DEVICE scalar (no host read), the slope uses
 * |*alpha|; grad_alpha (nullable) receives dL/dalpha = sign(alpha) * sum over neuron-steps of dL/ds_t * u/2 / (1 + (pi/2 |alpha| u)^2),
 * u = h_t - v_th, where dL/ds_t includes the path through the reset unless EAS_LIF_DETACH_RESET (fixed-order reduction). */
int eas_lif_bwd_patan(const float* grad_s, const float* grad_mean, const float* h_save, const float* v_init,
                      const float* x, const float* w_logit, float k_const, float v_th, float v_reset, int flags,
                      const float* alpha, float* grad_alpha, float* grad_x, float* grad_w, float* workspace, int T, int64_t M,
                      eas_stream_t stream);

/* mean over the leading T axis: [T][M] -> [M] (out_features[f].mean(axis=0)). */
int eas_time_mean(const float* x, float* out, int T, int64_t M, eas_stream_t stream);
/* out[c] = sum over (n, pixel) of g[n][c][pixel], g fp32 [N][C][HW]: the bias gradient of a convolution (ATen convolution_backward's
 * grad_bias, i.e. grad_y.sum((0, 2, 3)); the prediction convolutions of the head, yolo_head.py:60-90, carry biases).  One block per
 * channel, summed in double in a fixed order: deterministic. */
int eas_channel_sum(const float* g, float* out, int N, int C, int HW, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4 (BN + LIF half of the fused conv -> BN -> LIF step).  Replaces
 * layer.BatchNorm2d(step_mode='m') + ParametricLIFNode after every converted BaseConv
 * (yolox/models/network_blocks.py:52-53 after yolox/utils/utils_snn.py:28-53).
 * y: conv output [T][N][C][HW] fp32.  Statistics are per channel over T*N*HW. */

/* per-channel batch statistics: mean[C], invstd[C] (biased variance, eps inside), and the
 * running-stat update running = (1-momentum)*running + momentum*batch (unbiased variance),
 * skipped when running_mean == NULL.  replicas >= 1: y stands for that many identical copies (the T-broadcast of
 * one adaptive frame, spiking_yolox.py:52-57): mean/var are those of one copy, the unbiased-variance count is
 * TN*HW*replicas.  workspace: eas_bn_workspace_doubles(C) doubles. */
int eas_bn_stats(const float* y, int TN, int C, int HW, int replicas, float eps, float momentum, float* mean,
                 float* invstd, float* running_mean, float* running_var, double* workspace, eas_stream_t stream);
int64_t eas_bn_workspace_doubles(int C);

/* z = gamma*(y-mean)*invstd + beta, then the LIF recurrence over T (same neuron arguments as
 * eas_lif_fwd).  Nothing but y needs to be kept for backward.  y_bcast != 0: y is ONE plane [N][C][HW] that
 * feeds all T steps (identical input frames); eas_bn_lif_bwd then returns grad_y [N][C][HW] summed over the steps. */
int eas_bn_lif_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   const float* v_in, float* v_out, const float* w_logit, float k_const, float v_th, float v_reset,
                   int flags, float* spikes, float* mean_out, int T, int N, int C, int HW, int y_bcast,
                   eas_stream_t stream);

/* Backward, pass 1: recompute the forward from y, run the LIF backward and reduce per channel
 * sum(dz), sum(dz*xhat) (-> grad_beta, grad_gamma) and the neuron's grad_w.  Pass 2 (apply):
 * grad_y = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) in training mode, or
 * gamma*invstd*dz when batch_stats == 0 (eval / frozen BN).  Both passes are issued by this call.
 * workspace: eas_bn_workspace_doubles(C) doubles. */
int eas_bn_lif_bwd(const float* grad_s, const float* grad_mean, const float* y, const float* mean,
                   const float* invstd, const float* gamma, const float* beta, const float* v_init,
                   const float* w_logit, float k_const, float v_th, float v_reset, int flags, int surrogate,
                   float alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                   float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast, eas_stream_t stream);

/* ---- variants that save launches and copies inside a network (same arithmetic as the calls above) ----
 * eas_bn_stats_partial: only the per-chunk partial sums of eas_bn_stats; returns the number of chunks per channel (> 0)
 *   or a negative status.  The finalize (mean, invstd, running statistics) then happens inside the consuming kernel:
 *   pass an EasBnPending to eas_bn_lif_fwd_ex / eas_bn_silu_fwd_ex, whose mean / invstd arguments become OUTPUTS
 *   (needed by the backward calls).
 * eas_bn_lif_fwd_ex: residual (nullable, [T][N][C][HW]): spikes_out = spikes + residual, the SEW shortcut of Bottleneck
 *   (network_blocks.py:99-104) without a separate addition; out_ctot (0 or >= C): the spikes are written as C consecutive
 *   channels of a [T][N][out_ctot][HW] tensor, `spikes` pointing at the first of them -- the concatenations of CSPLayer
 *   (network_blocks.py:183-188) happen in place.  mean_out is the rate of the spikes themselves (without residual).
 *   spikes_planes: write the output as SPIKE PLANES instead of fp32 (`spikes` must then be NULL): bf16 in blocks of 8 channels,
 *   [T*N][out_ctot/8][HW][8] -- spikes and SEW sums are exact in bf16, every consumer's matrix-core fragment (8 consecutive input
 *   channels of one pixel) is one 16-byte load, at half the bytes of fp32 (eas_conv_fwd_planes, eas_conv_wgrad_planes_partial,
 *   eas_spike_planes_to_f32).  residual_planes / residual_ctot: the SEW shortcut given as planes ([T*N][residual_ctot/8][HW][8]).
 *   C, out_ctot, residual_ctot multiples of 8.
 * eas_bn_lif_bwd_ex: grad_s_ctot (0 or >= C): grad_s is such a channel slice of a wider gradient tensor.
 * y_ctot (0 or >= C) in all three: y -- and grad_y in the backward -- are C consecutive channels of a [T][N][y_ctot][HW]
 *   tensor, the pointers at the first of them: ONE convolution (concatenated weights) feeds the two 1x1 branches of a CSPLayer
 *   (network_blocks.py:175-188), its input is read once and its input gradient needs no addition of two branch gradients. */
typedef struct {
    const double* partial;   /* workspace filled by eas_bn_stats_partial; NULL = statistics already final */
    int chunks;              /* its return value */
    int replicas;            /* as eas_bn_stats */
    double count;            /* TN * HW */
    float eps, momentum;
    float* running_mean;     /* nullable pair */
    float* running_var;
    int pitch;               /* partials allocated per channel in `partial`: 0 = what eas_bn_stats_partial writes (64); the statistics buffer of
                                eas_conv_fwd_stats: its nb (= chunks; a channel slice: partial = stats + first_channel * nb * 2) */
} EasBnPending;
int eas_bn_stats_partial(const float* y, int y_ctot, int TN, int C, int HW, double* workspace, eas_stream_t stream);
int eas_bn_lif_fwd_ex(const float* y, int y_ctot, float* mean, float* invstd, const float* gamma, const float* beta,
                      const float* v_in, float* v_out, const float* w_logit, float k_const, float v_th, float v_reset,
                      int flags, float* spikes, float* mean_out, int T, int N, int C, int HW, int y_bcast,
                      const EasBnPending* pending, const float* residual, int out_ctot, void* spikes_planes, const void* residual_planes,
                      int residual_ctot, eas_stream_t stream);
int eas_bn_lif_bwd_ex(const float* grad_s, int grad_s_ctot, const float* grad_mean, const float* y, int y_ctot, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, const float* v_init,
                      const float* w_logit, float k_const, float v_th, float v_reset, int flags, int surrogate,
                      float alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                      float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast, eas_stream_t stream);
/* eas_bn_lif_bwd_ex for the learnable arctan surrogate (see eas_lif_bwd_patan). */
int eas_bn_lif_bwd_patan(const float* grad_s, int grad_s_ctot, const float* grad_mean, const float* y, int y_ctot, const float* mean,
                         const float* invstd, const float* gamma, const float* beta, const float* v_init,
                         const float* w_logit, float k_const, float v_th, float v_reset, int flags, const float* alpha,
                         float* grad_alpha, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                         float* grad_w, double* workspace, int T, int N, int C, int HW, int y_bcast, eas_stream_t stream);

/* BatchNorm2d + SiLU fused for the real-valued BaseConv blocks (stem, PAFPN neck, head:
 * yolox/models/network_blocks.py:52-53 with nn.SiLU); y: conv output [N][C][HW]; mean/invstd from eas_bn_stats
 * (training) or the running statistics (eval).  Backward = two passes like eas_bn_lif_bwd; only y is kept.
 * workspace: eas_bn_workspace_doubles(C) doubles. */
int eas_bn_silu_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                    float* out, int N, int C, int HW, eas_stream_t stream);
/* eas_bn_silu_fwd with the statistics finalize folded in (see EasBnPending above): mean / invstd become outputs.
 * out_ctot (0 or >= C): out is C consecutive channels of a [N][out_ctot][HW] tensor, the pointer at the first of them -- the
 * torch.cat((x_1, x_2), dim=1) of the ANN CSPLayer (network_blocks.py:183-188) happens in place. */
int eas_bn_silu_fwd_ex(const float* y, float* mean, float* invstd, const float* gamma, const float* beta,
                       float* out, int N, int C, int HW, const EasBnPending* pending, int out_ctot, int y_ctot, eas_stream_t stream);
/* y_ctot (0 or >= C) in both: y -- and grad_y in the backward -- are C consecutive channels of a [N][y_ctot][HW] tensor (the output of ONE
 * convolution with concatenated weights feeding two BN + SiLU layers: conv1 | conv2 of the real-valued CSPLayer, the first cls / reg
 * tower convolutions of the head).
 * grad_out_ctot (0 or >= C): grad_out is such a channel slice of a wider gradient tensor (no contiguous copy of the slice).  Channels
 * whose N*HW values fit the registers of one block (the 8x10 / 16x20 maps) run both passes in one launch. */
int eas_bn_silu_bwd(const float* grad_out, const float* y, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, int batch_stats, float* grad_y, float* grad_gamma, float* grad_beta,
                    double* workspace, int N, int C, int HW, int grad_out_ctot, int y_ctot, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  adaptive sampler step (AdaptiveRSNNEmbedding.forward loop body,
 *     yolox/models/embedding.py:170-201 + update :132-139; tail :203-217), dense masked form.
 * conv_in / conv_rec: [N][2*C2][HW] outputs of input_conv(ev[t]) / gate_conv(spike): channels
 * [0,C2) gate pre-activation, [C2,2*C2) current.  State per (n,c,hw): v, vsum (fp32), seg, t_last
 * (int32).  agg: [Ts][N][C2][HW].  readout: 0 sum, 1 last, 2 avg; 3 (step kernels only) = plain gated recurrence of
 * SpikingEmbedding / LIFEmbedding (embedding.py:229-316, 28-76): vsum is a running sum that is never reset and no
 * segment is written.  v_reset_mode: 0 hard reset to
 * v_reset, 1 soft (v - thresh*spike).  Saved for backward: gate, vn (pre-reset), seg_before,
 * t_last_before (all nullable in inference).
 * v == vsum == NULL (forward) / v_prev == vsum_prev == NULL (backward): the first step of a sequence -- zero potentials and sums,
 * seg = 0, t_last = -1 (embedding.py:159-167); seg / t_last are then outputs only, so the caller needs no zero fills. */
/* seg (segments written so far, 0..Ts) and t_last (step of the last spike, -1..Tm-1) are int8 tensors: Ts, Tm <= 127. */
int eas_arsnn_step_fwd(const float* conv_in, const float* conv_rec, const float* v, const float* vsum,
                       int8_t* seg, int8_t* t_last, float* agg, float* v_out, float* vsum_out, float* spike_out,
                       float* gate_save, float* vn_save, int8_t* seg_before, int8_t* t_last_before, int t,
                       int Ts, int readout, int spike_attach, float thresh, float v_reset, int soft_reset, int N,
                       int C2, int HW, eas_stream_t stream);

/* eas_arsnn_step_fwd with the SECOND convolutions of the input stack and of the gate stack inside the launch (embedding.py:171-176;
 * C2 = 2): their 4 + 4 output planes per pixel never reach HBM.  a_in [N][4][H][W] = ReLU(conv1_in(events of this micro-step)),
 * a_g [N][4][H][W] = ReLU(conv1_gate(spikes entering the step)) or NULL -- then r_const [4][H][W] (nullable = zeros) is the gate stack's
 * output for every sample (step 0: the constant-zero spike input).  wr_in / wr_g: eas_smallconv_pack_weights mode 0, o_total 4;
 * b_in / b_g [4] nullable.  State arguments as eas_arsnn_step_fwd.  Results are bit-identical to the separate launches. */
int eas_arsnn_fused_step_fwd(const float* a_in, const float* wr_in, const float* b_in, const float* a_g, const float* wr_g, const float* b_g,
                             const float* r_const, const float* v, const float* vsum, int8_t* seg, int8_t* t_last, float* agg, float* v_out,
                             float* vsum_out, float* spike_out, float* gate_save, float* vn_save, int8_t* seg_before, int8_t* t_last_before,
                             int t, int Ts, int readout, int spike_attach, float thresh, float v_reset, int soft_reset, int N, int H, int W, int k,
                             eas_stream_t stream);

int eas_arsnn_step_bwd(const float* g_v_out, const float* g_vsum_out, const float* g_spike, const float* g_agg,
                       const float* v_prev, const float* vsum_prev, const float* gate_save, const float* vn_save,
                       const int8_t* seg_before, const int8_t* t_last_before, float* g_conv, float* g_v_prev,
                       float* g_vsum_prev, int t, int Ts, int readout, int spike_attach, float thresh,
                       float v_reset, int soft_reset, float sg_alpha, int N, int C2, int HW, eas_stream_t stream);

/* tail write (embedding.py:203-217): at elements whose LAST spike is 0 and seg < Ts add
 * (sum|last|avg readout) * (write_zero ? 0 : 1) into agg[seg]; optional relu (abs=True, :218-220)
 * is applied by eas_relu_inplace. */
int eas_arsnn_tail_fwd(const float* v, const float* vsum, const float* spike_last, const int8_t* seg,
                       const int8_t* t_last, float* agg, int Tm, int Ts, int readout, int write_zero, int N,
                       int C2, int HW, eas_stream_t stream);
int eas_arsnn_tail_bwd(const float* g_agg, const float* spike_last, const int8_t* seg, const int8_t* t_last,
                       float* g_v, float* g_vsum, int Tm, int Ts, int readout, int write_zero, int N, int C2,
                       int HW, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Direct convolution for the sampler's conv stacks, Conv2d(2->4,k,p=k/2) [+ReLU+Conv2d(4->4,k)]
 * (yolox/models/embedding.py:106-111; replaces the F.conv2d calls and their autograd): stride 1, "same" zero
 * padding, NCHW fp32, (Cin,Cout) in {(2,4),(4,4),(2,2)}, k in {3,5,7}.  LDS-tiled (16x64 output tile + halo for all
 * input channels staged once; every output channel computed from registers with packed fp32 FMAs).
 *
 * The kernels read their weights through the scalar cache in the order wr[n_in][k][k][o_total] (output channel fastest);
 * eas_smallconv_pack_weights arranges up to 8 weight tensors per launch: mode 0 (forward) wr[i][ky][kx][o_off + o] = w[o][i][ky][kx],
 * mode 1 (input gradient) wr[co][ky][kx][o_off + ci] = w[co][ci][k-1-ky][k-1-kx].  wr: eas_smallconv_packed_floats(n_in, k, o_total)
 * floats, 16-byte aligned, n_in = Cin (mode 0) or Cout (mode 1). */
typedef struct {
    const float* w;          /* [Cout][Cin][k][k] */
    float* wr;
    int Cin, Cout, k, mode;
    int o_total, o_off;      /* the packed tensor may hold several weight tensors side by side along its output axis */
} EasSmallconvPackJob;
int64_t eas_smallconv_packed_floats(int n_in, int k, int o_total);
int eas_smallconv_pack_weights(const EasSmallconvPackJob* jobs, int njobs, eas_stream_t stream);
/* x: [N][Cin][H][W], wr: weights packed with mode 0 (o_total = Cout), b: [Cout] (nullable), y: [N][Cout][H][W]; relu != 0 applies max(.,0).
 * x_tm > 0: x is the COLLATED micro-slice tensor [N / x_tm][x_tm][Cin][H][W] as the data loader hands it over (trainer.py:99) and image
 * n = t * (N / x_tm) + s of the convolution is micro-slice x_tm - 1 - t of sample s -- the "reshape + flip(time) + time first" of
 * embedding.py:147-156 done by the load addresses instead of a copy of the input; y stays [N][Cout][H][W] in that time-major order.
 * x_tm = 0: x is [N][Cin][H][W].  N % x_tm == 0. */
int eas_smallconv_fwd(const float* x, const float* wr, const float* b, float* y, int N, int Cin, int Cout, int H,
                      int W, int k, int relu, int x_tm, eas_stream_t stream);
/* grad_x = correlation of grad_y with the flipped, channel-transposed filter (wr: packed with mode 1, o_total = Cin); when
 * relu_mask != NULL (the ReLU output that fed this conv's input, i.e. the tensor the gradient flows back into), grad_x is zeroed
 * where relu_mask <= 0 (fused ReLU backward). */
int eas_smallconv_bwd_input(const float* grad_y, const float* wr, const float* relu_mask, float* grad_x, int N, int Cin,
                            int Cout, int H, int W, int k, eas_stream_t stream);
/* The input gradients of TWO 4 -> 4 convolutions that received the same grad_y [N][4][H][W] (the second convolutions of the sampler's
 * input stack and gate stack) in one pass over grad_y: wr = both weights packed with mode 1, o_total 8, o_off 0 and 4. */
int eas_smallconv_bwd_input_dual(const float* grad_y, const float* wr, const float* mask_a, const float* mask_b, float* grad_xa,
                                 float* grad_xb, int N, int H, int W, int k, eas_stream_t stream);
/* grad_w [Cout][Cin][k][k] and grad_b [Cout] (nullable); deterministic two-stage reduction through
 * workspace (eas_smallconv_wgrad_workspace_floats(Cin,Cout,k) floats).  x_tm: as eas_smallconv_fwd (the layout of x; grad_y is always
 * [N][Cout][H][W]); the images are visited in the same order either way, so the sums are bit-identical to those over a flipped copy. */
int eas_smallconv_bwd_weight(const float* grad_y, const float* x, float* grad_w, float* grad_b, float* workspace,
                             int N, int Cin, int Cout, int H, int W, int k, int x_tm, eas_stream_t stream);
int64_t eas_smallconv_wgrad_workspace_floats(int Cin, int Cout, int k);

/* ---------------------------------------------------------------------------------------------
 * K4 (conv half)  dense 1x1 / 3x3 convolutions of the conv -> BN -> LIF step on the matrix cores.
 *     Replaces SeqToANNContainer(nn.Conv2d) / nn.Conv2d inside BaseConv (yolox/models/network_blocks.py:31-56
 *     after yolox/utils/utils_snn.py:25-27; ATen/MIOpen convolution forward and backward).
 * fp32 NCHW tensors; padding = ksize/2; ksize in {1,3}; stride in {1,2}; no groups/dilation.  Every fp32 operand
 * is split into exact bf16 terms and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (csrc/conv_mfma.hip), i.e. an fp32-accumulated sum of the same products an fp32 convolution forms.
 *
 * eas_conv_pack_weights: w[Cout][Cin][k][k] -> MFMA A-fragment order, 3 bf16 terms; mode 0 for the forward conv,
 *     mode 1 (transposed + flipped) for the input gradient of a stride-1 conv, mode 2 (transposed, grouped by input-pixel
 *     parity class) for the input gradient of a stride-2 3x3 conv.  `packed` holds
 *     eas_conv_packed_weight_bytes(Cout,Cin,k,mode) bytes.
 * eas_conv_fwd: y[NI][Cout][Ho][Wo] (+bias[Cout] if not NULL).  x_terms = 1: x holds small integers (spikes and
 *     their SEW sums; *inexact_flag, if not NULL, is OR-ed with 1 when an element is not exact in bf16);
 *     x_terms = 3: general fp32 x.  Input gradient of a stride-1 conv: call with x = grad_y, Cin/Cout swapped,
 *     weights packed in mode 1, x_terms = 3. */
int64_t eas_conv_packed_weight_bytes(int Cout, int Cin, int ksize, int mode);
int eas_conv_pack_weights(const float* w, void* packed, int Cout, int Cin, int ksize, int mode, eas_stream_t stream);
/* the same for many weight tensors in one launch: `jobs` is a device array of njobs x 8 int64
 * {weight pointer, packed pointer, Cout, Cin, ksize, mode, second weight pointer or 0, Cout of the first}.  With a second pointer the
 * packed weight is the concatenation of the two tensors along Cout (Cout = the total; modes 0 and 1): ONE convolution then computes two
 * layers that read the same input -- conv1 | conv2 of a CSPLayer (network_blocks.py:175-188), the first cls / reg tower convolutions of
 * the head (yolo_head.py) -- without a torch.cat of their weights. */
int eas_conv_pack_weights_many(const void* jobs, int njobs, eas_stream_t stream);
int eas_conv_fwd(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi,
                 int Wi, int ksize, int stride, int x_terms, int* inexact_flag, eas_stream_t stream);
/* y = act(conv(x) + bias) in one kernel; act: 0 none, 1 SiLU (x * sigmoid(x)).  The eval-mode form of a real-valued BaseConv once
 * fuse_model folded its BatchNorm into weights and bias (yolox/utils/model_utils.py:35-80, network_blocks.py:55-56 fuseforward). */
int eas_conv_fwd_act(const float* x, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi,
                     int Wi, int ksize, int stride, int x_terms, int act, int* inexact_flag, eas_stream_t stream);
/* Fused conv -> BN statistics (the conv -> BN -> LIF step of network_blocks.py:52-53 without a statistics pass over y): eas_conv_fwd
 * without bias whose epilogue also sums its output tile per channel -- stats[Cout][nb][2] doubles, (sum, sum of squares) of each of the
 * nb pixel blocks of the launch; nb = eas_conv_fwd_stats_blocks(same geometry) (0 = no tile).  Hand the buffer to the BatchNorm kernel
 * behind it as EasBnPending.partial with chunks = pitch = nb (a channel slice: stats + first_channel * nb * 2); it adds the partials in a
 * fixed order, so results are reproducible run to run.  y is bit-identical to eas_conv_fwd's.  (Per lane the tile's <= 5 pixel values are
 * added in fp32, everything above that in double.) */
int eas_conv_fwd_stats(const float* x, const void* packed_w, float* y, int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride,
                       int x_terms, int* inexact_flag, double* stats, int nb, eas_stream_t stream);
int eas_conv_fwd_stats_blocks(int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms);
/* SPIKE PLANES: the storage form of spike tensors between the fused layers.  A spike tensor [NI][C][HW] (NI = T*N images; 0/1 spikes
 * and SEW sums: exact in bf16) is kept as bf16 in blocks of 8 channels, planes[NI][C/8][HW][8] -- the MFMA operand of every consumer
 * (8 consecutive input channels of one pixel) is one 16-byte load, at half the bytes of fp32.  Written by eas_bn_lif_fwd_ex
 * (spikes_planes), read by eas_conv_fwd_planes / eas_conv_wgrad_planes_partial; converted at the borders of the fused path by
 * eas_spike_planes_from_f32 / _to_f32.  C % 8 == 0, HW % 4 == 0, 16-byte aligned.  src_ctot / dst_ctot (0 = C): the C channels are a
 * channel (group) slice of a wider tensor, the pointer at the first of them.  *inexact_flag (may be NULL) is OR-ed with 1 when a value
 * of x is not exact in bf16, i.e. the tensor did not hold spikes / small integers as promised (same flag as eas_conv_fwd's). */
int eas_spike_planes_from_f32(const float* x, int src_ctot, void* planes, int dst_ctot, int64_t NI, int C, int HW, int* inexact_flag,
                              eas_stream_t stream);
int eas_spike_planes_to_f32(const void* planes, int src_ctot, float* x, int dst_ctot, int64_t NI, int C, int HW, eas_stream_t stream);
/* eas_conv_fwd / eas_conv_fwd_stats (1x1 and 3x3, stride 1 / 2) reading x as spike planes: the same products summed in the same order as
 * eas_conv_fwd with x_terms = 1 on the fp32 values -- bit-identical y.  stats / nb: NULL / 0, or as eas_conv_fwd_stats.  The geometry
 * queries (eas_conv_fwd_supported, eas_conv_fwd_stats_blocks, eas_conv_wgrad_workspace_floats) take x_terms = 2 for this input form. */
int eas_conv_fwd_planes(const void* x_planes, const void* packed_w, const float* bias, float* y, int NI, int Cin, int Cout, int Hi, int Wi,
                        int ksize, int stride, double* stats, int nb, eas_stream_t stream);
/* 1 when eas_conv_fwd has a tile for this geometry, else 0 (it would return EAS_ERR_UNSUPPORTED).  3x3 layers whose staged input
 * rows do not fit LDS in one piece (real-valued inputs on rows wider than ~280 pixels) run in 2, 4 or 8 column parts. */
int eas_conv_fwd_supported(int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms);
/* ---- Fused eval-mode step: conv -> BatchNorm (running statistics) -> (P)LIF over T in ONE kernel --------------------------------
 * BaseConv.forward of a converted block in eval mode (yolox/models/network_blocks.py:52-53 after utils_snn.py:16-58; the "fused
 * conv -> BN -> LIF step" of the north star, eval-folded).  The convolution output never reaches HBM: the matrix-core kernel's
 * accumulators are the BatchNorm inputs of the lane's neurons for all T steps (the GEMM's pixel index enumerates (time, pixel)), the
 * epilogue normalises with the running statistics and walks the neuron over T, and only spikes are written -- 2 bytes per neuron-step
 * as spike planes instead of the 4 (y written) + 4 (y read) + 2 of eas_conv_fwd + eas_bn_lif_fwd_ex.  Same arithmetic
 * (z = fma(y, gamma*invstd, beta - mean*gamma*invstd), soft reset, fire at >=): spikes bit-identical to the two-kernel path.
 * Neurons: ParametricLIFNode / LIFNode with decay_input = False, v_reset = None (soft reset), the form utils_snn.py:44-53 builds.
 *
 * EasLifRange = one neuron layer = a range of the convolution's output channels: [0, csplit) and [csplit, Cout) (csplit = Cout: one
 * layer; two: conv1 | conv2 of a CSPLayer computed by one convolution, network_blocks.py:175-188).  Per range: the output as spike
 * planes [T*N][out_ctot/8][HW][8] or fp32 [T][N][out_ctot][HW], written at channel out_c0 of the destination (concatenation in place);
 * an optional SEW shortcut (planes or fp32, its channel 0 = the range's first channel) added to the spikes; optional firing rate
 * [N][C_range][HW] (mean over T of the spikes), initial / final membrane potentials [N][C_range][HW] (NULL: reset value / not kept).
 * x: spike planes [T*N][Cin/8][Hi*Wi][8] (x_terms = 2), or, with x_shared = 1, fp32 [N][Cin][Hi][Wi] (x_terms 1 / 3) used for every time
 * step (T identical frames: the first spiking layer behind the stateless stem, spiking_yolox.py:52-57).  csplit, Cout, out_ctot, out_c0,
 * res_ctot multiples of 8; T 3 or 5 for distinct frames (the wave tile holds the T time steps), 1..8 for x_shared. */
typedef struct {
    const float *gamma, *beta, *mean, *invstd;   /* [C_range]: BatchNorm affine parameters, running mean, 1 / sqrt(running var + eps) */
    const float* w_logit;      /* PLIF: k = sigmoid(*w_logit); NULL: k_const */
    float k_const, v_th;
    void* planes;              /* output as spike planes, or */
    float* out_f32;            /* as fp32 */
    int out_ctot, out_c0;
    const void* res_planes;    /* SEW shortcut as planes, or */
    const float* res_f32;      /* as fp32; both NULL: none */
    int res_ctot;
    float* rate;
    const float* v_in;
    float* v_out;
} EasLifRange;
typedef struct {
    const void* x;
    const void* packed_w;      /* eas_conv_pack_weights mode 0 */
    int x_terms, x_shared;
    int T, N, Cin, Cout, Hi, Wi, ksize, stride;
    int csplit;
    EasLifRange range[2];
} EasConvBnLifEval;
int eas_conv_bn_lif_eval(const EasConvBnLifEval* d, eas_stream_t stream);
/* 1 when eas_conv_bn_lif_eval has a tile for this geometry (else the caller runs eas_conv_fwd + eas_bn_lif_fwd_ex) */
int eas_conv_bn_lif_eval_supported(int T, int N, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms, int x_shared);

/* ---- Eval-mode real-valued block: conv -> BatchNorm (running statistics) -> SiLU in ONE kernel ---------------------------------------
 * BaseConv.forward of an unconverted block in eval mode, "act(bn(conv(x)))" (yolox/models/network_blocks.py:52-53): the ANN PAFPN neck and
 * head behind the spiking backbone.  The convolution output never reaches HBM: the matrix-core kernel's epilogue normalises its
 * accumulators with the running statistics (z = fma(y, gamma*invstd, beta - mean*gamma*invstd)), applies the activation and writes the
 * result -- 4 bytes per element instead of the 4 + 4 + 4 of eas_conv_fwd + eas_bn_silu_fwd_ex, same arithmetic: bit-identical outputs.
 * Unlike fuse_model's folded weights (eas_conv_fwd_act) nothing is rounded differently from the unfused model, and the block keeps the
 * forms of the unfused model: EasBnActRange = one BatchNorm = a range of the convolution's output channels, [0, csplit) and
 * [csplit, Cout) (conv1 | conv2 of a CSPLayer or the cls | reg tower convolutions computed by one convolution), each written at channel
 * out_c0 of a destination with out_ctot channels (concatenation in place).  x: fp32 [NI][Cin][Hi][Wi], x_terms = 3 (real-valued: what
 * these blocks read; EAS_ERR_UNSUPPORTED for the spike forms 1 / 2, whose blocks are the converted ones of eas_conv_bn_lif_eval).
 * act: 0 none, 1 SiLU.  Cin, csplit, Cout multiples of 8.  Geometries: those of eas_conv_fwd (eas_conv_fwd_supported). */
typedef struct {
    const float *gamma, *beta, *mean, *invstd;   /* [C_range] */
    float* out;                                  /* fp32 [NI][out_ctot][Ho][Wo] */
    int out_ctot, out_c0;
} EasBnActRange;
typedef struct {
    const void* x;
    const void* packed_w;      /* eas_conv_pack_weights mode 0 */
    int x_terms;
    int NI, Cin, Cout, Hi, Wi, ksize, stride;
    int act;
    int csplit;
    EasBnActRange range[2];
} EasConvBnActEval;
int eas_conv_bn_act_eval(const EasConvBnActEval* d, int* inexact_flag, eas_stream_t stream);

/* Input gradient of a stride-2 3x3 convolution: grad_x[NI][Cin][Hi][Wi] from grad_y[NI][Cout][Ho][Wo] and the weights
 * packed with mode 2 (eas_conv_pack_weights), by parity class of the input pixel (1/2/2/4 taps per class). */
int eas_conv_dgrad_s2(const float* grad_y, const void* packed_w, float* grad_x, int NI, int Cin, int Cout, int Hi, int Wi,
                      eas_stream_t stream);

/* ABI 8.  Input gradient of a 3x3 stride-1 convolution (padding 1) with at most 8 input channels -- the stem of CSPDarknet
 * (yolox/models/darknet.py: BaseConv(in_channels = 8 sampler channels, 32) ahead of dark2), whose gradient flows on into the event sampler:
 * grad_x[NI][Cin][H][W] from grad_y[NI][Cout][H][W] (Cout <= 64) and the fp32 weights [Cout][Cin][3][3] themselves (no packed form).
 * Replaces eas_conv_fwd on grad_y with mode-1 weights there: the nine taps are stacked along the M dimension of the matrix-core tiles
 * (72 of 96 rows used instead of 8 of 32) and shifted afterwards on the fp32 results (csrc/conv_small_dgrad.hip).  Same exact bf16-term
 * products; fixed summation order.  eas_conv_dgrad_small_supported: 1 when the geometry is taken. */
int eas_conv_dgrad_small_supported(int NI, int Cin, int Cout, int H, int W);
int eas_conv_dgrad_small(const float* grad_y, const float* w, float* grad_x, int NI, int Cin, int Cout, int H, int W, eas_stream_t stream);

/* grad_w[Cout][Cin][3][3] of a 3x3 convolution (padding 1, stride 1 or 2) from x[NI][Cin][Hi][Wi] and
 * grad_y[NI][Cout][Ho][Wo] (ATen convolution_backward, weight part).  Reduction over output pixels on the matrix
 * cores (csrc/conv_wgrad_mfma.hip): grad_y as three exact bf16 terms, x as one (x_terms = 1, spikes / small integers)
 * or three; per-block partial sums are reduced in fixed order through `workspace`
 * (eas_conv_wgrad_workspace_floats(...) floats) -- deterministic. */
int64_t eas_conv_wgrad_workspace_floats(int NI, int Cin, int Cout, int Hi, int Wi, int ksize, int stride, int x_terms);
/* number of column parts per row eas_conv_wgrad uses for a 3x3 layer: 1 = whole rows fit one reduction tile, 2..8 = column parts
 * (same kernel, one launch), 0 = unsupported. */
int eas_conv_wgrad_parts(int NI, int Cin, int Cout, int Hi, int Wi, int stride, int x_terms);
int eas_conv_wgrad(const float* x, const float* grad_y, float* grad_w, float* workspace, int NI, int Cin, int Cout, int Hi,
                   int Wi, int ksize, int stride, int x_terms, eas_stream_t stream);
/* Deferred slab reduction.  eas_conv_wgrad = slab kernel + a fixed-order reduction of the slabs; a training step has ~80 weight
 * gradients and nothing reads them before the optimizer step, so the host may launch only the slab kernels
 * (eas_conv_wgrad_partial / eas_conv_wgrad_planes_partial: same arguments without grad_w, return the number of slabs written, > 0, or a
 * negative status) and reduce ALL of them with one launch: eas_conv_wgrad_reduce_many(jobs on the HOST, njobs).  Same summation
 * order as eas_conv_wgrad: bit-identical gradients. */
typedef struct {
    const float* slabs;   /* the workspace a *_partial call filled */
    float* grad_w;        /* n floats */
    int n;                /* Cout*Cin*ksize*ksize */
    int slabs_count;      /* return value of the *_partial call */
} EasWgradReduceJob;
int eas_conv_wgrad_partial(const float* x, const float* grad_y, float* workspace, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                           int stride, int x_terms, eas_stream_t stream);
/* x given as spike planes (see eas_conv_fwd_planes); workspace: eas_conv_wgrad_workspace_floats(..., x_terms = 2) */
int eas_conv_wgrad_planes_partial(const void* x_planes, const float* grad_y, float* workspace, int NI, int Cin, int Cout, int Hi, int Wi, int ksize,
                                  int stride, eas_stream_t stream);
int eas_conv_wgrad_reduce_many(const EasWgradReduceJob* jobs, int njobs, eas_stream_t stream);

/* SPP pooling block fused: out[N][4C][H][W] = cat[x, maxpool_k0(x), maxpool_k1(x), maxpool_k2(x)] (stride 1, padding k/2,
 * odd k; ATen tie rule: first maximum in row-major order) and its backward (arg-max recomputed from x; deterministic gather).
 * Replaces SPPBottleneck.forward's three MaxPool2d + torch.cat (yolox/models/network_blocks.py:143-147).  H*W <= 1024. */
int eas_spp_pool_fwd(const float* x, float* out, int64_t N, int C, int H, int W, int k0, int k1, int k2, eas_stream_t stream);
int eas_spp_pool_bwd(const float* x, const float* grad_out, float* grad_x, int64_t N, int C, int H, int W, int k0, int k1, int k2,
                     eas_stream_t stream);
/* The same block on SPIKE PLANES (what the spiking dark5 hands to it): x_planes [N][C/8][HW][8] -> out_planes [N][4C/8][HW][8] (channel groups
 * [0, C/8) = x, then the three pooled copies; packed 16-bit maxima: spikes / small integers are non-negative, so value order = bit order),
 * and eas_spp_pool_bwd with x read from its planes.  C % 8 == 0. */
int eas_spp_pool_planes_fwd(const void* x_planes, void* out_planes, int64_t N, int C, int H, int W, int k0, int k1, int k2, eas_stream_t stream);
int eas_spp_pool_planes_bwd(const void* x_planes, const float* grad_out, float* grad_x, int64_t N, int C, int H, int W, int k0, int k1, int k2,
                            eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Spike-count / synaptic-operation statistics.  Replaces calc_layer_sop of
 * energy_estimation (yolox/evaluators/event_evaluator.py:473-487, inputs gathered by RecordHook
 * yolox/utils/hooks.py:31-44): for the input x [NI][Cin][H][W] of a ksize x ksize convolution
 * (padding (ksize-1)/2, stride, Cout outputs, groups 1)
 *     out[0] = sum x   (spike count)      out[1] = conv(x, all-ones weights).sum()   (accumulate operations)
 * workspace: eas_spike_sop_workspace_doubles() doubles.  Deterministic (fixed-order reduction). */
int64_t eas_spike_sop_workspace_doubles(void);
int eas_spike_sop(const float* x, int64_t NI, int Cin, int H, int W, int ksize, int stride, int Cout, double* out,
                  double* workspace, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Detection post-processing.  Replaces ``postprocess`` (yolox/utils/boxes.py:33-77; called by the
 * evaluators, e.g. psee_evaluator.py:212) including torchvision.ops.nms / batched_nms (torchvision 0.16.1, restated).
 * pred [B][A][5+ncls] = decoded head output (cx, cy, w, h, obj, class scores), NOT modified (the reference overwrites the
 * first four columns with the corners in place).  out [B][A][7]: rows (x1, y1, x2, y2, obj_conf, class_conf, class_pred) of
 * the kept detections of image b in descending score order, out_count[b] of them.  A <= 16384.
 * workspace: eas_postprocess_workspace_bytes(B, A) bytes. */
int64_t eas_postprocess_workspace_bytes(int B, int A);
int eas_postprocess(const float* pred, int B, int A, int ncls, float conf_thre, float nms_thre, int class_agnostic, float* out,
                    int* out_count, void* workspace, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SimOTA label assignment of the detection loss for the whole batch in one launch (no gradient flows through it):
 * get_assignments + dynamic_k_matching, yolox/models/spiking_yolo_head.py:401-560 (inherited YOLOXHead logic), looping over
 * each image's VALID ground-truth rows only.  grids [A][2], strides [A]; gt_boxes [B][G][4] (cx, cy, w, h), gt_cls [B][G]
 * (class ids as floats), gt_valid [B][G] (0/1 bytes); bbox [B][A][4] decoded boxes, obj [B][A] and cls [B][A][nc] raw logits.
 * Out: fg [B][A] (0/1), matched [B][A] (int64 ground-truth row, 0 where none), matched_iou [B][A].  G <= 255, A <= 12288 (above 4096 anchors -- the 640x640 canvas has 8400 -- part of the per-anchor state is parked in the output arrays). */
int eas_simota_assign(const float* grids, const float* strides, const float* gt_boxes, const float* gt_cls,
                      const unsigned char* gt_valid, const float* bbox, const float* obj, const float* cls, int B, int G, int A,
                      int nc, unsigned char* fg, long long* matched, float* matched_iou, eas_stream_t stream);

/* eas_simota_assign on decoded rows [B][A][5+nc] (cx, cy, w, h, obj logit, class logits) as eas_det_decode writes them. */
int eas_simota_assign_rows(const float* grids, const float* strides, const float* gt_boxes, const float* gt_cls,
                           const unsigned char* gt_valid, const float* dec, int B, int G, int A, int nc, unsigned char* fg,
                           long long* matched, float* matched_iou, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Detection loss after the assignment, forward and gradient in one pass: the loss terms of YOLOXHead.get_losses
 * (yolox/models/yolo_head.py:296-420 as inherited by spiking_yolo_head.py; IOUloss yolox/models/losses.py:10-53 with
 * loss_type 'iou').  reg / obj / cls: HOST arrays of L <= 4 device pointers to the raw head maps [B][4|1|nc][H_l][W_l];
 * hw: host array [L][2]; strides: host array [L].
 * eas_det_decode: decoded rows dec [B][A][5+nc], A = sum H_l*W_l (the cat / permute / (o+grid)*stride / exp(o)*stride chain).
 * eas_det_loss: out[0..5] = total, 5*iou, obj, cls, l1, num_fg/num_gts and out[6] = 1/num_fg; g_reg / g_obj / g_cls (host
 *   arrays of device pointers, shapes of the raw maps, every element written) receive d(total * num_fg)/d(raw map): the caller
 *   multiplies them by grad_total * out[6].  num_gts: device float scalar.  workspace: eas_det_loss_workspace_doubles(). */
/* label preparation of the loss as one launch: labels [B][G][5] (class, cx, cy, w, h; all-zero rows = padding) -> gt_valid [B][G] (row index <
 * number of rows whose values sum to > 0), gt_cls [B][G], gt_boxes [B][G][4], num_gts (device float).  B <= 1024. */
int eas_det_labels(const float* labels, int B, int G, uint8_t* gt_valid, float* gt_cls, float* gt_boxes, float* num_gts, eas_stream_t stream);
int eas_det_decode(int L, const float* const* reg, const float* const* obj, const float* const* cls, const int* hw,
                   const float* strides, int B, int nc, float* dec, eas_stream_t stream);
/* ABI 8.  eas_det_decode with sigmoid on the objectness and class columns: the INFERENCE output of YOLOXHead.forward (yolo_head.py:187-199:
 * per level cat[reg, obj.sigmoid(), cls.sigmoid()], levels concatenated along the anchors) after decode_outputs (:201-216) in one launch. */
int eas_det_decode_eval(int L, const float* const* reg, const float* const* obj, const float* const* cls, const int* hw,
                        const float* strides, int B, int nc, float* dec, eas_stream_t stream);
int64_t eas_det_loss_workspace_doubles(void);
int eas_det_loss(int L, const float* const* reg, const float* const* obj, const float* const* cls, float* const* g_reg,
                 float* const* g_obj, float* const* g_cls, const int* hw, const float* strides, int B, int nc, const float* dec,
                 const float* gt_boxes, const float* gt_cls, int G, const unsigned char* fg, const long long* matched,
                 const float* matched_iou, const float* num_gts, int use_l1, float* out, double* workspace,
                 eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Neck glue: out [M][Ca+Cb][H*up][W*up] = cat[upsample_nearest(a [M][Ca][H][W], x up), b [M][Cb][H*up][W*up]] along channels,
 * up = 2 (nn.Upsample(scale_factor=2) + torch.cat, yolox/models/yolo_pafpn.py:101-113) or up = 1 (the plain two-tensor
 * concatenations of the bottom-up path, :115-121).  Backward: grad_a = sum over each up x up cell (row-major order, as
 * upsample_nearest2d_backward), grad_b = contiguous copy, one kernel.  (W * up) % 4 == 0, W % 2 == 0. */
int eas_upcat_fwd(const float* a, const float* b, float* out, int64_t M, int Ca, int Cb, int H, int W, int up, eas_stream_t stream);
int eas_upcat_bwd(const float* grad_out, float* grad_a, float* grad_b, int64_t M, int Ca, int Cb, int H, int W, int up,
                  eas_stream_t stream);
/* The same on spike planes (see "SPIKE PLANES"): a [M][Ca/8][H*W][8], b [M][Cb/8][(H*up)*(W*up)][8] -> out [M][(Ca+Cb)/8][(H*up)*(W*up)][8], bf16;
 * Ca, Cb multiples of 8.  The neck of a converted YOLOPAFPN (yolox/models/yolo_pafpn.py:101-121) between two fused layers; backward =
 * eas_upcat_bwd on the fp32 gradient. */
int eas_upcat_planes_fwd(const void* a_planes, const void* b_planes, void* out_planes, int64_t M, int Ca, int Cb, int H, int W, int up,
                         eas_stream_t stream);

/* Focus, space to depth (yolox/models/network_blocks.py:198-213: four strided slices + torch.cat): x [M][C][2*Ho][2*Wo] ->
 * out [M][4*C][Ho][Wo], out[m][k*C+c][h][w] = x[m][c][2h+dy_k][2w+dx_k], (dy,dx) = (0,0),(1,0),(0,1),(1,1); inverse = 1 is the
 * inverse permutation (the backward).  Wo % 2 == 0. */
int eas_focus(const float* src, float* dst, int64_t M, int C, int Ho, int Wo, int inverse, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Grouped (multi-problem) launches (ABI 7).  The detection head runs the same layers on three pyramid levels that are independent
 * until the loss (yolox/models/yolo_head.py:149-200: `for k, (cls_conv, reg_conv, stride_this_level, x) in enumerate(zip(...))`),
 * and the 16x20 / 8x10 levels cannot fill 256 CUs by themselves.  A grouped call takes a table of problems -- each its own geometry,
 * weights and pointers -- and runs them as ONE grid: blockIdx -> (problem, tile).  Per problem the arithmetic and summation order are
 * those of the single-problem entry point (eas_conv_fwd / eas_bn_silu_fwd_ex / eas_bn_silu_bwd / eas_conv_wgrad_partial /
 * eas_channel_sum): outputs are bit-identical for the same tile plan; the BatchNorm partial sums and the weight-gradient slabs are cut
 * by the GROUP's tile / slice plan (deterministic, fixed order).
 * EAS_ERR_UNSUPPORTED from a *_plan call: run the problems one by one.  Stride 1, padding ksize / 2, fp32 tensors, x_terms = 3. */
typedef struct {
    const void* x;          /* [NI][Cin][Hi][Wi] fp32 (the input gradient: grad_y, Cin / Cout swapped, weights packed with mode 1) */
    const void* packed_w;   /* eas_conv_pack_weights */
    const float* bias;      /* nullable [Cout] */
    float* y;               /* [NI][Cout][Hi][Wi] */
    double* stats;          /* nullable: per-channel partial sums of y as eas_conv_fwd_stats leaves them, [Cout][nb][2] doubles with
                               nb = eas_conv_fwd_group_plan's nb_out[problem] */
    int NI, Cin, Cout, Hi, Wi;
    int accumulate;         /* ksize 1 only: y += conv(x) -- the second reader of a tensor adds its input gradient in place */
} EasConvProblem;
int eas_conv_fwd_group_plan(const EasConvProblem* problems, int n, int ksize, int x_terms, int* nb_out);
int eas_conv_fwd_group(const EasConvProblem* problems, int n, int ksize, int x_terms, eas_stream_t stream);

/* eas_conv_bn_act_eval (above) for several layers in one launch: the eval-mode head's stage over its pyramid levels.  Same descriptors, same
 * arithmetic per problem (bit-identical outputs); all problems ksize 1 or all ksize 3, stride 1, x_terms 3.  n <= 8. */
int eas_conv_bn_act_eval_group(const EasConvBnActEval* problems, int n, eas_stream_t stream);

/* eas_bn_silu_fwd_ex / eas_bn_silu_bwd for several layers in one launch (two launches for the backward: sums, apply).
 * workspace: eas_bn_workspace_doubles(C) doubles per problem. */
typedef struct {
    const float* y;
    float* mean;            /* outputs when pending.partial != NULL (batch statistics), inputs otherwise */
    float* invstd;
    const float* gamma;
    const float* beta;
    float* out;
    int N, C, HW, out_ctot, y_ctot;
    EasBnPending pending;
} EasBnSiluFwdProblem;
int eas_bn_silu_fwd_group(const EasBnSiluFwdProblem* problems, int n, eas_stream_t stream);
typedef struct {
    const float* grad_out;
    const float* y;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    float* grad_y;
    float* grad_gamma;
    float* grad_beta;
    double* workspace;
    int batch_stats, N, C, HW, grad_out_ctot, y_ctot;
} EasBnSiluBwdProblem;
int eas_bn_silu_bwd_group(const EasBnSiluBwdProblem* problems, int n, eas_stream_t stream);

/* eas_conv_wgrad_partial for several layers in one launch: the pixel slices of every problem are sized for the group's total block count
 * (a layer alone needs hundreds of slabs to give 256 CUs a block each; in a group a few do).  plan: slabs_out[p] = slabs problem p
 * writes (workspace = slabs * Cout * Cin * ksize^2 floats), also the count to hand to eas_conv_wgrad_reduce_many. */
typedef struct {
    const void* x;          /* [NI][Cin][Hi][Wi] fp32 */
    const float* grad_y;    /* [NI][Cout][Hi][Wi] */
    float* workspace;
    int NI, Cin, Cout, Hi, Wi;
} EasWgradProblem;
int eas_conv_wgrad_group_plan(const EasWgradProblem* problems, int n, int ksize, int x_terms, int* slabs_out);
int eas_conv_wgrad_group_partial(const EasWgradProblem* problems, int n, int ksize, int x_terms, eas_stream_t stream);

/* eas_channel_sum for several tensors in one launch (the bias gradients of the prediction convolutions of all levels) */
typedef struct {
    const float* g;
    float* out;
    int N, C, HW;
} EasChannelSumProblem;
int eas_channel_sum_group(const EasChannelSumProblem* problems, int n, eas_stream_t stream);

/* ABI 8.  Input gradients of 1x1 convolutions with very few OUTPUT channels -- the head's prediction convolutions (yolo_head.py:161-163 of
 * the reference: cls_preds / reg_preds / obj_preds, num_classes / 4 / 1 channels) -- for several inputs in one launch (n <= 8):
 *   gx[n][c][p] = sum_k w_a[k][c] * gy_a[n][k][p]  (+ sum_k w_b[k][c] * gy_b[n][k][p] for a second convolution reading the same input:
 *   obj_preds next to reg_preds; Kb = 0: none),  Ka + Kb <= 8, HW % 4 == 0, tensors 16-byte aligned.
 * w_a / w_b are the fp32 weights of the convolutions ([K][C][1][1]).  Plain fp32 FMAs in the fixed order k = 0 .. (first reader, then second). */
typedef struct {
    const float* gy_a;
    const float* w_a;
    int Ka;
    const float* gy_b;
    const float* w_b;
    int Kb;
    float* gx;
    int N, C, HW;
} EasPredDgradProblem;
int eas_pred_dgrad_group(const EasPredDgradProblem* problems, int n, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ABI 8.  Adam step of every parameter group in one launch (csrc/adam.hip): torch.optim.Adam as the reference builds it
 * (yolox/exp/event_yolox_base.py:352-414: BN weights | convolution weights with weight decay | biases | neuron parameters | sampler parameters),
 * update rule and float / double promotions of torch's fused kernel (ADAM_MODE::ORIGINAL, no amsgrad, no grad scaler).
 * table: DEVICE array of ntensors entries of eas_adam_table_entry_bytes() = 96 bytes (ABI 9; 80 in ABI 8):
 *   { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; float* step (fp32 device scalar); const float* lr_ptr (device scalar or NULL);
 *     float* ema (ABI 9: the tensor's twin in the averaged model, or NULL); double lr (used when lr_ptr is NULL and group < 0); double weight_decay;
 *     int64 numel; int64 first_block; int32 group (ABI 9: >= 0 selects EasAdamHyper.group_lr[group]; -1: lr above); int32 pad }
 * first_block = prefix sum of ceil(numel / eas_adam_chunk()) over the entries, total_blocks = the sum.  eas_adam_step takes step number
 * *step + 1 on every tensor WITHOUT writing the counters; eas_adam_advance_steps (called right after it) adds 1 to each.
 * An entry with grad == NULL takes no Adam step (exp_avg / exp_avg_sq / step unused): a tensor only the average follows. */
int eas_adam_table_entry_bytes(void);
int eas_adam_chunk(void);
int eas_adam_step(const void* table, int ntensors, long long total_blocks, double beta1, double beta2, double eps, eas_stream_t stream);
int eas_adam_advance_steps(const void* table, int ntensors, eas_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ABI 9.  The weight average of the training loop inside the optimizer launch, and learning rates that change every step without a new table.
 * Replaces ModelEMA.update (yolox/utils/ema.py:44-60), which the reference calls right after optimizer.step() every iteration
 * (yolox/core/trainer.py:120-121; ema = True is the default of yolox/exp/event_yolox_base.py:116):
 *     d = ema_decay * (1 - exp(-n / ema_ramp))  (double; n = *ema_updates + 1, the update being taken; the reference: 0.9998, 2000)
 *     ema = ema * (float)d;  ema = ema + (float)(1 - d) * param          (param = the value AFTER this step's Adam update)
 * for every entry whose `ema` pointer is set -- parameters in the same pass that writes them, BatchNorm running statistics (float buffers
 * of the state dict) as entries with grad == NULL.  ema_updates is a DEVICE double that eas_adam_advance_steps_ex increments after the step, so
 * a step captured into a HIP graph keeps the reference's ramp going; ema_updates == NULL: no entry may carry an `ema` pointer.
 * group_lr: learning rates of up to EAS_ADAM_MAX_GROUPS parameter groups as launch arguments (an eager trainer whose scheduler hands out a new
 * python float every iteration re-uses its table; a captured trainer uses lr_ptr device scalars instead). */
#define EAS_ADAM_MAX_GROUPS 16
typedef struct {
    double beta1, beta2, eps;
    double group_lr[EAS_ADAM_MAX_GROUPS];
    const double* ema_updates;
    double ema_decay;
    double ema_ramp;
} EasAdamHyper;
int eas_adam_step_ex(const void* table, int ntensors, long long total_blocks, const EasAdamHyper* hyper, eas_stream_t stream);
int eas_adam_advance_steps_ex(const void* table, int ntensors, double* ema_updates, eas_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EAS_HIP_H */
