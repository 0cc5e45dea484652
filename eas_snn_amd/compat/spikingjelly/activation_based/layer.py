"""Multi-step containers (spikingjelly ``layer.SeqToANNContainer`` / ``layer.BatchNorm2d``), call sites
yolox/utils/utils_snn.py:24-32.  ``BatchNorm2d`` stays a subclass of ``nn.BatchNorm2d`` (isinstance checks at
yolox/exp/event_yolox_base.py:181,366,392 and yolox/utils/allreduce_norm.py:14-21 rely on it)."""
import torch
import torch.nn as nn

from eas_snn_amd import ops

from . import base, functional


class SeqToANNContainer(nn.Sequential, base.StepModule):
    """Folds [T, N] into the batch axis around stateless modules (child keys '0', '1', ... as upstream)."""

    def __init__(self, *args):
        super().__init__(*args)
        self._step_mode = 'm'

    def supported_step_mode(self):
        return ('m',)

    def _run_inner(self_, x):
        """The wrapped stateless module(s) on a [N,C,H,W] batch; a lone Conv2d goes to the matrix-core kernels."""
        if len(self_) == 1 and type(self_[0]) is nn.Conv2d and x.is_cuda:
            return ops.conv2d(x, self_[0])
        return nn.Sequential.forward(self_, ops.dense(x))

    def forward(self, x_seq):
        base = getattr(x_seq, '_eas_base', None)
        if base is not None:
            # x_seq is T identical frames (stride-0 broadcast of ``base``, spiking_yolox.py:52-57): a stateless module
            # gives T identical results, so it runs once and the result is broadcast again (bit-identical, 1/T the work)
            with ops.replicated(x_seq.shape[0]):      # BN inside sees N samples that stand for T*N
                y0 = self._run_inner(base)
            y = y0.unsqueeze(0).expand(x_seq.shape[0], *y0.shape)
            y._eas_base = y0
            return y
        small = ops.is_small_int(x_seq)
        x = ops.fold_time(x_seq)            # [T*N, ...] with the spike tags (small-integer mark, spike planes) carried along
        y = self._run_inner(x)
        y = y.view(x_seq.shape[0], x_seq.shape[1], *y.shape[1:])
        if small and len(self) == 1 and isinstance(self[0], (nn.MaxPool2d, nn.Upsample)):
            ops.mark_small_int(y)           # pooling / nearest upsampling of spikes are spikes
        return y


class BatchNorm2d(nn.BatchNorm2d, base.StepModule):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, step_mode='s'):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.step_mode = step_mode

    def extra_repr(self):
        return super().extra_repr() + f', step_mode={self.step_mode}'

    def forward(self, x):
        if self.step_mode == 's':
            if x.dim() != 4:
                raise ValueError(f'expected x with shape [N, C, H, W], but got x with shape {x.shape}!')
            return super().forward(x)
        if x.dim() != 5:
            raise ValueError(f'expected x with shape [T, N, C, H, W], but got x with shape {x.shape}!')
        return functional.seq_to_ann_forward(x, super().forward)   # statistics over T*N*H*W per channel

    # ---- fused path: BN + following LIF neuron in one HIP kernel pair (eas_bn_lif_*)
    def _use_batch_stats(self):
        return self.training or (self.running_mean is None and self.running_var is None)

    def can_fuse(self, y_seq, node=None):
        # (a forward hook on this module or on the neuron wants the module CALLED: the fused kernel pair would skip both)
        watched = self._forward_hooks or self._forward_pre_hooks or (node is not None and (node._forward_hooks or node._forward_pre_hooks))
        return bool(not watched and self.affine and self.step_mode == 'm' and ops.bn_lif_supported(y_seq, y_seq.shape[0])
                    and (self.momentum is not None or not self.training))

    def fused_with(self, node, y_seq, want_mean=False, residual=None, cat=None, planes=False):
        """spikes = node(self(y_seq)) for y_seq [T,N,C,H,W] without materialising the BN output.
        residual: the result is spikes + residual (SEW shortcut) from the same kernel; cat = (buffer [T,N,Ctot,H,W], first
        channel[, spike planes of the buffer]): the result is written into that channel range of the buffer and returned as
        a view; planes: the result is written as bf16 spike planes and the returned tensor is the ghost that carries them
        (eas_snn_amd.ops "SPIKE PLANES")."""
        if not self.can_fuse(y_seq, node):
            if cat is not None:
                raise RuntimeError('in-place concatenation needs the fused BN+LIF path (callers check network_blocks._fusable)')
            y_seq = y_seq.contiguous()
            out = node(self(y_seq))
            if residual is not None:
                out = out + ops.dense(residual)
            return (out, ops.time_mean(out)) if want_mean else out
        batch = self._use_batch_stats()
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            ops.bump_counter(self.num_batches_tracked)
        a = node.lif_args()
        update = batch and self.training and self.track_running_stats
        base = getattr(y_seq, '_eas_base', None)       # conv output shared by all T steps (identical input frames)
        spikes, v_out, mean = ops.bn_lif_multistep(
            y_seq if base is None else base, self.weight, self.bias, self.running_mean if (update or not batch) else None,
            self.running_var if (update or not batch) else None, batch, self.momentum if update else None, self.eps,
            node._v_in(y_seq[0]), a['w'], a['k_const'], a['v_th'], a['v_reset'], a['flags'], a['surrogate'], a['alpha'],
            want_mean=want_mean, t_bcast=0 if base is None else y_seq.shape[0], residual=residual, cat=cat,
            planes=planes)
        if v_out is not None:
            node.v = v_out
        if residual is None or ops.is_small_int(residual):
            ops.mark_small_int(spikes)      # the neuron's output is 0/1 by construction (plus a spike-count residual)
        return (spikes, mean) if want_mean else spikes


class Conv2d(nn.Conv2d, base.StepModule):
    def __init__(self, *args, step_mode='s', **kwargs):
        super().__init__(*args, **kwargs)
        self.step_mode = step_mode

    def forward(self, x):
        if self.step_mode == 's':
            return super().forward(x)
        return functional.seq_to_ann_forward(x, super().forward)


def fused_pair(bn_a, node_a, bn_b, node_b, y12, cat_a=None, cat_b=None, planes_a=False, planes_b=False):
    """spikes of node_a(bn_a(y12[:, :, :Ca])) and node_b(bn_b(y12[:, :, Ca:])) from ONE convolution output y12 [T,N,Ca+Cb,H,W]
    (ops.bn_lif_pair): the two 1x1 branches of a CSPLayer.  cat_a / cat_b = (buffer, first channel[, its planes]) and planes_a / planes_b
    as in ``fused_with``."""
    packs = []
    for bn, node, cat, planes in ((bn_a, node_a, cat_a, planes_a), (bn_b, node_b, cat_b, planes_b)):
        batch = bn._use_batch_stats()
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            ops.bump_counter(bn.num_batches_tracked)
        update = batch and bn.training and bn.track_running_stats
        a = node.lif_args()
        state = (bn.running_mean if (update or not batch) else None, bn.running_var if (update or not batch) else None, bool(batch),
                 None if not update or bn.momentum is None else float(bn.momentum), float(bn.eps))
        cfg = (state, node._v_in(y12[0, :, :bn.num_features]), float(a['k_const']), float(a['v_th']), float(a['v_reset']), int(a['flags']),
               ops.SURROGATE_IDS[a['surrogate']] if isinstance(a['surrogate'], str) else int(a['surrogate']), float(a['alpha']),
               bool(ops.state_writeback()), cat, int(bn.num_features), bool(planes) and bn.num_features % 8 == 0)
        packs.append((bn.weight, bn.bias, a['w'], cfg))
    sa, va, sb, vb = ops.bn_lif_pair(y12, packs[0], packs[1])
    if va is not None:
        node_a.v = va
    if vb is not None:
        node_b.v = vb
    ops.mark_small_int(sa)
    ops.mark_small_int(sb)
    return sa, sb
