"""Conv/BN/activation topology of (spiking) YOLOX.  Mirrors the module and attribute names of the reference
(yolox/models/network_blocks.py:31-213) so ``state_dict`` keys are identical and published checkpoints load.

What is different: once ``convert_to_spiking`` has swapped ``bn``/``act`` for the multi-step BatchNorm and a
spiking neuron, ``BaseConv.forward`` runs conv -> [BN + LIF fused in one HIP kernel pair] instead of three
separate module calls."""
import os

import torch
import torch.nn as nn

from eas_snn_amd import ops
from spikingjelly.activation_based import layer as sj_layer
from spikingjelly.activation_based import neuron as sj_neuron


def get_activation(name='silu', inplace=True):
    table = {'silu': lambda: nn.SiLU(inplace=inplace), 'relu': lambda: nn.ReLU(inplace=inplace),
             'lrelu': lambda: nn.LeakyReLU(0.1, inplace=inplace), 'idnt': nn.Identity}
    if name not in table:
        raise AttributeError('Unsupported act type: {}'.format(name))
    return table[name]()


def _cat(parts):
    """Channel concatenation; a concatenation of spike tensors is still a spike tensor."""
    small = all(ops.is_small_int(p) for p in parts)
    parts = [ops.dense(p) for p in parts]
    out = torch.cat(parts, dim=-3)
    if small:
        ops.mark_small_int(out)
    return out


def _pool_ksize(m):
    """kernel size of a (possibly SeqToANNContainer-wrapped) stride-1 'same' MaxPool2d, else None."""
    if isinstance(m, sj_layer.SeqToANNContainer) and len(m) == 1:
        m = m[0]
    if type(m) is nn.MaxPool2d and m.stride in (1, (1, 1)) and m.dilation in (1, (1, 1)) and not m.ceil_mode:
        k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
        p = m.padding if isinstance(m.padding, int) else m.padding[0]
        if p == k // 2:
            return k
    return None


def _fusable(block, x):
    """a converted BaseConv (1x1 or 3x3, stride 1) whose BN+LIF runs on the fused HIP kernels for a 5-D CUDA spike tensor"""
    return (isinstance(block, BaseConv) and block.spiking() and x.dim() == 5 and x.is_cuda and not block.emit_rate
            and not (block.bn._forward_hooks or block.bn._forward_pre_hooks or block.act._forward_hooks or block.act._forward_pre_hooks)
            and block.bn.affine and block.bn.step_mode == 'm' and (block.bn.momentum is not None or not block.bn.training)
            and (x.shape[-1] * x.shape[-2]) % 4 == 0 and x.shape[0] <= 8 and _stride1(block.conv))


def _stride1(conv):
    c = conv[0] if isinstance(conv, nn.Sequential) and len(conv) == 1 else conv
    return isinstance(c, nn.Conv2d) and c.stride in (1, (1, 1))


class BaseConv(nn.Module):
    """Conv2d -> BatchNorm -> activation (spiking: SeqToANNContainer(Conv2d) -> BN('m') -> PLIF)."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act='silu'):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride, padding=(ksize - 1) // 2,
                              groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = get_activation(act, inplace=True)
        self.emit_rate = False     # spiking only: also return the firing rate (mean over T) from the fused kernel
        self.planes_out = False    # spiking only: the fused BN+LIF kernel writes its spikes as bf16 spike planes (ops "SPIKE PLANES"); set
                                   # by ``enable_spike_planes`` on blocks whose readers take planes or unpack them (ops.dense)

    def spiking(self):
        return isinstance(self.act, sj_neuron.BaseNode) and isinstance(getattr(self, 'bn', None), sj_layer.BatchNorm2d)

    def forward(self, x, residual=None, cat=None):
        """residual / cat: only for converted (spiking) blocks on the fused BN+LIF path -- the SEW shortcut addition and the
        channel concatenation of the caller happen inside the BN+LIF kernel (see Bottleneck / CSPLayer); cat = (buffer, first channel[,
        spike planes of the buffer])."""
        if ops._TIMER is not None:
            ops.set_tag(getattr(self, '_eas_name', None))            # development: per-layer timing (scripts/layer_times.py)
        if self.spiking():
            if not self.training and ops.fused_eval_ok(x, self.conv, [(self.bn, self.act)]):
                # eval mode: conv -> BN (running statistics) -> LIF over T as ONE kernel, the convolution output never reaches HBM
                return ops.conv_bn_lif_eval(x, self.conv, [(self.bn, self.act)], want_mean=self.emit_rate, residual=residual, cats=[cat],
                                            planes=[self.wants_planes()])
            with ops.conv_stats_scope(self.bn._use_batch_stats()):      # the convolution sums its output for the BN behind it
                y = self.conv(x)
            return self.bn.fused_with(self.act, y, want_mean=self.emit_rate, residual=residual, cat=cat,
                                      planes=self.wants_planes() and x.dim() == 5)
        assert residual is None
        norm = self.eval_norm() if not self.training else None
        if norm is not None and ops.fused_ann_eval_ok(x, self.conv, [norm]):
            # eval mode: conv -> BN (running statistics) -> SiLU as ONE kernel, the convolution output never reaches HBM
            return ops.conv_bn_act_eval(x, self.conv, [norm], cats=[cat])
        with ops.conv_stats_scope(self.bn.training or self.bn.running_mean is None):
            y = ops.conv2d(x, self.conv) if type(self.conv) is nn.Conv2d and x.is_cuda else self.conv(x)
        if self.ann_fusable(y):
            return ops.bn_silu(y, self.bn, cat=cat)     # the statistics from the convolution + one fused normalise/SiLU pass (HIP)
        out = self.act(self.bn(y))
        if cat is not None:
            # (eval mode only, ``ann_block_ok``: a layer of the block that neither fused kernel takes -- a statistics tap, a map whose size
            # the kernels do not cover -- still delivers into the caller's concatenation buffer)
            assert not torch.is_grad_enabled()
            dst = cat[0].narrow(1, int(cat[1]), out.shape[1])
            dst.copy_(out)
            return dst
        return out

    def wants_planes(self):
        return self.planes_out and ops.planes_enabled()

    def eval_norm(self):
        """what follows the convolution of a real-valued conv -> BN -> SiLU block in eval mode: its BatchNorm2d, or -- once ``fuse_model`` has
        folded that into the convolution -- the convolution's bias; None: not such a block (the fused eval kernel does not apply)"""
        if not isinstance(self.act, nn.SiLU) or type(self.conv) is not nn.Conv2d:
            return None
        if self.act._forward_hooks or self.act._forward_pre_hooks:
            return None                 # somebody watches the activation module: it has to be called (the fused kernel would skip it)
        bn = getattr(self, 'bn', None)
        if bn is None:
            return self.conv.bias
        return bn if type(bn) is nn.BatchNorm2d and not bn.training else None

    def ann_block_ok(self, probe):
        """the block can write its output into a channel range of a concatenation buffer: BN + SiLU on the fused HIP kernel (training and
        eval), or -- eval mode, also after ``fuse_model`` -- the whole block as one kernel"""
        if self.ann_fusable(probe):
            return True
        return (not self.training and not torch.is_grad_enabled() and ops.FUSED_ANN_EVAL and self.eval_norm() is not None
                and probe.is_cuda and probe.dtype == torch.float32 and probe.dim() == 4)

    def ann_fusable(self, y):
        """BN + SiLU of this (real-valued) block run as the fused HIP kernel on ``y`` (then the output can also go straight into a
        concatenation buffer)"""
        bn = getattr(self, 'bn', None)          # (None: fuse_model folded it into the convolution)
        return (type(bn) is nn.BatchNorm2d and isinstance(self.act, nn.SiLU) and bn.affine and ops.bn_silu_supported(y)
                and (bn.momentum is not None or not bn.training))

    def fuseforward(self, x, residual=None, cat=None):
        """the forward of a block whose BatchNorm ``fuse_model`` folded into the convolution (network_blocks.py:55-56 of the reference): on the
        GPU without autograd bias and activation run in the convolution's epilogue -- one kernel for the whole block, written straight into
        the caller's concatenation buffer where there is one (cat = (buffer, first channel))"""
        assert residual is None
        out = None
        if type(self.conv) is nn.Conv2d and isinstance(self.act, nn.SiLU) and x.is_cuda and x.dim() == 4 and self.conv.bias is not None:
            if not self.training and ops.fused_ann_eval_ok(x, self.conv, [self.conv.bias]):
                return ops.conv_bn_act_eval(x, self.conv, [self.conv.bias], cats=[cat])
            out = ops.conv_act_eval(x, self.conv, 'silu')
            if out is None:
                out = self.act(ops.conv2d(x, self.conv))
        if out is None:
            out = self.act(self.conv(x))
        if cat is not None:
            dst = cat[0].narrow(1, int(cat[1]), out.shape[1])
            dst.copy_(out)
            return dst
        return out


class DWConv(nn.Module):
    def __init__(self, in_channels, out_channels, ksize, stride=1, act='silu'):
        super().__init__()
        self.dconv = BaseConv(in_channels, in_channels, ksize=ksize, stride=stride, groups=in_channels, act=act)
        self.pconv = BaseConv(in_channels, out_channels, ksize=1, stride=1, groups=1, act=act)

    def forward(self, x):
        return self.pconv(self.dconv(x))


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act='silu'):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = (DWConv if depthwise else BaseConv)(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward(self, x, cat=None):
        """cat = (buffer, first channel): write the block's output into that channel range of a concatenation buffer."""
        h = self.conv1(x)
        if _fusable(self.conv2, x):
            # SEW residual (spike sums 0/1/2..) and the caller's concatenation from the BN+LIF kernel of conv2
            return self.conv2(h, residual=x if self.use_add else None, cat=cat)
        if cat is not None:                      # real-valued block without a shortcut: conv2's BN + SiLU writes into the buffer
            assert not self.use_add
            return self.conv2(h, cat=cat)
        y = self.conv2(h)
        if not self.use_add:
            return y
        small = ops.is_small_int(y) and ops.is_small_int(x)
        out = ops.dense(y) + ops.dense(x)        # SEW residual: spike sums 0/1/2.. when spiking
        if small:
            ops.mark_small_int(out)
        return out


class SPPBottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation='silu'):
        super().__init__()
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = BaseConv(hidden * (len(kernel_sizes) + 1), out_channels, 1, stride=1, act=activation)

    def forward(self, x):
        x = self.conv1(x)
        ks = [_pool_ksize(m) for m in self.m]
        if all(k is not None for k in ks) and ops.spp_pool_supported(x, ks):
            return self.conv2(ops.spp_pool_cat(x, ks))      # three poolings + concatenation in one kernel
        x = ops.dense(x)
        return self.conv2(_cat([x] + [m(x) for m in self.m]))


class CSPLayer(nn.Module):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act='silu'):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def _branch_convs(self):
        """the nn.Conv2d of conv1 and of conv2 (inside their SeqToANNContainer once converted), or None"""
        out = []
        for blk in (self.conv1, self.conv2):
            c = blk.conv[0] if isinstance(blk.conv, nn.Sequential) and len(blk.conv) == 1 else blk.conv
            if type(c) is not nn.Conv2d:
                return None
            out.append(c)
        return out

    def eas_dual_convs(self):
        """conv1 | conv2 read the same input: ``ops.packed_weights`` packs their concatenated weight for ``ops.conv2d_dual``"""
        cs = self._branch_convs()
        if cs is None or cs[0].kernel_size != (1, 1) or cs[1].kernel_size != (1, 1) or (cs[0].bias is None) != (cs[1].bias is None) \
                or cs[0].in_channels != cs[1].in_channels or cs[0].groups != 1 or cs[1].groups != 1:
            return []                   # (biases on both: the pair fuse_model folded, whose eval forward still runs as one convolution)
        return [('c12', cs[0], cs[1])]

    def _dual_ok(self):
        """conv1 / conv2 as one convolution: both plain bias-free 1x1 convolutions without hooks (EAS_NO_DUAL: development switch)"""
        import os
        if os.environ.get('EAS_NO_DUAL'):
            return False
        for blk in (self.conv1, self.conv2):
            c = blk.conv[0] if isinstance(blk.conv, nn.Sequential) and len(blk.conv) == 1 else None
            if (type(c) is not nn.Conv2d or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.bias is not None or c.groups != 1
                    or c._forward_hooks or c._forward_pre_hooks or c.in_channels % 8 != 0 or blk.emit_rate
                    or getattr(blk.act.surrogate_function, 'hip_id', None) == 'patan'):     # learnable slope: per-layer BN+LIF calls
                return False
        return not torch.nn.modules.module._global_forward_hooks

    def forward(self, x):
        last = self.m[-1] if len(self.m) else self.conv1
        tail = last.conv2 if isinstance(last, Bottleneck) else last
        if _fusable(tail, x) and _fusable(self.conv2, x) and not (isinstance(last, Bottleneck) and isinstance(last.conv2, DWConv)):
            # both branches write their spikes straight into the two halves of the concatenation (no torch.cat, and the
            # backward hands each branch its half of the gradient as a view)
            T, N = x.shape[:2]
            h = self.conv2.conv[0].out_channels if isinstance(self.conv2.conv, nn.Sequential) else self.conv2.conv.out_channels
            Ho, Wo = x.shape[-2:]
            if tail.wants_planes() and self.conv2.wants_planes() and h % 8 == 0:
                # the concatenation exists as spike planes only (read by conv3): a ghost stands for it in the autograd graph
                sp_buf = ops.new_planes(T, N, 2 * h, Ho, Wo, x.device)
                buf = ops.ghost((T, N, 2 * h, Ho, Wo), x.device)
            else:
                sp_buf = None
                buf = torch.empty((T, N, 2 * h, Ho, Wo), dtype=torch.float32, device=x.device)
            want_a = (sp_buf is not None) if not len(self.m) else self.conv1.wants_planes()
            if self._dual_ok():
                # conv1 and conv2 read the same x: ONE 1x1 convolution with the concatenated weights, then the two BN+LIF layers on
                # the two channel halves of its output (x read once; the input gradient is one convolution, no branch addition)
                c1, c2 = self.conv1.conv[0], self.conv2.conv[0]
                pair = [(self.conv1.bn, self.conv1.act), (self.conv2.bn, self.conv2.act)]
                if not self.training and ops.fused_eval_ok(x, (c1, c2), pair):
                    # eval mode: the one convolution for both branches AND their two BN + LIF layers in one kernel
                    packs = (getattr(self, '_eas_dual_packs', None) or {}).get('c12')
                    a, b = ops.conv_bn_lif_eval(x, (c1, c2), pair, cats=[None if len(self.m) else (buf, 0, sp_buf), (buf, h, sp_buf)],
                                                planes=[want_a, sp_buf is not None], packs=ops.current_packs(packs))
                    if len(self.m):
                        for blk in self.m[:-1]:
                            a = blk(a)
                        a = self.m[-1](a, cat=(buf, 0, sp_buf))
                    return self.conv3(ops.join_channels(buf, a, b, sp_buf=sp_buf))
                x4 = ops.fold_time(x)
                sink = ops.conv_sink()
                if sink is not None:
                    sink(c1, x4, 1)
                    sink(c2, x4, 1)
                with ops.conv_stats_scope(self.conv1.bn._use_batch_stats() and self.conv2.bn._use_batch_stats()):
                    if ops.conv_dual_ok(x4, c1, c2):        # weights packed from the two parameters, no torch.cat
                        y12 = ops.conv2d_dual(x4, c1, c2, self, 'c12').view(T, N, 2 * h, Ho, Wo)
                    else:
                        y12 = ops.conv2d_weight(x4, torch.cat([c1.weight, c2.weight], 0)).view(T, N, 2 * h, Ho, Wo)
                a, b = sj_layer.fused_pair(self.conv1.bn, self.conv1.act, self.conv2.bn, self.conv2.act, y12,
                                           cat_a=None if len(self.m) else (buf, 0, sp_buf), cat_b=(buf, h, sp_buf),
                                           planes_a=want_a, planes_b=sp_buf is not None)
            else:
                a = self.conv1(x, cat=None if len(self.m) else (buf, 0, sp_buf))
                b = self.conv2(x, cat=(buf, h, sp_buf))
            if len(self.m):
                for blk in self.m[:-1]:
                    a = blk(a)
                a = self.m[-1](a, cat=(buf, 0, sp_buf))
            return self.conv3(ops.join_channels(buf, a, b, sp_buf=sp_buf))
        if self._ann_inplace_cat(x):
            # real-valued CSPLayer (PAFPN neck): both branches write their BN + SiLU output into the two halves of the concatenation
            # (no torch.cat; the backward reads the two halves of the gradient in place)
            N = x.shape[0]
            h = self.conv2.conv.out_channels
            buf = torch.empty((N, 2 * h) + tuple(x.shape[-2:]), dtype=torch.float32, device=x.device)
            cs = self._branch_convs()
            dual = cs is not None and not os.environ.get('EAS_NO_DUAL') and not os.environ.get('EAS_NO_ANN_DUAL')
            norms = [self.conv1.eval_norm(), self.conv2.eval_norm()] if (dual and not self.training) else [None]
            if None not in norms and ops.fused_ann_eval_ok(x, (cs[0], cs[1]), norms):
                # eval mode (also after fuse_model): the one convolution for both branches AND their BN + SiLU in one kernel
                packs = (getattr(self, '_eas_dual_packs', None) or {}).get('c12')
                a, b = ops.conv_bn_act_eval(x, (cs[0], cs[1]), norms, cats=[None if len(self.m) else (buf, 0), (buf, h)],
                                            packs=ops.current_packs(packs))
            elif dual and self.conv1.ann_fusable(x[:, :1]) and ops.conv_dual_ok(x, cs[0], cs[1]):
                # conv1 and conv2 read the same x: ONE 1x1 convolution (weights packed from the two parameters), then the two BN + SiLU
                # layers on the channel halves of its output; the input gradient is one convolution, no addition of two branch gradients
                sink = ops.conv_sink()
                if sink is not None:
                    sink(cs[0], x, 1)
                    sink(cs[1], x, 1)
                with ops.conv_stats_scope(all(bn.training or bn.running_mean is None for bn in (self.conv1.bn, self.conv2.bn))):
                    y12 = ops.conv2d_dual(x, cs[0], cs[1], self, 'c12')
                a, b = ops.bn_silu_pair(y12, self.conv1.bn, self.conv2.bn, cat_a=None if len(self.m) else (buf, 0), cat_b=(buf, h))
            else:
                b = self.conv2(x, cat=(buf, h))
                a = self.conv1(x, cat=None if len(self.m) else (buf, 0))
            if len(self.m):
                for blk in self.m[:-1]:
                    a = blk(a)
                a = self.m[-1](a, cat=(buf, 0))
            return self.conv3(ops.join_channels(buf, a, b))
        return self.conv3(_cat((self.m(self.conv1(x)), self.conv2(x))))

    def _ann_inplace_cat(self, x):
        if os.environ.get('EAS_ANN_INPLACE_CAT', '1') != '1':       # development switch
            return False
        last = self.m[-1] if len(self.m) else self.conv1
        if not (x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and not self.conv1.spiking() and not self.conv2.spiking()
                and type(self.conv2.conv) is nn.Conv2d and _stride1(self.conv2.conv) and not torch.nn.modules.module._global_forward_hooks):
            return False
        if isinstance(last, Bottleneck):
            tail = last.conv2
            if last.use_add or not isinstance(tail, BaseConv) or tail.spiking() or type(tail.conv) is not nn.Conv2d or not _stride1(tail.conv):
                return False
        else:
            tail = last
        probe = x[:, :1]          # shape / dtype probe for bn_silu_supported (same H, W as every output of the layer)
        return tail.ann_block_ok(probe) and self.conv2.ann_block_ok(probe) and not tail._forward_hooks and not self.conv2._forward_hooks


class Focus(nn.Module):
    """space-to-depth (2x2 -> 4C) followed by a conv; stays an ANN block inside the spiking backbone."""

    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act='silu'):
        super().__init__()
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)

    def forward(self, x):
        if ops.focus_supported(x):
            return self.conv(ops.focus(x))                 # the four strided slices + concatenation as one permutation kernel
        parts = (x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2])
        return self.conv(torch.cat(parts, dim=1))


def enable_spike_planes(net, flag=True):
    """Let every converted (spiking) BaseConv inside ``net`` hand its spikes on as bf16 spike planes (ops "SPIKE PLANES").  For a network
    whose spike tensors stay inside these blocks (the spiking CSPDarknet: every reader is a convolution here, a block that unpacks, or
    CSPDarknet.forward, which unpacks what it hands out)."""
    for m in net.modules():
        if isinstance(m, BaseConv) and m.spiking():
            m.planes_out = bool(flag)
    return net
