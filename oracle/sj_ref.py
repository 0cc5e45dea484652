"""TEST INFRASTRUCTURE -- torch-CPU restatement of the spikingjelly 0.0.0.0.14
``activation_based`` API that EAS-SNN calls.

PARITY UNPINNED: spikingjelly==0.0.0.0.14 (reference pip-requirements.txt:135,
conda-env.yml:369, readme.md:17) is a third-party dependency that is absent from
/root/reference and cannot be installed here.  This file restates its published
algorithm for exactly the call sites the reference has:

  yolox/utils/utils_snn.py:13      isinstance checks on neuron.{BaseNode,LIFNode,ParametricLIFNode}
  yolox/utils/utils_snn.py:24-27   layer.SeqToANNContainer(module)
  yolox/utils/utils_snn.py:32      layer.BatchNorm2d(C, eps, momentum, step_mode='m')
  yolox/utils/utils_snn.py:44-53   neuron.ParametricLIFNode(init_tau=2.0, decay_input=False, v_threshold=1.0,
                                   v_reset=None, surrogate_function=..., detach_reset=False, step_mode='m',
                                   backend='torch')
  yolox/exp/event_yolox_base.py:147-148   surrogate.ATan(alpha), surrogate.Sigmoid(alpha)
  yolox/core/trainer.py:116-117, yolox/evaluators/event_evaluator.py:197-198   functional.reset_net(model)

Cross-check available inside the reference: ``LIFLayer``/``LIFCell``
(yolox/models/layer.py:38-69, cell.py:37-65) is the same recurrence with
sigmoid(decay)=0.5 and soft reset, differing only in '>' vs '>=' at v==v_th;
tests/test_oracle_golden.py compares against vectors generated from it.

Arithmetic order follows upstream so that fp32 results are reproducible:
  charge (decay_input=False, v_reset None|0):  v = v * (1. - sigmoid(w)) + x
  fire:                                        s = heaviside(v - v_th),  heaviside(x) = (x >= 0)
  soft reset:                                  v = v - s * v_th       (s detached iff detach_reset)
  hard reset:                                  v = (1. - s) * v + s * v_reset
"""
import copy
import math
import sys
import types

import torch
import torch.nn as nn


# ----------------------------------------------------------------------------- surrogate
class _ATanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        if x.requires_grad:
            ctx.save_for_backward(x)
            ctx.alpha = alpha
        return (x >= 0).to(x)

    @staticmethod
    def backward(ctx, grad_output):
        x, = ctx.saved_tensors
        a = ctx.alpha
        return a / 2 / (1 + (math.pi / 2 * a * x).pow(2)) * grad_output, None


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        if x.requires_grad:
            ctx.save_for_backward(x)
            ctx.alpha = alpha
        return (x >= 0).to(x)

    @staticmethod
    def backward(ctx, grad_output):
        x, = ctx.saved_tensors
        sg = (x * ctx.alpha).sigmoid()
        return grad_output * (1. - sg) * sg * ctx.alpha, None


class _SurrogateBase(nn.Module):
    def __init__(self, alpha, spiking=True):
        super().__init__()
        self.alpha = alpha
        self.spiking = spiking

    def extra_repr(self):
        return f"alpha={self.alpha}, spiking={self.spiking}"


class ATan(_SurrogateBase):
    """g'(x) = alpha/2 / (1 + (pi/2*alpha*x)^2)."""

    def __init__(self, alpha=2.0, spiking=True):
        super().__init__(alpha, spiking)

    def forward(self, x):
        if self.spiking:
            return _ATanFn.apply(x, self.alpha)
        return (math.pi / 2 * self.alpha * x).atan() / math.pi + 0.5


class Sigmoid(_SurrogateBase):
    """g'(x) = alpha * sigmoid(alpha x) * (1 - sigmoid(alpha x))."""

    def __init__(self, alpha=4.0, spiking=True):
        super().__init__(alpha, spiking)

    def forward(self, x):
        if self.spiking:
            return _SigmoidFn.apply(x, self.alpha)
        return (x * self.alpha).sigmoid()


class InvArcTanh(nn.Module):
    """sigma(x) = atan(pi/2 * |alpha| * x) / pi + 1/2 with a learnable alpha of shape [1]
    (restates yolox/models/activation.py:121-130; pinned by tests/golden/lif_patan*.npz generated from the reference class)."""

    def __init__(self, alpha=1.0, learnable=True):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor([float(alpha)])) if learnable else alpha

    def forward(self, x, gates=None):
        return 1.0 / math.pi * torch.atan(math.pi / 2.0 * torch.abs(self.alpha) * x) + 0.5


class EfficientNoisySpikeII(nn.Module):
    """'patan' = EfficientNoisySpikeII(InvArcTanh(alpha), p=0) (event_yolox_base.py:145-150; activation.py:181-205).
    Training: sigma(x) + ((x >= 0) - sigma(x)) * mask, detached second term, mask ~ Bernoulli(1 - p) drawn once;
    eval: Heaviside.  With p = 0 the mask is all ones: the forward value is exactly the Heaviside step (for x >= 0
    sigma >= 1/2, so 1 - sigma and sigma + (1 - sigma) are exact in fp32) and the gradient is sigma's."""

    def __init__(self, inv_sg, p=0.5, spike=True):
        super().__init__()
        self.inv_sg = inv_sg
        self.p = p
        self.spike = spike
        self.mask = None

    def forward(self, x, gates=None):
        sigx = self.inv_sg(x, gates)
        if self.training:
            if self.mask is None:
                self.mask = torch.bernoulli(torch.ones_like(x) * (1 - self.p))
            return sigx + (((x >= 0).float() - sigx) * self.mask).detach()
        return (x >= 0).float() if self.spike else sigx


def PATan(alpha=2.0):
    return EfficientNoisySpikeII(InvArcTanh(alpha), p=0)


# ----------------------------------------------------------------------------- base
class MemoryModule(nn.Module):
    """Stateful module: memories live outside ``state_dict`` and are restored by ``reset()``."""

    def __init__(self):
        super().__init__()
        self._memories = {}
        self._memories_rv = {}
        self.step_mode = 's'
        self.backend = 'torch'

    def register_memory(self, name, value):
        self._memories[name] = value
        self._memories_rv[name] = copy.deepcopy(value)

    def reset(self):
        for k in self._memories:
            self._memories[k] = copy.deepcopy(self._memories_rv[k])

    def __getattr__(self, name):
        if '_memories' in self.__dict__ and name in self.__dict__['_memories']:
            return self.__dict__['_memories'][name]
        return super().__getattr__(name)

    def __setattr__(self, name, value):
        mem = self.__dict__.get('_memories')
        if mem is not None and name in mem:
            mem[name] = value
        else:
            super().__setattr__(name, value)

    def forward(self, *args, **kwargs):
        if self.step_mode == 's':
            return self.single_step_forward(*args, **kwargs)
        if self.step_mode == 'm':
            return self.multi_step_forward(*args, **kwargs)
        raise ValueError(self.step_mode)


# ----------------------------------------------------------------------------- neuron
class BaseNode(MemoryModule):
    def __init__(self, v_threshold=1., v_reset=0., surrogate_function=None, detach_reset=False,
                 step_mode='s', backend='torch', store_v_seq=False):
        super().__init__()
        assert isinstance(v_reset, float) or v_reset is None
        assert isinstance(v_threshold, float)
        self.register_memory('v', 0. if v_reset is None else v_reset)
        self.v_threshold = v_threshold
        self.v_reset = v_reset
        self.detach_reset = detach_reset
        self.surrogate_function = surrogate_function if surrogate_function is not None else Sigmoid()
        self.step_mode = step_mode
        self.backend = backend
        self.store_v_seq = store_v_seq

    def neuronal_charge(self, x):
        raise NotImplementedError

    def neuronal_fire(self):
        return self.surrogate_function(self.v - self.v_threshold)

    def neuronal_reset(self, spike):
        spike_d = spike.detach() if self.detach_reset else spike
        if self.v_reset is None:
            self.v = self.v - spike_d * self.v_threshold
        else:
            self.v = (1. - spike_d) * self.v + spike_d * self.v_reset

    def v_float_to_tensor(self, x):
        if isinstance(self.v, float):
            self.v = torch.full_like(x.data, self.v)

    def single_step_forward(self, x):
        self.v_float_to_tensor(x)
        self.neuronal_charge(x)
        spike = self.neuronal_fire()
        self.neuronal_reset(spike)
        return spike

    def multi_step_forward(self, x_seq):
        out = []
        v_seq = []
        for t in range(x_seq.shape[0]):
            out.append(self.single_step_forward(x_seq[t]))
            if self.store_v_seq:
                v_seq.append(self.v)
        if self.store_v_seq:
            self.v_seq = torch.stack(v_seq)
        return torch.stack(out)


class IFNode(BaseNode):
    def neuronal_charge(self, x):
        self.v = self.v + x


class LIFNode(BaseNode):
    def __init__(self, tau=2., decay_input=True, v_threshold=1., v_reset=0., surrogate_function=None,
                 detach_reset=False, step_mode='s', backend='torch', store_v_seq=False):
        assert isinstance(tau, float) and tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.tau = tau
        self.decay_input = decay_input

    def neuronal_charge(self, x):
        vr = 0. if self.v_reset is None else self.v_reset
        if self.decay_input:
            if vr == 0.:
                self.v = self.v + (x - self.v) / self.tau
            else:
                self.v = self.v + (x - (self.v - vr)) / self.tau
        else:
            if vr == 0.:
                self.v = self.v * (1. - 1. / self.tau) + x
            else:
                self.v = self.v - (self.v - vr) / self.tau + x


class ParametricLIFNode(BaseNode):
    def __init__(self, init_tau=2.0, decay_input=True, v_threshold=1., v_reset=0., surrogate_function=None,
                 detach_reset=False, step_mode='s', backend='torch', store_v_seq=False):
        assert isinstance(init_tau, float) and init_tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.decay_input = decay_input
        self.w = nn.Parameter(torch.as_tensor(-math.log(init_tau - 1.)))

    def neuronal_charge(self, x):
        vr = 0. if self.v_reset is None else self.v_reset
        if self.decay_input:
            if vr == 0.:
                self.v = self.v + (x - self.v) * self.w.sigmoid()
            else:
                self.v = self.v + (x - (self.v - vr)) * self.w.sigmoid()
        else:
            if vr == 0.:
                self.v = self.v * (1. - self.w.sigmoid()) + x
            else:
                self.v = self.v - (self.v - vr) * self.w.sigmoid() + x


# ----------------------------------------------------------------------------- layer
def seq_to_ann_forward(x_seq, stateless):
    y_shape = [x_seq.shape[0], x_seq.shape[1]]
    y = stateless(x_seq.flatten(0, 1))
    y_shape.extend(y.shape[1:])
    return y.view(y_shape)


class SeqToANNContainer(nn.Sequential):
    """[T, N, ...] -> flatten(T, N) -> wrapped stateless modules -> [T, N, ...]."""

    def forward(self, x_seq):
        return seq_to_ann_forward(x_seq, super().forward)


class BatchNorm2d(nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, step_mode='s'):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.step_mode = step_mode

    def forward(self, x):
        if self.step_mode == 's':
            return super().forward(x)
        # statistics over T*N*H*W per channel
        return seq_to_ann_forward(x, super().forward)


# ----------------------------------------------------------------------------- functional
def reset_net(net):
    for m in net.modules():
        if hasattr(m, 'reset'):
            m.reset()


# ----------------------------------------------------------------------------- stand-in registration
def install_as_spikingjelly():
    """Register this restatement under the import name the reference uses.

    Only the fixture generator (oracle/gen_golden.py, run in the build
    container where /root/reference exists) calls this.
    """
    me = sys.modules[__name__]
    root = types.ModuleType('spikingjelly')
    ab = types.ModuleType('spikingjelly.activation_based')
    subs = {}
    for name, symbols in {
        'neuron': ['BaseNode', 'IFNode', 'LIFNode', 'ParametricLIFNode'],
        'layer': ['SeqToANNContainer', 'BatchNorm2d'],
        'surrogate': ['ATan', 'Sigmoid'],
        'functional': ['reset_net', 'seq_to_ann_forward'],
        'base': ['MemoryModule'],
    }.items():
        m = types.ModuleType(f'spikingjelly.activation_based.{name}')
        for s in symbols:
            setattr(m, s, getattr(me, s))
        subs[name] = m
        setattr(ab, name, m)
        sys.modules[m.__name__] = m
    root.activation_based = ab
    root.__path__ = []
    ab.__path__ = []
    sys.modules['spikingjelly'] = root
    sys.modules['spikingjelly.activation_based'] = ab
    return subs
