"""ANN -> SNN module-tree rewrite (interface of the reference's yolox/utils/utils_snn.py:16-58).

The rewrite is a rule table applied to every child of a module, first match wins; unmatched children are descended into.
The rules (and therefore the resulting ``state_dict`` keys, pinned by tests/golden/model_*.npz) are the reference's:

  Focus                          -> SeqToANNContainer(Focus)   and NOT descended into (the stem keeps SiLU + plain BatchNorm)
  Conv2d / Upsample / MaxPool2d  -> SeqToANNContainer(module)
  BatchNorm2d                    -> multi-step BatchNorm2d with FRESH parameters and statistics (same C, eps, momentum)
  child named ``*act`` or a ReLU / SiLU / LeakyReLU -> ParametricLIFNode(tau 2, no input decay, v_th 1, soft reset,
                                    own copy of the surrogate, reset not detached, multi-step)
"""
import copy

import torch.nn as nn

from spikingjelly.activation_based import layer, neuron

_ANN_ACTIVATIONS = (nn.ReLU, nn.SiLU, nn.LeakyReLU)


def is_spiking_neuron(module):
    return isinstance(module, neuron.BaseNode)        # LIFNode / ParametricLIFNode derive from it


def _plif(spike_fn):
    return neuron.ParametricLIFNode(init_tau=2.0, decay_input=False, v_threshold=1.0, v_reset=None, detach_reset=False,
                                    surrogate_function=copy.deepcopy(spike_fn), step_mode='m', backend='hip')


def _replacement(name, child, spike_fn):
    """the module that takes ``child``'s place, or None when the child is to be descended into"""
    from yolox.models.network_blocks import Focus
    # rule order as in the reference (utils_snn.py:23-56): the name rule for '*act' children comes before the MaxPool2d rule
    if isinstance(child, Focus) or isinstance(child, (nn.Conv2d, nn.Upsample)):
        return layer.SeqToANNContainer(child)
    if isinstance(child, nn.BatchNorm2d):
        return layer.BatchNorm2d(child.num_features, child.eps, child.momentum, step_mode='m')
    if name.endswith('act') or isinstance(child, _ANN_ACTIVATIONS):
        return _plif(spike_fn)
    if isinstance(child, nn.MaxPool2d):
        return layer.SeqToANNContainer(child)
    return None


def convert_to_spiking(model, spike_fn):
    pending = [model]
    while pending:
        parent = pending.pop()
        for name, child in list(parent.named_children()):
            new = _replacement(name, child, spike_fn)
            if new is None:
                pending.append(child)
            else:
                setattr(parent, name, new)
    return model
