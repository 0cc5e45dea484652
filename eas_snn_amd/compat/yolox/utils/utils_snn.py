"""ANN -> SNN module-tree rewrite (reference: yolox/utils/utils_snn.py:16-58)."""
import copy

import torch.nn as nn

from spikingjelly.activation_based import layer, neuron


def is_spiking_neuron(module):
    return isinstance(module, (neuron.BaseNode, neuron.LIFNode, neuron.ParametricLIFNode))


def convert_to_spiking(model, spike_fn):
    from yolox.models.network_blocks import Focus
    for name, module in model.named_children():
        if isinstance(module, Focus):
            # the stem is wrapped but NOT recursed: it keeps SiLU and a plain BatchNorm
            setattr(model, name, layer.SeqToANNContainer(module))
        elif isinstance(module, (nn.Conv2d, nn.Upsample, nn.MaxPool2d)):
            setattr(model, name, layer.SeqToANNContainer(module))
        elif isinstance(module, nn.BatchNorm2d):
            # fresh affine parameters / statistics, like the reference
            setattr(model, name, layer.BatchNorm2d(module.num_features, module.eps, module.momentum, step_mode='m'))
        elif name.endswith('act') or isinstance(module, (nn.ReLU, nn.SiLU, nn.LeakyReLU)):
            setattr(model, name, neuron.ParametricLIFNode(
                init_tau=2.0, decay_input=False, v_threshold=1.0, v_reset=None,
                surrogate_function=copy.deepcopy(spike_fn), detach_reset=False, step_mode='m', backend='hip'))
        else:
            convert_to_spiking(module, spike_fn)
    return model
