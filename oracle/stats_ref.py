"""TEST INFRASTRUCTURE -- CPU restatement of the reference's SOP / energy estimate.

Follows EventEvaluator.energy_estimation (yolox/evaluators/event_evaluator.py:466-565) and RecordHook
(yolox/utils/hooks.py:31-44) statement by statement, with the ``.cuda()`` / ``torch.cuda.FloatTensor`` moves dropped
(the reference method cannot run without a CUDA device, so it cannot be executed in the build container: this file is
its restatement, and its arithmetic is exactly the reference's -- a deep copy of each layer with all-ones weights applied
by PyTorch itself).  Only tests/ may import this module.
"""
import copy

import torch
import torch.nn as nn

NO_AC_LAYERS = ('input_conv.0', 'input_conv.2', 'gate_conv.2', 'stem.0.conv.conv', 'dark2.0.conv.0')   # :533


class RecordHook:
    """hooks.py:31-44: keeps a clone of the first positional input of every call."""

    def __init__(self):
        self.inputs = []

    def __call__(self, module, input, output):
        self.inputs.append(input[0].data.clone())


def calc_layer_sop(layer, inputs, ac=True):
    """event_evaluator.py:473-487."""
    sop_ac = 0
    spike_counts = inputs.sum(0)
    analog_counts = torch.ones_like(spike_counts)
    aux_layer = copy.deepcopy(layer)
    aux_layer.weight = torch.nn.Parameter(torch.ones_like(aux_layer.weight))
    if aux_layer.bias is not None:
        aux_layer.bias = torch.nn.Parameter(torch.zeros_like(aux_layer.bias))
    if ac:
        sop_ac = aux_layer(spike_counts).sum()
    sop_mac = aux_layer(analog_counts).sum()
    return sop_ac, sop_mac


@torch.no_grad()
def energy_estimation(model, batches, T, reset_fn=None, exact=True):
    """Returns the dictionary eas_snn_amd.stats.energy_estimation returns.  ``exact``: run the all-ones layers in float64
    (the reference sums ~1e8 float32 terms per layer; float64 gives the value those sums approximate)."""
    hook_cls = (nn.Conv2d, nn.Linear)
    model.eval()
    groups = ('embedding', 'backbone', 'fpn', 'head')
    module_ac = {k: 0.0 for k in groups}
    module_mac = {k: 0.0 for k in groups}
    cali_layers = {k: {} for k in groups}
    for m in model.named_children():                                                   # :492-510
        module_name = m[0]
        if m[0] == 'backbone':
            for m_bb in m[1].named_children():
                module_name = 'backbone' if m_bb[0] == 'backbone' else 'fpn'
                for l in m_bb[1].named_modules():
                    if isinstance(l[1], hook_cls):
                        cali_layers[module_name][l[0]] = l[1]
        else:
            for l in m[1].named_modules():
                if isinstance(l[1], hook_cls):
                    cali_layers[module_name][l[0]] = l[1]
    num_samples = 0
    tot_ac = tot_mac = 0.0
    for batch in batches:                                                              # :513-541
        imgs = batch[0] if isinstance(batch, (tuple, list)) else batch
        num_samples += len(imgs)
        hooks = {k: {} for k in groups}
        for key, module_layers in cali_layers.items():
            for name, layer in module_layers.items():
                hooker = RecordHook()
                hooks[key][name] = (layer.register_forward_hook(hooker), hooker)
        model(imgs)
        for key, module_layers in cali_layers.items():
            for name, layer in module_layers.items():
                handler, hooker = hooks[key][name]
                if key != 'embedding':
                    assert len(hooker.inputs) == 1
                    inputs = hooker.inputs[0].reshape([T, -1] + list(hooker.inputs[0].shape[1:]))
                else:
                    inputs = torch.stack(hooker.inputs)
                if_ac = name not in NO_AC_LAYERS
                handler.remove()
                lay = layer
                if exact:
                    lay = copy.deepcopy(layer).double()
                    inputs = inputs.double()
                sop_ac, sop_mac = calc_layer_sop(lay, inputs, ac=if_ac)
                module_ac[key] += float(sop_ac)
                tot_ac += float(sop_ac)
                module_mac[key] += float(sop_mac)
                tot_mac += float(sop_mac)
        if reset_fn is not None:
            reset_fn(model)
    return {'module_ac': module_ac, 'module_mac': module_mac, 'tot_ac': tot_ac, 'tot_mac': tot_mac, 'num_samples': num_samples,
            'sop_snn_G': tot_ac / num_samples / 1e9, 'sop_ann_G': tot_mac / num_samples / 1e9,
            'snn_energy': 0.9 * tot_ac / num_samples / 1e9, 'ann_energy': 4.6 * tot_mac / num_samples / 1e9}
