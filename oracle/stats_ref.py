"""TEST INFRASTRUCTURE -- CPU restatement of the reference's SOP / energy estimate.

What the reference does (EventEvaluator.energy_estimation, yolox/evaluators/event_evaluator.py:466-565, with the
input-recording hook of yolox/utils/hooks.py:31-44), restated here without its CUDA-only tensor moves.  PINNED: oracle/gen_golden.py
runs the reference method itself on the CPU (its ``.cuda()`` moves patched to the identity) and
tests/test_oracle_golden.py::test_energy_estimation_restatement_matches_the_reference_function holds this file to the figures it
printed (tests/golden/energy_estimation.npz).  The reference does not reset the neurons between batches: ``reset_fn=None``
reproduces that.

* the layers that count are the nn.Conv2d modules found by walking ``model.named_children()``: the children of
  ``model.backbone`` are filed under 'backbone' (the child called 'backbone') or 'fpn' (every other child), the
  remaining top-level children under their own name; inside a group a layer is keyed by its name RELATIVE to the child it
  was found in, so later layers replace earlier ones of the same relative name (:492-510);
* per batch, every counted layer's input is recorded during one forward; a non-embedding layer's single recorded input
  [M, C, H, W] is viewed as [T, M/T, C, H, W], an embedding layer's inputs (one per call) are stacked (:519-531);
* accumulate operations of a layer = sum of ``conv(all-ones weights, zero bias)`` applied to the inputs summed over their
  first axis, except for five named layers with real-valued inputs; multiply-accumulates = the same convolution applied to
  a tensor of ones of that summed shape (:473-487, :533-539);
* energies: 0.9 pJ per accumulate, 4.6 pJ per multiply-accumulate, reported per sample in units of 1e9 (:556-565).

The arithmetic is PyTorch's own convolution, so this file only fixes the bookkeeping.  Only tests/ may import it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

NO_AC_LAYERS = ('input_conv.0', 'input_conv.2', 'gate_conv.2', 'stem.0.conv.conv', 'dark2.0.conv.0')   # :533
GROUPS = ('embedding', 'backbone', 'fpn', 'head')


def counted_layers(model):
    """{group: {relative name: conv module}} with the reference's replacement of equal relative names."""
    table = {g: {} for g in GROUPS}
    for top_name, top in model.named_children():
        if top_name == 'backbone':
            for part_name, part in top.named_children():
                group = 'backbone' if part_name == 'backbone' else 'fpn'
                table[group].update({n: m for n, m in part.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))})
        else:
            table[top_name].update({n: m for n, m in top.named_modules() if isinstance(m, (nn.Conv2d, nn.Linear))})
    return table


def ones_conv_sum(conv, x):
    """sum of the layer's convolution with every weight set to one and no bias, applied to x [N, C, H, W]"""
    w = torch.ones_like(conv.weight, dtype=x.dtype)
    return F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups).sum()


def layer_sop(conv, inputs, count_accumulates, dtype=torch.float64):
    """(accumulate operations, multiply-accumulates) of one layer for its recorded ``inputs`` [K, N, C, H, W]"""
    summed = inputs.to(dtype).sum(0)
    mac = ones_conv_sum(conv, torch.ones_like(summed))
    ac = ones_conv_sum(conv, summed) if count_accumulates else torch.zeros((), dtype=dtype)
    return float(ac), float(mac)


@torch.no_grad()
def energy_estimation(model, batches, T, reset_fn=None):
    """Returns the dictionary eas_snn_amd.stats.energy_estimation returns (float64 sums: the value the reference's float32
    sums over ~1e8 terms approximate)."""
    model.eval()
    table = counted_layers(model)
    module_ac = {g: 0.0 for g in GROUPS}
    module_mac = {g: 0.0 for g in GROUPS}
    num_samples = 0
    for batch in batches:
        imgs = batch[0] if isinstance(batch, (tuple, list)) else batch
        num_samples += len(imgs)
        seen = {(g, n): [] for g in GROUPS for n in table[g]}
        handles = [m.register_forward_hook(lambda mod, inp, out, key=(g, n): seen[key].append(inp[0].detach().clone()))
                   for g in GROUPS for n, m in table[g].items()]
        model(imgs)
        for h in handles:
            h.remove()
        for g in GROUPS:
            for name, conv in table[g].items():
                rec = seen[(g, name)]
                if g == 'embedding':
                    inputs = torch.stack(rec)
                else:
                    assert len(rec) == 1, f'{g}.{name} ran {len(rec)} times'
                    inputs = rec[0].reshape([T, -1] + list(rec[0].shape[1:]))
                ac, mac = layer_sop(conv, inputs, name not in NO_AC_LAYERS)
                module_ac[g] += ac
                module_mac[g] += mac
        if reset_fn is not None:
            reset_fn(model)
    tot_ac, tot_mac = sum(module_ac.values()), sum(module_mac.values())
    return {'module_ac': module_ac, 'module_mac': module_mac, 'tot_ac': tot_ac, 'tot_mac': tot_mac, 'num_samples': num_samples,
            'sop_snn_G': tot_ac / num_samples / 1e9, 'sop_ann_G': tot_mac / num_samples / 1e9,
            'snn_energy': 0.9 * tot_ac / num_samples / 1e9, 'ann_energy': 4.6 * tot_mac / num_samples / 1e9}
