"""Spike-count / synaptic-operation (SOP) statistics and the energy estimate of the reference, on the device.

Reference: ``EventEvaluator.energy_estimation`` (yolox/evaluators/event_evaluator.py:466-565).  It registers a
``RecordHook`` (yolox/utils/hooks.py:31-44) on every ``nn.Conv2d`` / ``nn.Linear``, clones each layer's input to the CPU,
and per layer runs ``calc_layer_sop`` (:473-487): a copy of the layer with all-ones weights applied to the inputs summed
over T (accumulate operations of the spiking net, "SOP_AC") and to a tensor of ones (multiply-accumulates of the
equivalent ANN, "SOP_MAC").  Here the layer inputs never leave the GPU: a tap in ``ops.conv2d`` hands each convolution
input to ``eas_spike_sop`` (one pass at the HBM roofline, wavefront reductions), the MAC count is closed form, and the
kernels that compute the model are the ones that always run (no hooks, so no fallback to the hooked ATen path).

Kept from the reference, quirks included:
* modules are grouped as embedding / backbone / fpn / head by the same walk over ``named_children`` (:492-510), including
  its name collisions in the neck (only the last module of each relative name is counted, see ``classify_layers``);
* the five layers ('input_conv.0', 'input_conv.2', 'gate_conv.2', 'stem.0.conv.conv', 'dark2.0.conv.0') whose inputs are
  real-valued count no accumulate operations (:533-534);
* every other layer counts ``conv(ones-weights)(sum of its inputs)``, also the real-valued neck / head layers;
* a non-embedding layer's input [M,C,H,W] is viewed as [T, M/T, C,H,W] (:528-529), so its MAC count is for M/T samples --
  T times fewer than the batch for the layers after the firing-rate readout, and M must be divisible by T;
* embedding layers are counted per call: MACs of ONE micro-step, accumulate operations summed over all micro-steps.
Deviation: the sampler here always runs its Tm micro-steps (the reference leaves the loop once every pixel has fired Ts
times, embedding.py:200-201, which needs all 2*H*W pixels of every sample to fire); the spikes of all steps are counted.
"""
import torch
import torch.nn as nn

from . import ops

NO_AC_LAYERS = ('input_conv.0', 'input_conv.2', 'gate_conv.2', 'stem.0.conv.conv', 'dark2.0.conv.0')
GROUPS = ('embedding', 'backbone', 'fpn', 'head')


def classify_layers(model):
    """{module: (group, name)} for the nn.Conv2d layers the reference counts, as event_evaluator.py:492-510 builds
    ``cali_layers``: a dictionary per group keyed by the layer's name RELATIVE to the child it was found in.  For the neck
    these relative names collide ('conv' of lateral_conv0 / reduce_conv1 / bu_conv2 / bu_conv1, 'conv1.conv' of the four
    C3 blocks, ...) and the later module replaces the earlier one, so only the last module of each name is counted
    (C3_n4 and bu_conv1 for YOLOPAFPN) -- reproduced here."""
    cali = {g: {} for g in GROUPS}
    for child_name, child in model.named_children():
        if child_name == 'backbone':
            for sub_name, sub in child.named_children():
                group = 'backbone' if sub_name == 'backbone' else 'fpn'
                for name, m in sub.named_modules():
                    if isinstance(m, nn.Linear):
                        raise NotImplementedError('nn.Linear layers are not part of (S)YOLOX')
                    if isinstance(m, nn.Conv2d):
                        cali[group][name] = m
        else:
            if child_name not in GROUPS:
                raise KeyError(f'unexpected top-level module {child_name!r} (the reference knows {GROUPS})')
            for name, m in child.named_modules():
                if isinstance(m, nn.Conv2d):
                    cali[child_name][name] = m
    return {m: (group, name) for group, layers in cali.items() for name, m in layers.items()}


def fan_sum(n_in, k, stride):
    """sum over input positions of the number of output positions that read them (1-D, padding (k-1)//2)."""
    pad = (k - 1) // 2
    n_out = (n_in + 2 * pad - k) // stride + 1
    total = 0
    for i in range(n_in):
        for t in range(k):
            num = i + pad - t
            if num >= 0 and num % stride == 0 and num // stride < n_out:
                total += 1
    return total


def conv_mac(conv, samples, H, W):
    """conv(ones-weights)(ones[samples, Cin, H, W]).sum() in closed form."""
    kh, kw = conv.kernel_size
    return samples * conv.in_channels * (conv.out_channels // conv.groups) * fan_sum(H, kh, conv.stride[0]) * fan_sum(W, kw, conv.stride[1])


class SopMeter:
    """The tap installed by ``energy_estimation``: accumulates per-group accumulate operations on the device (float64,
    no host synchronisation per layer) and MAC counts on the host (exact integers)."""

    def __init__(self, model, T):
        self.layers = classify_layers(model)
        self.T = int(T)
        self.sampler_spikes = []
        self.ac = None
        self.mac = {g: 0 for g in GROUPS}
        self.spikes = None
        self.seen = set()

    def _add_ac(self, group, x, conv):
        kh, kw = conv.kernel_size
        if kh != kw or conv.stride[0] != conv.stride[1] or conv.padding[0] != (kh - 1) // 2 or conv.dilation != (1, 1):
            raise NotImplementedError('SOP statistics: square kernels with "same"-style padding only')
        r = ops.spike_sop(x, kh, conv.stride[0], conv.out_channels // conv.groups)
        if self.ac is None:
            self.ac = torch.zeros(len(GROUPS), dtype=torch.float64, device=x.device)
            self.spikes = torch.zeros(len(GROUPS), dtype=torch.float64, device=x.device)
        self.ac[GROUPS.index(group)] += r[1]
        self.spikes[GROUPS.index(group)] += r[0]

    def __call__(self, conv, x, replicas=1):
        """a dense convolution of backbone / fpn / head is about to run on x [M,C,H,W] (standing for ``replicas`` copies)."""
        if conv not in self.layers:
            return
        group, name = self.layers[conv]
        if group == 'embedding':
            return                                  # counted by embedding_layers()
        M = x.shape[0] * replicas
        if M % self.T:
            raise ValueError(f'{name}: {M} input images are not divisible by T={self.T} (event_evaluator.py:529 reshapes by T)')
        self.seen.add(conv)
        self.mac[group] += conv_mac(conv, M // self.T, x.shape[-2], x.shape[-1])
        if name not in NO_AC_LAYERS:
            if replicas != 1:
                raise NotImplementedError('a de-duplicated layer with spike input')
            self._add_ac(group, x, conv)

    def embedding_layers(self, embedding, N, H, W):
        """count the sampler's convolutions after its forward (they run inside one autograd node, not through ops.conv2d)."""
        for stack_name in ('input_conv', 'gate_conv'):
            stack = getattr(embedding, stack_name, None)
            if stack is None:
                continue
            for idx, m in enumerate(stack):
                if not isinstance(m, nn.Conv2d):
                    continue
                name = f'{stack_name}.{idx}'
                self.seen.add(m)
                self.mac['embedding'] += conv_mac(m, N, H, W)
                if name not in NO_AC_LAYERS:
                    if name != 'gate_conv.0':
                        raise NotImplementedError(f'accumulate count of {name}')
                    for s in self.sampler_spikes:
                        self._add_ac('embedding', s, m)
        self.sampler_spikes = []

    def result(self, num_samples):
        ac = self.ac.cpu().tolist() if self.ac is not None else [0.0] * len(GROUPS)
        sp = self.spikes.cpu().tolist() if self.spikes is not None else [0.0] * len(GROUPS)
        module_ac = dict(zip(GROUPS, ac))
        tot_ac, tot_mac = sum(ac), sum(self.mac.values())
        return {'module_ac': module_ac, 'module_mac': dict(self.mac), 'module_spikes': dict(zip(GROUPS, sp)), 'tot_ac': tot_ac,
                'tot_mac': tot_mac, 'num_samples': num_samples, 'sop_snn_G': tot_ac / num_samples / 1e9,
                'sop_ann_G': tot_mac / num_samples / 1e9, 'snn_energy': 0.9 * tot_ac / num_samples / 1e9,
                'ann_energy': 4.6 * tot_mac / num_samples / 1e9}


@torch.no_grad()
def energy_estimation(model, batches, T, reset_between_batches=False):
    """``EventEvaluator.energy_estimation`` (event_evaluator.py:466-565) over an iterable of input batches
    [B,Tl,Tm,2,H,W] (or (imgs, ...) tuples as the reference's loader yields).  Returns the numbers the reference prints.

    Like the reference, the neurons are NOT reset between batches by default (its loop at :515-541 only calls ``model(imgs)``):
    the membrane potentials left by one batch are the initial state of the next, so all batches must have the same shape.
    ``reset_between_batches=True`` makes every batch start from rest.  The network is reset once at the end."""
    from spikingjelly.activation_based import functional
    was_training = model.training
    model.eval()
    meter = SopMeter(model, T)
    prev = ops.set_conv_sink(meter)
    num = 0
    keep_state = ops.state_writeback_scope(True)          # the carried state must reach HBM whatever the caller's setting
    keep_state.__enter__()
    try:
        for batch in batches:
            imgs = batch[0] if isinstance(batch, (tuple, list)) else batch
            num += len(imgs)
            model(imgs)
            emb = getattr(model, 'embedding', None)
            if emb is not None and imgs.dim() >= 5:
                meter.embedding_layers(emb, imgs.shape[0] * (imgs.shape[1] if imgs.dim() > 5 else 1), imgs.shape[-2], imgs.shape[-1])
            if reset_between_batches:
                functional.reset_net(model)
    finally:
        keep_state.__exit__(None, None, None)
        functional.reset_net(model)
        ops.set_conv_sink(prev)
        model.train(was_training)
    return meter.result(num)
