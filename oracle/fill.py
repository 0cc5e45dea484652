"""TEST INFRASTRUCTURE -- procedural, RNG-free parameter fill.

Golden fixtures for whole blocks / models would need the reference's weights
(8.9 M floats for SYOLOX-S).  Instead every parameter and buffer is filled
from a closed-form function of (state_dict key, element index), so that the
fixture generator (running the real reference) and the tests (running the
oracle and the HIP product) build bit-identical weights without storing them.
A CRC of the filled state_dict is stored in each fixture and re-checked.
"""
import zlib

import numpy as np
import torch


def _wave(key, numel, lo, hi):
    """Integer-only hash (splitmix64 finaliser) -> 24-bit uniform -> [lo, hi): exact on every host."""
    seed = np.uint64(zlib.crc32(key.encode()))
    with np.errstate(over='ignore'):
        x = np.arange(numel, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32)


def procedural_fill_(module, conv_gain=1.0, plif_w=None, ann_regex=None, ann_gain=1.0):
    """Fill ``module``'s parameters/buffers in place. Returns crc32 over all filled values.

    ``conv_gain`` scales conv weights in front of spiking neurons (binary inputs need a larger gain to keep
    firing through depth); keys matching ``ann_regex`` are real-valued ANN layers and get ``ann_gain`` and a
    contracting BN gamma so activations stay O(1) through ~30 layers in eval mode."""
    import re
    crc = 0
    ann = re.compile(ann_regex) if ann_regex else None
    sd = module.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        if not torch.is_floating_point(t):
            t.zero_()
            continue
        n = t.numel()
        leaf = key.rsplit('.', 1)[-1]
        is_ann = ann is not None and ann.search(key) is not None
        if t.dim() == 4:                                   # conv weight: uniform, variance ~ gain^2 * 2 / fan_in
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            r = (ann_gain if is_ann else conv_gain) * (6.0 / fan_in) ** 0.5
            v = _wave(key, n, -r, r)
        elif leaf == 'running_var':
            v = _wave(key, n, 0.5, 1.5)
        elif leaf == 'running_mean':
            v = _wave(key, n, -0.2, 0.2)
        elif leaf == 'weight':                             # BN gamma
            v = _wave(key, n, 0.5, 1.0) if is_ann else _wave(key, n, 0.8, 1.6)
        elif leaf == 'bias':
            v = _wave(key, n, -0.1, 0.3)
        elif leaf == 'w':                                  # PLIF decay logit
            v = _wave(key, n, -0.5, 0.5) if plif_w is None else np.full(n, plif_w, np.float32)
        elif leaf == 'decay':
            v = _wave(key, n, -0.3, 0.3)
        else:
            v = _wave(key, n, -0.5, 0.5)
        with torch.no_grad():
            t.copy_(torch.from_numpy(v).view(t.shape))
        crc = zlib.crc32(v.tobytes(), crc)
    return crc


def poisson_events(shape, lam=0.3, seed=0):
    """Synthetic micro-slice count frames (Poisson(0.3) counts)."""
    rng = np.random.default_rng(seed)
    return rng.poisson(lam, size=shape).astype(np.float32)


# which state_dict keys are real-valued ANN layers, per EventExp.use_spike mode
ANN_KEYS = {
    'True': r'^(head\.|backbone\.(?!backbone\.dark))',       # spiking: backbone.backbone.dark2..5 only
    'full_spike': r'^head\.',
    'full_spike_v2': None,
    'False': r'^(head|backbone)\.',
}
