"""development: which of (packed forward, unpacked forward, repeated forward) differ after fused-Adam steps"""
import contextlib
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eas_snn_amd
from eas_snn_amd import ops, data
from oracle import fill
from spikingjelly.activation_based import functional
from yolox.exp import get_exp

dev = torch.device('cuda:0')
exp = get_exp(None, 'e-yolox-s')
exp.merge(['T', '3', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
           'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'spike_fn', 'atan', 'use_spike', 'True'])
hip = exp.get_model()
fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS['True'])
hip.to(dev)
opt = exp.get_optimizer(2)
for gr in opt.param_groups:
    gr['lr'] = 3e-3
x = torch.from_numpy(fill.poisson_events((2, 1, 4, 2, 64, 96), 0.5, seed=11)).to(dev)
tg = data.synth_targets(2, (64, 96), dev)


def fwd(packed=True):
    real = ops.packed_weights
    if not packed:
        ops.packed_weights = lambda model: contextlib.nullcontext()
    try:
        with torch.no_grad():
            o = hip(x).clone()
    finally:
        ops.packed_weights = real
    functional.reset_net(hip)
    return o


hip.eval()
b0, b1, b2 = fwd(), fwd(False), fwd()
print('before: packed==unpacked', torch.equal(b0, b1), 'packed==packed', torch.equal(b0, b2), float((b0 - b1).abs().max()))
hip.train(); hip.head.use_l1 = True
for _ in range(2):
    out = hip(x, tg)
    opt.zero_grad(set_to_none=True)
    out['total_loss'].backward()
    opt.step()
    functional.reset_net(hip)
hip.eval()
a0, a1, a2, a3 = fwd(), fwd(False), fwd(), fwd(False)
print('after: p==u', torch.equal(a0, a1), 'p==p', torch.equal(a0, a2), 'u==u', torch.equal(a1, a3), float((a0 - a1).abs().max()), float((a0 - b0).abs().max()))
# which layer differs first: hook conv outputs
names, outs = [], {}
def run(tag, packed):
    hs = []
    for n, m in hip.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            hs.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.setdefault((tag, n), o.detach().clone())))
    fwd(packed)
    for h in hs:
        h.remove()
# hooks force the module path (conv(x) directly) - so instead compare BN-layer inputs via BaseConv hooks
from yolox.models.network_blocks import BaseConv
def run2(tag, packed):
    hs = []
    for n, m in hip.named_modules():
        if isinstance(m, BaseConv):
            hs.append(m.register_forward_hook(lambda mod, i, o, n=n: outs.setdefault((tag, n), (o[0] if isinstance(o, tuple) else o).detach().clone())))
    fwd(packed)
    for h in hs:
        h.remove()
run2('p', True); run2('u', False)
for (tag, n), o in outs.items():
    if tag == 'p':
        u = outs.get(('u', n))
        if u is not None and not torch.equal(o, u):
            print('first differing BaseConv:', n, float((o - u).abs().max()), o.shape)
            break
else:
    print('no BaseConv output differs')
