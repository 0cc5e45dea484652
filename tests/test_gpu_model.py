"""GPU parity of the assembled blocks / whole SYOLOX against golden vectors from the reference and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-4

BASE_OPTS = ['T', '3', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
             'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'spike_fn', 'atan']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    import eas_snn_amd
    eas_snn_amd.hip_library()
    torch.backends.cudnn.benchmark = False
    return torch.device('cuda:0')


def _frac_close(a, b, rtol, atol):
    return float(np.isclose(a, b, rtol=rtol, atol=atol).mean())


def _blocks():
    from yolox.models.network_blocks import BaseConv, CSPLayer, SPPBottleneck
    return {'baseconv1x1': lambda: BaseConv(8, 16, 1, 1), 'baseconv3x3s2': lambda: BaseConv(8, 16, 3, 2),
            'csp': lambda: CSPLayer(8, 8, n=2), 'spp': lambda: SPPBottleneck(8, 8)}


@pytest.mark.parametrize('kind', ['baseconv1x1', 'baseconv3x3s2', 'csp', 'spp'])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_blocks_golden(dev, kind, mode):
    from oracle import fill
    from spikingjelly.activation_based import functional, surrogate
    from yolox.utils.utils_snn import convert_to_spiking
    g = load_golden(f'block_{kind}_{mode}')
    mod = convert_to_spiking(_blocks()[kind](), surrogate.ATan(2.0))
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    assert fill.procedural_fill_(mod, conv_gain=2.5) == int(g['crc'])
    mod.to(dev).train(mode == 'train')
    x = torch.from_numpy(g['x']).to(dev).requires_grad_(True)
    out = mod(x)
    got = out.detach().cpu().numpy()
    flips = float((got != g['out']).mean())
    assert flips < 5e-3, f'spike flip fraction {flips:.2e}'
    if mode == 'train':
        out.backward(torch.from_numpy(g['gout']).to(dev))
        for n, b in mod.named_buffers():
            np.testing.assert_allclose(b.cpu().numpy(), g[f'buf/{n}'], rtol=1e-4, atol=1e-5, err_msg=n)
        # gradients are compared unconditionally: the four train fixtures have no spike flip on MI355X (asserted), so every
        # element must match
        nflip = int((got != g['out']).sum())
        print(f'block_{kind}_{mode}: {nflip} spike flips of {got.size}')
        assert nflip == 0, f'block_{kind}_train: {nflip} spikes differ from the reference'
        np.testing.assert_allclose(x.grad.cpu().numpy(), g['gx'], rtol=2e-3, atol=2e-5)
        for n, p in mod.named_parameters():
            np.testing.assert_allclose(p.grad.cpu().numpy(), g[f'grad/{n}'], rtol=5e-3, atol=5e-4, err_msg=n)
    functional.reset_net(mod)


MODELS = {
    'model_s_true_64': ('e-yolox-s', ['use_spike', 'True']),
    'model_s_full_64': ('e-yolox-s', ['use_spike', 'full_spike']),
    'model_s_fullv2_64': ('e-yolox-s', ['use_spike', 'full_spike_v2']),
    'model_s_false_64': ('e-yolox-s', ['use_spike', 'False']),
    'model_s_true_256x320': ('e-yolox-s', ['use_spike', 'True']),
    'model_m_fullv2_t5_64x96': ('e-yolox-m', ['use_spike', 'full_spike_v2', 'T', '5']),
    'model_s_true_ts3_64': ('e-yolox-s', ['use_spike', 'True', 'Ts', '3']),
    'model_s_true_patan_64': ('e-yolox-s', ['use_spike', 'True', 'spike_fn', 'patan']),
}


USE_SPIKE = {k: v[1][1] for k, v in MODELS.items()}


def _build(name, dev):
    from oracle import fill
    from yolox.exp import get_exp
    g = load_golden(name)
    exp_name, extra = MODELS[name]
    exp = get_exp(None, exp_name)
    exp.merge(BASE_OPTS + extra)
    model = exp.get_model()
    assert [str(k) for k in g['keys']] == list(model.state_dict().keys())
    assert fill.procedural_fill_(model, conv_gain=float(g['gain']), ann_regex=fill.ANN_KEYS[USE_SPIKE[name]]) == int(g['crc'])
    return g, model.to(dev)


@pytest.mark.parametrize('name', sorted(MODELS))
def test_model_logits_golden(dev, name):
    """Detection logits vs the reference (CPU).  MIOpen convs sum in a different order than ATen-CPU, so a neuron whose
    potential lies within fp32 rounding of v_th can flip and perturb a few anchors; the bulk must match to 1e-4 and
    the report prints the fraction that does."""
    from spikingjelly.activation_based import functional
    g, model = _build(name, dev)
    model.eval()
    with torch.no_grad():
        logits = model(torch.from_numpy(g['x']).to(dev))
    functional.reset_net(model)
    got, ref = logits.cpu().numpy(), g['logits']
    assert got.shape == ref.shape and np.isfinite(got).all()
    frac = _frac_close(got, ref, RTOL, 1e-4)
    print(f'{name}: {frac * 100:.3f}% of logits within 1e-4 rel; max abs diff {np.abs(got - ref).max():.3e}')
    if name == 'model_s_true_256x320':
        # ~3 M neuron-steps per layer: a handful of neurons sit within conv rounding error of v_th, and the spiking
        # backbone amplifies every flip (chaotic cascade), so end-to-end bit parity at this size is not attainable by
        # ANY conv that sums in another order than ATen-CPU.  Parity at this size is asserted layer by layer with
        # oracle inputs (test_layerwise_teacher_forced_parity_256x320: the spiking backbone; test_neck_and_head_teacher_forced_every_logit:
        # the real-valued neck + head, EVERY logit at 1e-4); end to end only the bulk statistics can be asserted.
        assert abs(np.median(got[..., 4]) - np.median(ref[..., 4])) < 0.05 and abs(got[..., 2:4].mean() / ref[..., 2:4].mean() - 1) < 0.1
        return
    assert frac > 0.97
    assert np.median(np.abs(got - ref) / (np.abs(ref) + 1e-3)) < 1e-5


def _exp_opts(cfg, H, W):
    opts = list(BASE_OPTS) + ['input_size', f'({H},{W})', 'test_size', f'({H},{W})']
    for k, v in cfg.items():
        if k in ('depth', 'width'):
            continue
        if k in opts[::2]:
            opts[opts[::2].index(k) * 2 + 1] = str(v)
        else:
            opts += [k, str(v)]
    return opts


def _teacher_forced(dev, exp_name, cfg, shape, train, max_flip=2e-5):
    """Every spiking conv->BN->PLIF block of the model, fed the ORACLE's input for that block while the oracle runs its forward:
    spikes may differ only where the potential is within conv rounding of the threshold (flip fraction per layer < max_flip), the
    membrane potentials of all other neurons agree to 1e-4 relative (the north_star criterion), and in train mode the BatchNorm
    running statistics agree.  Returns (layers, flips, neuron-steps, worst membrane error)."""
    from eas_snn_amd import ops
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    H, W = shape[-2:]
    exp = get_exp(None, exp_name)
    exp.merge(_exp_opts(cfg, H, W))
    hip = exp.get_model()
    ref = model_ref.build_model(**cfg)
    rx = fill.ANN_KEYS[cfg['use_spike']]
    assert fill.procedural_fill_(hip, 2.0, ann_regex=rx) == fill.procedural_fill_(ref, 2.0, ann_regex=rx)
    hip.to(dev).train(train); ref.train(train)
    assert ops.state_writeback()
    stats = dict(layers=0, flips=0, steps=0, worst_v=0.0, worst_flip=0.0, worst_rm=0.0)

    def mk(name):
        hm = hip.get_submodule(name)

        def f(mod, inp, out):
            x_in = inp[0].detach()
            xd = x_in.to(dev)
            if float(x_in.abs().max()) <= 16 and bool((x_in == x_in.round()).all()):
                ops.mark_small_int(xd)                   # spikes / SEW sums: the one-term convolution path, as inside the model
            with torch.no_grad():
                got = hm(xd)
            got = (got[0] if isinstance(got, tuple) else got).cpu().numpy()
            want = (out[0] if isinstance(out, tuple) else out).detach().numpy()
            v_ref = mod.act.v.detach().numpy()
            vh = hm.act.v.cpu().numpy()
            functional.reset_net(hm)
            flips = got != want
            same = ~flips.any(axis=0)                    # neurons whose spike train is identical
            err = float((np.abs(vh - v_ref)[same] / (np.abs(v_ref)[same] + 1.0)).max())
            stats['layers'] += 1; stats['flips'] += int(flips.sum()); stats['steps'] += flips.size
            stats['worst_v'] = max(stats['worst_v'], err); stats['worst_flip'] = max(stats['worst_flip'], float(flips.mean()))
            assert flips.mean() < max_flip, f'{name}: flip fraction {flips.mean():.2e}'
            assert err < RTOL, f'{name}: membrane potential error {err:.2e}'
            if train:
                rm, rv = hm.bn.running_mean.cpu().numpy(), hm.bn.running_var.cpu().numpy()
                np.testing.assert_allclose(rm, mod.bn.running_mean.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)
                np.testing.assert_allclose(rv, mod.bn.running_var.numpy(), rtol=1e-4, atol=1e-6, err_msg=name)
        return f
    names = [n for n, m in ref.named_modules() if isinstance(m, model_ref.BaseConv) and isinstance(getattr(m, 'act', None), sj_ref.BaseNode)]
    hooks = [ref.get_submodule(n).register_forward_hook(mk(n)) for n in names]
    # the prediction convolutions (reg / obj / cls of every level; no neuron behind them, so no flip excuse): the oracle's input in, EVERY
    # output element within 1e-4 (relative + 1e-4 of the map's largest value); their outputs are kept for the decode check below
    raws = {}
    stats['pred_worst'] = 0.0

    def mkpred(kind, k):
        hm = getattr(hip.head, f'{kind}_preds')[k]

        def f(mod, inp, out):
            from yolox.models.yolo_head import _pred
            xd = inp[0].detach().to(dev)
            if float(xd.abs().max()) <= 16 and bool((xd == xd.round()).all()):
                ops.mark_small_int(xd)
            with torch.no_grad():
                got = _pred(hm, xd)
            want = out.detach()
            if want.dim() == 5:                         # converted head (full_spike_v2): mean current over T (spiking_yolo_head.py:175-178)
                got, want = ops.time_mean(got), want.mean(0)
            raws[(kind, k)] = want
            g, w = got.cpu().numpy(), want.numpy()
            err = np.abs(g - w) / (RTOL * np.abs(w) + RTOL * max(float(np.abs(w).max()), 1e-6))
            stats['pred_worst'] = max(stats['pred_worst'], float(err.max()))
            assert err.max() <= 1.0, f'{kind}_preds[{k}]: worst element {err.max():.2f} x the 1e-4 tolerance'
        return f
    for k in range(3):
        for kind in ('reg', 'obj', 'cls'):
            hooks.append(getattr(ref.head, f'{kind}_preds')[k].register_forward_hook(mkpred(kind, k)))
    x = torch.from_numpy(fill.poisson_events(shape, 0.5, seed=3))
    with torch.no_grad():
        if train:
            tg = torch.zeros(shape[0], 50, 5)
            tg[:, 0] = torch.tensor([0, 0.3 * W, 0.4 * H, 0.25 * W, 0.3 * H])
            ref(x, tg)
        else:
            want_logits = ref(x)
    for h in hooks:
        h.remove()
    sj_ref.reset_net(ref)
    assert stats['layers'] == len(names) and len(raws) == 9
    if not train:
        # decode (sigmoid, level concatenation, grid / stride arithmetic, exp) on the oracle's raw maps: every output element
        with torch.no_grad():
            got = hip.head.assemble_eval([tuple(raws[(kind, k)].to(dev) for kind in ('reg', 'obj', 'cls')) for k in range(3)]).cpu().numpy()
        np.testing.assert_allclose(got, want_logits.numpy(), rtol=RTOL, atol=RTOL)
    print(f"teacher-forced {exp_name} {cfg.get('use_spike')} T={cfg.get('T', 3)} {H}x{W} {'train' if train else 'eval'}: {stats['layers']} layers, "
          f"{stats['flips']} spike flips in {stats['steps']} neuron-steps (worst layer {stats['worst_flip']:.2e}), worst membrane-potential "
          f"rel err {stats['worst_v']:.2e}; prediction convolutions: worst element at {stats['pred_worst']:.2f} of the 1e-4 tolerance")
    return stats


def test_layerwise_teacher_forced_parity_256x320(dev):
    """BASELINE configs[1] (SYOLOX-S, T=3, 256x320): all 34 spiking blocks, eval-mode BatchNorm (train mode, and the backward of every
    block, for all four configurations: tests/test_gpu_bench_shapes.py, together with the proof that the bench's kernel instances are
    the checked ones)."""
    st = _teacher_forced(dev, 'e-yolox-s', dict(use_spike='True'), (1, 1, 4, 2, 256, 320), False)
    assert st['layers'] == 34


@pytest.mark.parametrize('train', [False, True])
def test_neck_and_head_teacher_forced_every_logit(dev, train):
    """BASELINE configs[1] at the benchmark canvas (SYOLOX-S, use_spike=True, 256x320, B=2): the real-valued PAFPN neck and the head,
    fed the ORACLE's backbone outputs (firing rates of dark3 / dark4 / dark5).  Nothing behind that point is a neuron, so there is no
    spike-flip excuse: EVERY logit [B, 1680, 7] of the eval forward (neck, towers, prediction convolutions, sigmoid, decode) must be
    within 1e-4 (relative + absolute) of the oracle's; in train mode (batch-statistics BatchNorm) every raw prediction map and the
    running statistics of every neck / head BatchNorm."""
    from eas_snn_amd import ops
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    H, W, B = 256, 320, 2
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(_exp_opts(dict(use_spike='True'), H, W))
    hip = exp.get_model()
    ref = model_ref.build_model(use_spike='True')
    rx = fill.ANN_KEYS['True']
    assert fill.procedural_fill_(hip, 2.0, ann_regex=rx) == fill.procedural_fill_(ref, 2.0, ann_regex=rx)
    hip.to(dev).train(train); ref.train(train)
    feats, raws = {}, {}
    hooks = [ref.backbone.backbone.register_forward_hook(lambda m, i, o: feats.update({k: v.detach() for k, v in o.items()}))]
    for k in range(3):
        for kind in ('reg', 'obj', 'cls'):
            hooks.append(getattr(ref.head, f'{kind}_preds')[k].register_forward_hook(
                lambda m, i, o, key=(kind, k): raws.__setitem__(key, o.detach())))
    x = torch.from_numpy(fill.poisson_events((B, 1, 4, 2, H, W), 0.5, seed=21))
    with torch.no_grad():
        if train:
            tg = torch.zeros(B, 50, 5)
            tg[:, 0] = torch.tensor([0, 0.3 * W, 0.4 * H, 0.25 * W, 0.3 * H])
            ref(x, tg)
        else:
            want = ref(x).numpy()
    for h in hooks:
        h.remove()
    sj_ref.reset_net(ref)
    rates = [feats[f].mean(0).to(dev) for f in ('dark3', 'dark4', 'dark5')]        # spiking_yolo_pafpn.py:98
    assert all(0.0 < float(r.mean()) < 1.0 for r in rates), 'the oracle backbone must fire for this test to mean something'
    hip.backbone._features = lambda _x: rates                                       # the neck of the HIP model on the oracle's firing rates
    with torch.no_grad(), ops.packed_weights(hip):
        fpn = hip.backbone(None)
        if not train:
            got = hip.head(fpn).cpu().numpy()
        else:
            got_raws = [hip.head._level(k, hip.head._prepare(f)) for k, f in enumerate(fpn)]
    del hip.backbone._features
    functional.reset_net(hip)
    if not train:
        assert got.shape == want.shape == (B, 1680, 7)
        err = np.abs(got - want) / (RTOL * np.abs(want) + RTOL)
        print(f'neck + head on the oracle firing rates, eval: worst logit at {err.max():.3f} of the 1e-4 tolerance ({got.size} logits)')
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=RTOL)
        return
    worst = 0.0
    for k in range(3):
        for j, kind in enumerate(('reg', 'obj', 'cls')):
            g, w = got_raws[k][j].cpu().numpy(), raws[(kind, k)].numpy()
            worst = max(worst, float((np.abs(g - w) / (RTOL * np.abs(w) + RTOL)).max()))
            np.testing.assert_allclose(g, w, rtol=RTOL, atol=RTOL, err_msg=f'{kind}[{k}]')
    nbn = 0
    rbufs = dict(ref.named_buffers())
    for n, b in hip.named_buffers():
        if (n.startswith('head.') or (n.startswith('backbone.') and not n.startswith('backbone.backbone.'))) and n.endswith(('running_mean', 'running_var')):
            np.testing.assert_allclose(b.cpu().numpy(), rbufs[n].numpy(), rtol=1e-4, atol=1e-6, err_msg=n)
            nbn += 1
    assert nbn == 2 * 39          # 24 neck + 15 head BatchNorm layers
    print(f'neck + head on the oracle firing rates, train-mode BN: worst raw prediction at {worst:.3f} of the 1e-4 tolerance, {nbn} running statistics')


def _grad_close(got, want, what, rtol=5e-4, atol_rel=1e-5, mask=None, extra_atol=0.0):
    """elementwise: |got - want| <= rtol * |want| + atol_rel * max|want| (+ extra); returns the worst element in units of that bound"""
    got, want = got.astype(np.float64), want.astype(np.float64)
    bound = rtol * np.abs(want) + atol_rel * max(float(np.abs(want).max()), 1e-30) + extra_atol
    ratio = np.abs(got - want) / bound
    if mask is not None:
        ratio = ratio[mask]
    worst = float(ratio.max()) if ratio.size else 0.0
    assert worst <= 1.0, f'{what}: worst element {worst:.2f} x the tolerance (rtol {rtol}, atol {atol_rel} of max |g|)'
    return worst


def _teacher_forced_backward(dev, exp_name, cfg, shape, expect_layers):
    """Backward of every spiking conv -> BN -> PLIF block in isolation, train mode (batch-statistics BatchNorm): the oracle model runs a
    whole training step (forward, loss, backward) on the CPU; for every block its input x and the gradient that reached its output are
    recorded, then the oracle block and the HIP block are each run on exactly that (x, grad_out): grad_x, grad of the convolution
    weight, of gamma / beta and of the PLIF w are compared ELEMENTWISE (5e-4 relative + 1e-5 of the tensor's largest value; measured
    on MI355X: the worst element of any block sits at 0.3 of that bound).  Where a
    spike flipped (potential within rounding of the threshold), the 3x3 neighbourhood of the flipped pixels is left out of grad_x
    and the parameter gradients get the flipped neurons' worth of slack."""
    from eas_snn_amd import ops
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    H, W = shape[-2:]
    exp = get_exp(None, exp_name)
    exp.merge(_exp_opts(cfg, H, W))
    hip = exp.get_model()
    ref = model_ref.build_model(**cfg)
    rx = fill.ANN_KEYS[cfg['use_spike']]
    assert fill.procedural_fill_(hip, 2.0, ann_regex=rx) == fill.procedural_fill_(ref, 2.0, ann_regex=rx)
    hip.to(dev).train(); ref.train()
    ref.head.use_l1 = True
    names = [n for n, m in ref.named_modules() if isinstance(m, model_ref.BaseConv) and isinstance(getattr(m, 'act', None), sj_ref.BaseNode)]
    rec = {}

    def mk(name):
        def f(mod, inp, out):
            rec[name] = [inp[0].detach().clone(), None]
            out.register_hook(lambda g: rec[name].__setitem__(1, g.detach().clone()))
        return f
    hooks = [ref.get_submodule(n).register_forward_hook(mk(n)) for n in names]
    x = torch.from_numpy(fill.poisson_events(shape, 0.5, seed=5))
    tg = torch.zeros(shape[0], 50, 5)
    tg[:, 0] = torch.tensor([0, 0.3 * W, 0.4 * H, 0.25 * W, 0.3 * H])
    tg[:, 1] = torch.tensor([1, 0.7 * W, 0.6 * H, 0.2 * W, 0.35 * H])
    ref(x, tg)['total_loss'].backward()
    for h in hooks:
        h.remove()
    sj_ref.reset_net(ref)
    ref.zero_grad(set_to_none=True)
    assert len(rec) == expect_layers and all(v[1] is not None for v in rec.values())
    stats = dict(layers=0, flips=0, worst=0.0, worst_name='')
    for name in names:
        x_in, g_out = rec.pop(name)
        rb, hb = ref.get_submodule(name), hip.get_submodule(name)
        xr = x_in.clone().requires_grad_(True)
        out_r = rb(xr)
        out_r.backward(g_out)
        sj_ref.reset_net(rb)
        xh = x_in.to(dev)
        if float(x_in.abs().max()) <= 16 and bool((x_in == x_in.round()).all()):
            ops.mark_small_int(xh)
        xh.requires_grad_(True)
        out_h = hb(xh)
        out_h = out_h[0] if isinstance(out_h, tuple) else out_h
        out_h.backward(g_out.to(dev))
        functional.reset_net(hb)
        flips = (out_h.detach().cpu() != out_r.detach())
        nflip = int(flips.sum())
        assert nflip <= 4e-5 * flips.numel() + 1, f'{name}: {nflip} spike flips'
        mask, slack = None, 0.0
        if nflip:
            # grad_x: a flipped output pixel reaches the input pixels under the kernel (and, at stride 2, their up-sampled positions);
            # leave out a generous neighbourhood at every time step and channel
            bad = flips.any(dim=2, keepdim=True).any(dim=0, keepdim=True).float()               # [1,N,1,Ho,Wo]
            s_ = x_in.shape[-1] // out_r.shape[-1]
            bad = torch.nn.functional.max_pool2d(bad[0], 5, 1, 2)
            bad = torch.nn.functional.interpolate(bad, scale_factor=s_, mode='nearest') if s_ > 1 else bad
            bad = torch.nn.functional.max_pool2d(bad, 2 * s_ + 1, 1, s_)
            mask = np.broadcast_to(~bad.bool().numpy()[None], tuple(x_in.shape))
            slack = nflip * float(g_out.abs().max()) * 4.0
        w = _grad_close(xh.grad.cpu().numpy(), xr.grad.numpy(), f'{name} grad_x', mask=mask, extra_atol=1e-3 * float(xr.grad.abs().max()) if nflip else 0.0)
        pr = dict(rb.named_parameters())
        for pn, p in hb.named_parameters():
            assert p.grad is not None and pr[pn].grad is not None, f'{name}.{pn}'
            scale = float(x_in.abs().max()) if 'conv' in pn else 1.0
            w = max(w, _grad_close(p.grad.cpu().numpy(), pr[pn].grad.numpy(), f'{name}.{pn}', extra_atol=slack * scale))
        hb.zero_grad(set_to_none=True); rb.zero_grad(set_to_none=True)
        stats['layers'] += 1; stats['flips'] += nflip
        if w > stats['worst']:
            stats['worst'], stats['worst_name'] = w, name
    print(f"teacher-forced backward {exp_name} {cfg.get('use_spike')} T={cfg.get('T', 3)} {H}x{W}: {stats['layers']} blocks, {stats['flips']} spike flips, "
          f"worst gradient element at {stats['worst']:.3f} of the tolerance ({stats['worst_name']})")
    return stats


# (_teacher_forced_backward runs for all four BASELINE configurations at their canvases in tests/test_gpu_bench_shapes.py)
M_WIDTH = dict(depth=0.67, width=0.75)
FULL_CANVAS_CASES = {
    # BASELINE configs[2]: SYOLOX-M, full_spike_v2, T=5, RPD, 256x320
    'cfg3_m_t5_256x320': ('e-yolox-m', dict(M_WIDTH, use_spike='full_spike_v2', T=5, Tm=4, num_classes=2), (1, 1, 4, 2, 256, 320)),
    # configs[3]: SYOLOX-M, 1 Mpx stacked-histogram input summed to 2 channels, canvas 384x640, T=3, 3 classes
    'cfg4_m_1mpx_384x640': ('e-yolox-m', dict(M_WIDTH, use_spike='full_spike_v2', T=3, Tm=4, num_classes=3), (1, 1, 4, 2, 384, 640)),
    # configs[4]: SYOLOX-M, N-Caltech101 180x240 -> canvas 192x256, 100 classes, T=7, Tm=8, Ts=7 (long-sequence adaptive sampling)
    'cfg5_m_ncaltech_192x256': ('e-yolox-m', dict(M_WIDTH, use_spike='full_spike_v2', T=7, Tm=8, Ts=7, num_classes=100, alpha=1.5),
                                (1, 1, 8, 2, 192, 256)),
}


@pytest.mark.parametrize('name', sorted(FULL_CANVAS_CASES))
def test_layerwise_teacher_forced_parity_m_width_full_canvas(dev, name):
    """BASELINE configs[2..4] at SYOLOX-M width and the full canvas (B = 1): every spiking block of backbone, neck and head
    teacher-forced against the torch-CPU oracle (flip fraction, membrane potentials), then the whole HIP model on the same input:
    finite logits of the right shape, bit-identical from run to run."""
    from oracle import fill
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    exp_name, cfg, shape = FULL_CANVAS_CASES[name]
    st = _teacher_forced(dev, exp_name, cfg, shape, train=False)
    assert st['layers'] == 97                                    # SYOLOX-M, full_spike_v2: 97 spiking BaseConv blocks (backbone, neck, head)
    H, W = shape[-2:]
    exp = get_exp(None, exp_name)
    exp.merge(_exp_opts(cfg, H, W))
    hip = exp.get_model()
    fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS[cfg['use_spike']])
    hip.to(dev).eval()
    x = torch.from_numpy(fill.poisson_events(shape, 0.5, seed=3)).to(dev)
    with torch.no_grad():
        a = hip(x).clone(); functional.reset_net(hip)
        b = hip(x).clone(); functional.reset_net(hip)
    A = sum((H // s) * (W // s) for s in (8, 16, 32))
    assert a.shape == (1, A, 5 + cfg['num_classes']) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b), f'{name}: eval forward differs between two runs by {float((a - b).abs().max()):.3g}'


def test_sampler_parity_at_256x320(dev):
    """The adaptive sampler alone at the benchmark canvas (B=2, Tm=4, README flags) against the oracle: outputs within 1e-4 except
    elements whose potential crossed the threshold within convolution rounding; eval and gradient (input + all 8 parameters)."""
    from oracle import fill, model_ref, sj_ref
    from yolox.exp import get_exp
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True'])
    hip = exp.get_model()
    ref = model_ref.build_model(use_spike='True')
    assert fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS['True']) == fill.procedural_fill_(ref, 2.0, ann_regex=fill.ANN_KEYS['True'])
    he, re_ = hip.embedding.to(dev), ref.embedding
    xn = fill.poisson_events((2, 1, 4, 2, 256, 320), 0.5, seed=9)
    xr = torch.from_numpy(xn).requires_grad_(True)
    xh = torch.from_numpy(xn).to(dev).requires_grad_(True)
    out_r = re_(xr)
    out_h = he(xh)
    got, want = out_h.detach().cpu().numpy(), out_r.detach().numpy()
    assert got.shape == want.shape
    bad = ~np.isclose(got, want, rtol=RTOL, atol=1e-5)
    print(f'sampler 256x320: {int(bad.sum())} of {bad.size} output elements differ (threshold crossings)')
    assert bad.mean() < 2e-5
    gout = torch.from_numpy(np.random.default_rng(2).standard_normal(want.shape).astype(np.float32))
    out_r.backward(gout)
    out_h.backward(gout.to(dev))
    mism = ~np.isclose(xh.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-3, atol=1e-3)
    assert mism.mean() <= bad.mean() * 33 * 33 * 8 + 1e-7, f'{mism.mean():.2e} of the input gradient differs'
    pr = dict(re_.named_parameters())
    for n, p in he.named_parameters():
        a, b = p.grad.cpu().numpy().astype(np.float64), pr[n].grad.numpy().astype(np.float64)
        rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12)
        assert rel < 2e-3 + 50 * bad.mean(), f'{n}: {rel:.2e}'


MODELS['model_m_fullv2_t5_64x96_train'] = ('e-yolox-m', ['use_spike', 'full_spike_v2', 'T', '5'])
USE_SPIKE['model_m_fullv2_t5_64x96_train'] = 'full_spike_v2'


TRAIN_STEP_FIXTURES = ['model_s_true_64', 'model_s_fullv2_64', 'model_s_true_patan_64', 'model_m_fullv2_t5_64x96_train']


def _train_step_figures(dev, name):
    """one training step of the fixture's model on the HIP path against the fixture (the unmodified reference classes' step): the achieved
    figures -- relative error of every loss term, of every gradient norm, share of the stored gradient elements within 1e-3 relative + 2e-5
    of the tensor's maximum (tests/parity_report.py prints them; test_model_train_step_golden asserts on them)"""
    from spikingjelly.activation_based import functional
    g, model = _build(name, dev)
    model.train()
    model.head.use_l1 = True
    out = model(torch.from_numpy(g['x']).to(dev), torch.from_numpy(g['targets']).to(dev))
    out['total_loss'].backward()
    functional.reset_net(model)
    fig = {'loss_rel_err': {k: abs(float(out[k]) - float(g[f'loss/{k}'])) / max(abs(float(g[f'loss/{k}'])), 1e-12)
                            for k in ('total_loss', 'iou_loss', 'l1_loss', 'conf_loss', 'cls_loss')},
           'num_fg': (float(out['num_fg']), float(g['loss/num_fg']))}
    norms = dict(zip([str(s) for s in g['gradnorm_keys']], g['gradnorm_vals']))
    rel = []
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        rel.append(abs(float(p.grad.norm()) - norms[n]) / (norms[n] + 1e-6))
    rel = np.array(rel)
    fig.update(gradnorm_rel_err_median=float(np.median(rel)), gradnorm_rel_err_max=float(rel.max()), gradnorm_within_5pct_share=float((rel < 5e-2).mean()),
               parameters=len(rel))
    params = dict(model.named_parameters())
    ok = tot = tensors = exact_tensors = 0
    for key in g.files:
        if key.startswith('grad/'):
            n = key[5:]
            got = params[n].grad.detach().cpu().numpy()
        elif key.startswith('gradsample/'):
            n = key[11:]
            flat = params[n].grad.detach().reshape(-1)
            got = flat[::flat.numel() // 1024][:1024].cpu().numpy()
        else:
            continue
        want = g[key]
        assert got.shape == want.shape, key
        close = np.abs(got - want) <= 1e-3 * np.abs(want) + 2e-5 * max(float(np.abs(want).max()), 1e-30)
        ok += int(close.sum()); tot += close.size; tensors += 1; exact_tensors += int(close.all())
    fig.update(gradient_elements=tot, gradient_elements_within_tol_share=ok / max(tot, 1), gradient_tensors=tensors,
               gradient_tensors_entirely_within=exact_tensors)
    return fig


@pytest.mark.parametrize('name', TRAIN_STEP_FIXTURES)
def test_model_train_step_golden(dev, name):
    """One training step (loss terms, every parameter gradient) against the UNMODIFIED reference classes' step on the same input
    (tests/golden/model_*.npz): losses, the norm of every gradient, and -- elementwise -- every gradient the fixture stores in full
    (``grad/``: PLIF w, sampler, BN gamma, small tensors) or as a strided sample of <= 1024 elements (``gradsample/``, the M fixture:
    SYOLOX-M full_spike_v2 T=5, BASELINE configs[2]'s model).  End to end a single rounding-level spike flip in the forward changes
    every gradient behind it, so the elementwise bar is statistical here (the share of elements within
    1e-3 relative + 2e-5 of the tensor's maximum); the per-block elementwise bar without that caveat is
    test_layerwise_teacher_forced_backward_*.  The achieved figures are printed here and collected by tests/parity_report.py."""
    f = _train_step_figures(dev, name)
    print(f'{name}: {f}')
    assert all(v <= 5e-3 for v in f['loss_rel_err'].values()), f['loss_rel_err']
    assert abs(f['num_fg'][0] - f['num_fg'][1]) <= 1e-6 * abs(f['num_fg'][1])
    assert f['gradnorm_rel_err_median'] < 1e-3 and f['gradnorm_within_5pct_share'] > 0.95
    assert f['gradient_tensors'] > 30
    assert f['gradient_elements_within_tol_share'] > 0.97 and f['gradient_tensors_entirely_within'] / f['gradient_tensors'] > 0.8


@pytest.mark.parametrize('name', ['model_s_true_64', 'model_s_fullv2_64', 'model_s_true_256x320'])
def test_train_step_is_bit_reproducible(dev, name):
    """Two runs of the same training step from the same state give the same loss and the same gradient of EVERY parameter, bit for
    bit, with the library's defaults: every convolution (forward, input and weight gradient) is an own kernel and every reduction
    (split-K slabs, block partials, BN sums, loss terms) is summed in a fixed order; nothing accumulates with float atomics."""
    from spikingjelly.activation_based import functional
    g, model = _build(name, dev)
    model.train()
    model.head.use_l1 = True
    x = torch.from_numpy(g['x']).to(dev)
    if 'targets' in g:
        tg = torch.from_numpy(g['targets']).to(dev)
    else:                                   # the 256x320 fixture holds eval logits only: two boxes per sample, as bench.py draws them
        from eas_snn_amd import data
        tg = data.synth_targets(x.shape[0], tuple(x.shape[-2:]), dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for _ in range(3):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        out = model(x, tg)
        out['total_loss'].backward()
        functional.reset_net(model)
        runs.append((out['total_loss'].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][0], runs[2][0])
    diff = sorted({n for r in runs[1:] for n in r[1] if not torch.equal(runs[0][1][n], r[1][n])})
    assert not diff, f'run-to-run gradient differences in {diff}'


@pytest.mark.parametrize('name', ['model_s_true_64', 'model_s_fullv2_64', 'model_s_true_256x320'])
def test_spike_planes_and_fp32_spikes_give_bit_identical_steps(dev, name, monkeypatch):
    """The spiking backbone with its spike tensors kept as bf16 spike planes (the default, ops "SPIKE PLANES") against the same step with
    fp32 spike tensors (EAS_SPIKE_PLANES=0): loss, every parameter gradient and every BatchNorm buffer bit-identical -- the planes hold
    exactly the values of the fp32 tensors and every kernel that reads them adds the same products in the same order."""
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    from yolox.models.network_blocks import BaseConv
    g, model = _build(name, dev)
    model.train()
    model.head.use_l1 = True
    x = torch.from_numpy(g['x']).to(dev)
    if 'targets' in g:
        tg = torch.from_numpy(g['targets']).to(dev)
    else:
        from eas_snn_amd import data
        tg = data.synth_targets(x.shape[0], tuple(x.shape[-2:]), dev)
    nplanes = sum(1 for m in model.modules() if isinstance(m, BaseConv) and m.planes_out)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for planes in (True, False):
        monkeypatch.setattr(ops, 'SPIKE_PLANES', planes)
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        try:
            out = model(x, tg)
            out['total_loss'].backward()
        finally:
            ops.set_timer(None)
        functional.reset_net(model)
        torch.cuda.synchronize()
        unpacks = timer.summary().get('eas_spike_planes', {}).get('calls', 0)
        runs.append((out['total_loss'].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()},
                     {n: b.clone() for n, b in model.named_buffers()}, unpacks))
    assert nplanes > 20                                      # every converted block hands planes on
    if '256x320' in name:
        assert runs[0][3] == 0, runs[0][3]                   # no unpack at all: the SPP pooling block runs on the planes too
    assert runs[1][3] == 0
    assert torch.equal(runs[0][0], runs[1][0])
    diff = sorted(n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[1][1][n]))
    assert not diff, f'gradient differences in {diff[:5]}'
    diff = sorted(n for n in runs[0][2] if not torch.equal(runs[0][2][n], runs[1][2][n]))
    assert not diff, f'buffer differences in {diff[:5]}'


def test_train_step_split_backward_buckets_and_graph_replay_are_bit_identical(dev):
    """yolox.core.trainer.TrainStep (the iteration tools/train_event.py runs and bench.py measures): the backward pass split at the
    backbone with the gradients packed into two flat buckets (the N > 1 form, here without a process group), eagerly and as three
    HIP-graph replays, against the plain one-backward step: every parameter gradient bit-identical."""
    from eas_snn_amd import ops
    from eas_snn_amd.parallel import BucketedGradAllReduce
    from yolox.core.trainer import DEFAULT_CUT, DEFAULT_LOWER, TrainStep
    g, model = _build('model_s_true_64', dev)
    model.train()
    model.head.use_l1 = True
    x, tg = torch.from_numpy(g['x']).to(dev), torch.from_numpy(g['targets']).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    torch.cuda.set_stream(torch.cuda.Stream())          # capture needs a non-default stream for the AccumulateGrad nodes
    try:
        plain = TrainStep(model, opt, lambda: (x, tg))
        plain()
        want = {n: p.grad.clone() for n, p in model.named_parameters()}
        loss0 = plain.loss.detach().clone()
        ex = BucketedGradAllReduce(model, split=DEFAULT_LOWER, broadcast_parameters=False)
        assert ex.nbuckets == 2
        step = TrainStep(model, opt, lambda: (x, tg), exchange=ex, cut=DEFAULT_CUT)
        assert step.cut == DEFAULT_CUT
        # the cut is a forward hook: it must not switch the model to fp32 spikes (N > 1 ranks would run another kernel set than one rank)
        from eas_snn_amd import ops_bn                   # (the BN + LIF operators allocate the planes they write: watch the name THEY call)
        made, new_planes = [], ops_bn.new_planes
        ops_bn.new_planes = lambda *a, **k: (made.append(1), new_planes(*a, **k))[1]
        try:
            step()
        finally:
            ops_bn.new_planes = new_planes
        assert made, 'the split step ran without spike planes'
        assert torch.equal(step.loss.detach(), loss0)
        bad = [n for n, p in model.named_parameters() if not torch.equal(p.grad, want[n])]
        assert not bad, f'split backward + buckets differ from the plain backward in {bad[:5]}'
        assert all(p.grad.data_ptr() == v.data_ptr() for ps, vs in zip(ex.buckets, ex.views) for p, v in zip(ps, vs))   # attached views
        verify, ops.VERIFY_SMALL_INT = ops.VERIFY_SMALL_INT, False      # the tag check reads the device: not inside a capture
        try:
            launch = step.capture(warm=1)
        finally:
            ops.VERIFY_SMALL_INT = verify
        assert 'three hip-graph replays' in launch
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        bad = [n for n, p in model.named_parameters() if not torch.equal(p.grad, want[n])]
        assert not bad, f'graph replays differ from the plain backward in {bad[:5]}'
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())


def test_trainer_loop_on_the_gpu_graph_replay_equals_eager(dev, tmp_path, monkeypatch):
    """yolox.core.Trainer.train() -- the loop tools/train_event.py runs -- for one epoch of five iterations on the synthetic loader: the
    step recorded on the first batch (warm-up launches, capture, one replay, then model / optimizer / weight-average state put back:
    ``capture(restore=True)``) and five replays (the default) against the SAME loop with the capture replaced by its bookkeeping only
    (device-side Adam step counters and learning rate, as a captured step needs them), i.e. five eager iterations of the same arithmetic:
    same learning-rate schedule, and the losses, parameters, BatchNorm buffers and EMA weights (made by the optimizer's launch in both) are
    bit-identical -- so recording leaves no trace.  EAS_TRAIN_GRAPH=0 (host-side learning rate: Adam rounds differently) must run too."""
    import types
    from eas_snn_amd import data, ops
    from yolox.exp import get_exp
    monkeypatch.setattr(ops, 'VERIFY_SMALL_INT', False)      # the suite's tag check reads the device: not inside a capture
    from yolox.core.trainer import TrainStep
    results = []
    for graph in ('1', 'bookkeeping', '0'):
        monkeypatch.setenv('EAS_TRAIN_GRAPH', '0' if graph == '0' else '1')
        if graph == 'bookkeeping':
            # capture() = bookkeeping + recording (executes nothing) + ONE replay, which is the batch's training step
            # (the Trainer records on its first batch with restore=True, then runs that batch's step: the bookkeeping twin records nothing)
            monkeypatch.setattr(TrainStep, 'capture', lambda self, warm=3, restore=False: self.make_capturable())
        exp = get_exp(None, 'e-yolox-s')
        exp.merge(BASE_OPTS + ['use_spike', 'True', 'input_size', '(64, 64)', 'test_size', '(64, 64)'])    # (a 2x3 stride-32 map would take the library convolution: not reproducible)
        exp.max_epoch, exp.print_interval, exp.output_dir = 1, 1, str(tmp_path / graph)
        exp.get_data_loader = lambda batch_size, is_distributed, no_aug=False, cache_img=None, exp=exp: \
            data.SyntheticEventLoader(exp, batch_size, iters=5, n_events=3000, sensor_hw=(60, 60))
        torch.manual_seed(5)
        tr = exp.get_trainer(types.SimpleNamespace(batch_size=4, fp16=False, experiment_name='loop', ckpt=None, resume=False))
        tr.train()
        assert (tr.step.graphs is not None) == (graph == '1') and len(tr.log) == 5
        assert all(np.isfinite(r['loss']) for r in tr.log)
        if tr.use_model_ema and getattr(tr.ema_model, '_fused_in', None) is tr.optimizer:
            # five iterations = five updates of the weight average, on the host mirror and on the device counter that drives the decay ramp
            # (recording and its warm-up launches leave no trace in either)
            assert tr.ema_model.updates == 5 == tr.optimizer.ema_updates_on_device(), (tr.ema_model.updates, tr.optimizer.ema_updates_on_device())
        if graph == '0':
            continue
        sd = {k: v.clone() for k, v in tr.bare_model.state_dict().items()}
        ema = {k: v.clone() for k, v in tr.ema_model.ema.state_dict().items()} if tr.use_model_ema else {}
        results.append((sd, ema, [r['loss'] for r in tr.log], [r['lr'] for r in tr.log]))
        torch.cuda.set_stream(torch.cuda.default_stream())
    (sd1, ema1, loss1, lr1), (sd0, ema0, loss0, lr0) = results
    assert lr1 == lr0
    assert loss1 == loss0, (loss1, loss0)
    bad = [k for k in sd1 if not torch.equal(sd1[k], sd0[k])]
    assert not bad, bad[:5]
    bad = [k for k in ema1 if not torch.equal(ema1[k], ema0[k])]
    assert not bad, bad[:5]


def test_trainer_resume_on_the_gpu_continues_optimizer_and_average(dev, tmp_path, monkeypatch):
    """``--resume`` (trainer.py:318-338 of the reference) through the graph-replayed loop: a run of two epochs x three iterations that is
    interrupted after the first epoch and resumed from ``latest_ckpt.pth``.  The resumed Trainer records its step on its first batch with
    model / optimizer / weight-average state put back afterwards (``capture(restore=True)``) -- which must put back the LOADED optimizer
    state, not Adam's zeros: after the resumed epoch every Adam step counter reads 6 (3 + 3), the second moments are those of six steps
    (non-zero, larger than after three), the weight average's counter continues from max_iter * start_epoch (the reference's rule,
    trainer.py:169-171), the learning-rate schedule continues, and the checkpoint names the next epoch."""
    import types
    from eas_snn_amd import data, ops
    from yolox.exp import get_exp
    monkeypatch.setattr(ops, 'VERIFY_SMALL_INT', False)      # the suite's tag check reads the device: not inside a capture

    def trainer(resume, epochs):
        exp = get_exp(None, 'e-yolox-s')
        exp.merge(BASE_OPTS + ['use_spike', 'True', 'input_size', '(64, 64)', 'test_size', '(64, 64)'])
        exp.max_epoch, exp.print_interval, exp.output_dir, exp.eval_interval = epochs, 1, str(tmp_path), 0
        exp.get_data_loader = lambda batch_size, is_distributed, no_aug=False, cache_img=None, exp=exp: \
            data.SyntheticEventLoader(exp, batch_size, iters=3, n_events=3000, sensor_hw=(60, 60))
        torch.manual_seed(5)
        return exp.get_trainer(types.SimpleNamespace(batch_size=4, fp16=False, experiment_name='resume', ckpt=None, resume=resume))
    first = trainer(False, 1)
    first.train()
    assert first.step.graphs is not None and len(first.log) == 3
    steps1 = {float(st_['step']) for st_ in first.optimizer.state.values()}
    v1 = sum(float(st_['exp_avg_sq'].sum()) for st_ in first.optimizer.state.values())
    assert steps1 == {3.0} and first.ema_model.updates == 3 == first.optimizer.ema_updates_on_device()
    ck = torch.load(os.path.join(str(tmp_path), 'resume', 'latest_ckpt.pth'), map_location='cpu')
    assert ck['start_epoch'] == 1
    torch.cuda.set_stream(torch.cuda.default_stream())
    second = trainer(True, 2)
    second.train()
    assert second.start_epoch == 1 and second.step.graphs is not None and len(second.log) == 3
    steps2 = {float(st_['step']) for st_ in second.optimizer.state.values()}
    v2 = sum(float(st_['exp_avg_sq'].sum()) for st_ in second.optimizer.state.values())
    assert steps2 == {6.0}, f'Adam step counters after the resumed epoch: {sorted(steps2)} (recording must put back the loaded state)'
    assert v2 > v1 > 0
    # the weight average: restarted from the checkpoint's (averaged) weights with the counter at max_iter * start_epoch = 3, three more updates
    assert second.ema_model.updates == 6 == second.optimizer.ema_updates_on_device()
    assert [r['epoch'] for r in second.log] == [1, 1, 1] and all(np.isfinite(r['loss']) for r in second.log)
    lr1, lr2 = [r['lr'] for r in first.log], [r['lr'] for r in second.log]
    assert lr1 != lr2 or len(set(lr1)) == 1           # the schedule moved on (or is flat by configuration)
    assert torch.load(os.path.join(str(tmp_path), 'resume', 'latest_ckpt.pth'), map_location='cpu')['start_epoch'] == 2
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_trainer_loop_with_sgd_stays_eager(dev, tmp_path, monkeypatch):
    """``optimizer SGD`` (the reference's other branch, event_yolox_base.py:361-377): its step takes the learning rate as a host number, so
    the Trainer must not capture it -- the loop runs eagerly, with a float lr, and trains"""
    import types
    from eas_snn_amd import data
    from yolox.exp import get_exp
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True', 'input_size', '(64, 64)', 'test_size', '(64, 64)', 'optimizer', 'SGD'])
    exp.max_epoch, exp.print_interval, exp.output_dir = 1, 1, str(tmp_path)
    exp.get_data_loader = lambda batch_size, is_distributed, no_aug=False, cache_img=None: \
        data.SyntheticEventLoader(exp, batch_size, iters=4, n_events=3000, sensor_hw=(60, 60))
    torch.manual_seed(5)
    tr = exp.get_trainer(types.SimpleNamespace(batch_size=4, fp16=False, experiment_name='sgd', ckpt=None, resume=False))
    before = None
    tr.train()
    assert isinstance(tr.optimizer, torch.optim.SGD) and not tr.use_graph and tr.step.graphs is None
    assert all(isinstance(g['lr'], float) for g in tr.optimizer.param_groups)
    assert len(tr.log) == 4 and all(np.isfinite(r['loss']) for r in tr.log)
    torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.mark.parametrize('train', [False, True])
def test_forward_hooks_on_container_batchnorm_and_neuron_fire_on_every_path(dev, train):
    """ADVICE r4 / VERDICT r5 next #8: the fused paths stand in for module calls -- eas_conv_bn_lif_eval for (container, BatchNorm, neuron), the
    BN + LIF kernel pair for (BatchNorm, neuron).  A forward hook on ANY of the three must keep firing: the block then takes the
    module-by-module path.  Hooks on the convolution container, the BatchNorm and the neuron of every converted block of SYOLOX-S: each fires
    exactly once per forward, eval and train mode, and the result stays the model's (logits close to the un-hooked forward; the
    module-by-module BatchNorm rounds differently from the fused kernel, so not bit-identical)."""
    import collections
    from oracle import fill
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    from yolox.models.network_blocks import BaseConv
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True', 'input_size', '(64, 64)', 'test_size', '(64, 64)'])
    model = exp.get_model()
    fill.procedural_fill_(model, 2.0, ann_regex=fill.ANN_KEYS['True'])
    model.to(dev).train(train)
    model.head.use_l1 = True
    x = torch.from_numpy(fill.poisson_events((2, 1, 4, 2, 64, 64), 0.5, seed=3)).to(dev)
    tg = torch.zeros(2, 50, 5, device=dev)
    tg[:, 0] = torch.tensor([0, 20.0, 24.0, 16.0, 20.0])

    def run():
        with torch.no_grad():
            out = model(x, tg)['total_loss'] if train else model(x)
        functional.reset_net(model)
        return out.float().cpu()
    plain = run()
    fired = collections.Counter()
    handles, watched = [], []
    for name, m in model.named_modules():
        if isinstance(m, BaseConv) and m.spiking():
            for part in ('conv', 'bn', 'act'):
                key = f'{name}.{part}'
                watched.append(key)
                handles.append(getattr(m, part).register_forward_hook(lambda mod, i, o, key=key: fired.update([key])))
    assert len(watched) == 3 * 34
    hooked = run()
    for h in handles:
        h.remove()
    silent = [k for k in watched if fired[k] != 1]
    assert not silent, f'hooks that did not fire exactly once: {silent[:6]} ({len(silent)} of {len(watched)})'
    assert torch.isfinite(hooked).all()
    if train:
        assert abs(float(hooked) - float(plain)) <= 0.05 * abs(float(plain)) + 1e-3
    else:
        assert float(np.isclose(hooked.numpy(), plain.numpy(), rtol=1e-3, atol=1e-3).mean()) >= 0.9
    again = run()                                    # hooks gone: the fused paths are back, bit for bit
    assert torch.equal(again, plain)


def test_evaluator_on_the_gpu(dev, tmp_path):
    """yolox.evaluators.EventEvaluator (what tools/eval_event.py:209-211 calls) on the GPU: (1) the default form -- two HIP-graph replays
    per batch -- returns the same detections as eager launches; (2) they equal the checker's post-processing (oracle/postprocess_ref.py)
    of the model's own logits; (3) fed the ORACLE's logits (the torch-CPU model behind a stand-in module), every detection equals
    the checker's post-processing of those logits."""
    from oracle import fill, model_ref, postprocess_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True', 'input_size', '(64, 128)', 'test_size', '(64, 128)', 'test_conf', '0.00003', 'nmsthre', '0.5'])
    exp.eval_samples, exp.eval_events, exp.eval_sensor_hw, exp.output_dir = 10, 3000, (60, 100), str(tmp_path)   # (64x128: every map has an even width -> own kernels only, reproducible)
    model = exp.get_model()
    ref = model_ref.build_model(use_spike='True')
    assert fill.procedural_fill_(model, 2.0, ann_regex=fill.ANN_KEYS['True']) == fill.procedural_fill_(ref, 2.0, ann_regex=fill.ANN_KEYS['True'])
    model.to(dev).eval()
    ref.eval()
    evaluator = exp.get_evaluator(2, False)                       # batch 4: batches of 4, 4, 2 samples
    assert len(evaluator.dataloader) == 3 and evaluator.dataloader.batch_size == 4
    (_, _, summary), graphed = evaluator.evaluate(model, False, False, None, None, exp.test_size, return_outputs=True)
    assert len(evaluator._graphed) == 2 and 'Average forward time' in summary and float(evaluator.last_statistics[2]) == 2
    # the loader's last batch is short: the full and the short batch each keep their recording, a second evaluation records nothing new
    assert evaluator.graphs_recorded == 2
    (_, _, _), again = evaluator.evaluate(model, False, False, None, None, exp.test_size, return_outputs=True)
    assert evaluator.graphs_recorded == 2 and again == graphed
    evaluator.use_graph = False
    (_, _, _), eager = evaluator.evaluate(model, False, False, None, None, exp.test_size, return_outputs=True)
    assert sorted(graphed) == sorted(eager) and len(graphed) == 10
    for i in graphed:
        assert graphed[i] == eager[i], f'image {i}: graph replay and eager launches disagree'
    # (2) against the checker on the model's own logits
    scale = min(64 / 60.0, 128 / 100.0)
    n_det = 0
    for frames, labels, info, ids in evaluator.dataloader:
        with torch.no_grad():
            logits = model(frames)
        functional.reset_net(model)
        want = postprocess_ref.postprocess(logits.cpu().numpy(), 2, exp.test_conf, exp.nmsthre)
        for i, w in zip(ids.tolist(), want):
            assert (w is None) == (i not in eager)
            if w is not None:
                np.testing.assert_allclose(np.array(eager[i]['bboxes'], np.float32), w[:, :4] / np.float32(scale), rtol=1e-6)
                np.testing.assert_allclose(np.array(eager[i]['scores'], np.float32), w[:, 4] * w[:, 5], rtol=1e-6)
                assert eager[i]['categories'] == w[:, 6].astype(int).tolist()
                n_det += len(w)
    assert n_det >= 20, 'the confidence threshold left almost nothing to compare'

    # (3) the oracle's logits through the same loop
    class OracleLogits(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1, device=dev))     # tells the evaluator where the model lives
            self.seen = []

        def forward(self, x):
            with torch.no_grad():
                out = ref(x.cpu())
            sj_ref.reset_net(ref)
            self.seen.append(out.numpy().copy())
            return out.to(dev)

    stand_in = OracleLogits()
    (_, _, _), got = evaluator.evaluate(stand_in, False, False, None, None, exp.test_size, return_outputs=True)
    k = 0
    for (frames, labels, info, ids), logits in zip(evaluator.dataloader, stand_in.seen):
        want = postprocess_ref.postprocess(logits, 2, exp.test_conf, exp.nmsthre)
        for i, w in zip(ids.tolist(), want):
            assert (w is None) == (i not in got)
            if w is not None:
                np.testing.assert_allclose(np.array(got[i]['bboxes'], np.float32), w[:, :4] / np.float32(scale), rtol=1e-6)
                assert got[i]['categories'] == w[:, 6].astype(int).tolist()
                k += len(w)
    assert k >= 20


@pytest.mark.parametrize('name', ['model_s_true_64', 'model_s_fullv2_64', 'model_s_true_256x320', 'model_m_fullv2_t5_64x96'])
def test_eval_forward_with_the_fused_step_is_bit_identical(dev, name, monkeypatch):
    """The whole eval forward with every converted block as ONE conv -> BN -> LIF kernel (eas_conv_bn_lif_eval, the default in eval mode)
    against the two-kernel path (EAS_FUSED_EVAL=0): every logit bit-identical; the trace shows the fused entry point took the spiking
    blocks (the first spiking layer through the shared-input form, the CSPLayer branch pairs as one call)."""
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    g, model = _build(name, dev)
    model.eval()
    x = torch.from_numpy(g['x']).to(dev)
    outs, calls = [], []
    for fused in (True, False):
        monkeypatch.setattr(ops, 'FUSED_EVAL', 'all' if fused else False)
        with torch.no_grad(), ops.no_state_writeback(), ops.kernel_trace() as tr:
            outs.append(model(x).clone())
        functional.reset_net(model)
        calls.append([c[0] for c in tr.calls])
    nf = calls[0].count('eas_conv_bn_lif_eval')
    nb = sum(1 for n in calls[1] if n == 'eas_bn_lif_fwd_ex')
    print(f'{name}: {nf} fused conv->BN->LIF calls replace {nb} BN+LIF calls (+ their convolutions); '
          f'{sum(1 for n in calls[0] if n == "eas_bn_lif_fwd_ex")} layers stay on the two-kernel path')
    assert nf >= 0.8 * nb and 'eas_conv_bn_lif_eval' not in calls[1]
    assert torch.equal(outs[0], outs[1]), f'max |diff| {float((outs[0] - outs[1]).abs().max()):.3g}'


@pytest.mark.parametrize('name', ['model_s_true_64', 'model_s_false_64', 'model_s_true_256x320'])
def test_fuse_model_eval_forward(dev, name):
    """``yolox.utils.fuse_model`` (tools/eval_event.py --fuse, model_utils.py:35-80): every real-valued BaseConv's BatchNorm folded into its
    convolution, which then runs convolution + bias + SiLU as ONE kernel (eas_conv_bn_act_eval with unit constants / eas_conv_fwd_act).  Folding rounds the weights once more, so the
    logits are compared at 1e-5 of the tensor's scale, not bit for bit; the trace shows the activation epilogue took the folded blocks
    and no BatchNorm + SiLU kernel is left."""
    import copy
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    from yolox.utils import fuse_model
    g, model = _build(name, dev)
    model.eval()
    x = torch.from_numpy(g['x']).to(dev)
    with torch.no_grad():
        want = model(x).clone()
    functional.reset_net(model)
    fused = fuse_model(copy.deepcopy(model))
    with torch.no_grad(), ops.kernel_trace() as tr:
        got = fused(x).clone()
    functional.reset_net(fused)
    names = [c[0] for c in tr.calls]
    # folded blocks: conv + bias + SiLU as one kernel -- eas_conv_bn_act_eval with unit BatchNorm constants (real-valued inputs; keeps the
    # convolution pairs and in-place concatenations of the unfused model) or eas_conv_fwd_act (inputs tagged as small integers)
    nact = names.count('eas_conv_fwd_act') + names.count('eas_conv_bn_act_eval')
    assert nact >= 20 and 'eas_bn_silu_fwd_ex' not in names, (nact, sorted(set(names)))
    if ops.FUSED_ANN_EVAL:       # (EAS_FUSED_ANN_EVAL=0, development: eas_conv_fwd_act alone)
        assert names.count('eas_conv_bn_act_eval') >= 15, names.count('eas_conv_bn_act_eval')
    if name == 'model_s_true_256x320':
        # (a rounding-level change in front of the spiking backbone is amplified by spike flips at this size, DESIGN section 5: bulk statistics)
        assert abs(float(got[..., 4].median()) - float(want[..., 4].median())) < 0.05
        return
    err = float((got - want).abs().max() / want.abs().max())
    print(f'{name}: {nact} convolution + SiLU kernels; fuse_model vs BatchNorm kept: max |diff| {err:.2e} of the logit scale')
    frac = float(((got - want).abs() <= 1e-4 * want.abs() + 1e-4).float().mean())
    assert frac > 0.97, frac


@pytest.mark.parametrize('name', ['model_s_true_64', 'model_s_false_64', 'model_s_true_256x320'])
def test_eval_forward_with_the_fused_real_valued_blocks_is_bit_identical(dev, monkeypatch, name):
    """Eval mode: every real-valued BaseConv (the ANN PAFPN neck and head; the whole network of the use_spike=False model) runs
    conv -> BatchNorm (running statistics) -> SiLU as ONE kernel (eas_conv_bn_act_eval), including the pairs of convolutions that run as
    one (CSPLayer branches, head towers) and the in-place concatenations -- logits bit-identical to the convolution + eas_bn_silu_fwd_ex
    path (EAS_FUSED_ANN_EVAL=0), and no BatchNorm + SiLU kernel is left in the trace."""
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    g, model = _build(name, dev)
    model.eval()
    x = torch.from_numpy(g['x']).to(dev)
    outs = {}
    for fused in (False, True):
        monkeypatch.setattr(ops, 'FUSED_ANN_EVAL', fused)
        with torch.no_grad(), ops.kernel_trace() as tr:
            outs[fused] = model(x).clone()
        functional.reset_net(model)
        names = [c[0] for c in tr.calls]
        if fused:
            nf = names.count('eas_conv_bn_act_eval')
            assert nf >= 20 and 'eas_bn_silu_fwd_ex' not in names, (nf, sorted(set(names)))
        else:
            assert 'eas_conv_bn_act_eval' not in names and names.count('eas_bn_silu_fwd_ex') >= 20
    assert torch.equal(outs[True], outs[False]), float((outs[True] - outs[False]).abs().max())
    assert float(outs[True].abs().max()) > 0


def test_deferred_weight_gradient_reductions_are_bit_identical(dev):
    """ops.deferred_wgrad_reductions(): the slab reductions of all weight gradients of a backward pass in ONE launch at its end
    (eas_conv_wgrad_reduce_many) -- same loss, every parameter gradient bit-identical to the immediate reductions, also on a second
    pass that ACCUMULATES into existing gradients (where the operator must fall back to the immediate form by itself)."""
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    g, model = _build('model_s_true_64', dev)
    model.train()
    model.head.use_l1 = True
    x, tg = torch.from_numpy(g['x']).to(dev), torch.from_numpy(g['targets']).to(dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(deferred, passes):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        for _ in range(passes):
            out = model(x, tg)
            with ops.deferred_wgrad_reductions(deferred):
                out['total_loss'].backward()
            functional.reset_net(model)
        return {n: p.grad.clone() for n, p in model.named_parameters()}
    for passes in (1, 2):
        a, b = run(False, passes), run(True, passes)
        diff = [n for n in a if not torch.equal(a[n], b[n])]
        assert not diff, f'{passes} pass(es): {diff[:5]}'
    assert not ops._PENDING_REDUCE


def test_forward_after_fused_adam_step_uses_updated_weights(dev):
    """The packed MFMA weights must follow the optimizer: ``torch.optim.Adam(fused=True)`` (what EventExp.get_optimizer builds on
    the GPU) rewrites parameters without touching ``Tensor._version``.  Forward -> backward -> fused Adam steps -> forward again
    must equal (a) the same forward with every convolution packing its weight on the spot, bit for bit, and (b) the torch-CPU
    oracle model holding the updated state, and must differ from the logits before the update."""
    from eas_snn_amd import ops
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    from eas_snn_amd import data
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True'])
    exp.optimizer = 'ADAM'
    hip = exp.get_model()
    fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS['True'])
    hip.to(dev)
    opt = exp.get_optimizer(2)
    assert all(gr.get('fused') for gr in opt.param_groups), 'the GPU optimizer of the exp is the fused Adam'
    for gr in opt.param_groups:
        gr['lr'] = 3e-3                      # large enough that every weight visibly moves
    x = torch.from_numpy(fill.poisson_events((2, 1, 4, 2, 64, 128), 0.5, seed=11)).to(dev)
    tg = data.synth_targets(2, (64, 128), dev)
    hip.eval()
    with torch.no_grad():
        before = hip(x).clone()
    functional.reset_net(hip)
    hip.train()
    hip.head.use_l1 = True
    w0 = hip.backbone.backbone.dark3[0].conv[0].weight.detach().clone()
    for _ in range(2):
        out = hip(x, tg)
        opt.zero_grad(set_to_none=True)
        out['total_loss'].backward()
        opt.step()
        functional.reset_net(hip)
    assert not torch.equal(w0, hip.backbone.backbone.dark3[0].conv[0].weight)
    hip.eval()
    with torch.no_grad():
        after = hip(x).clone()
        functional.reset_net(hip)
        # (a) no packing scope: every convolution packs the weight tensor it is handed
        real_scope = ops.packed_weights
        try:
            import contextlib
            ops.packed_weights = lambda model: contextlib.nullcontext()
            unpacked = hip(x).clone()
        finally:
            ops.packed_weights = real_scope
        functional.reset_net(hip)
    # stale weights would reproduce `before` (the update moves the logits by O(1)); the two packing routes agree to rounding (the few
    # layers the library convolution still runs -- odd widths, 1/2/4-channel outputs -- are not bit-reproducible between calls)
    moved = float((after - before).abs().max())
    gap = float((after - unpacked).abs().max())
    print(f'logits moved by {moved:.3g} with the update; packed-scope vs per-call packing differ by {gap:.3g}')
    assert moved > 0.05 and gap < 1e-3 * max(1.0, moved), 'forward after the optimizer step used stale packed weights'
    # (b) the oracle with the updated state
    ref = model_ref.build_model(use_spike='True')
    ref.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        lr = ref(x.cpu()).numpy()
    sj_ref.reset_net(ref)
    frac = _frac_close(after.cpu().numpy(), lr, RTOL, 1e-4)
    print(f'after 2 fused-Adam steps: {frac * 100:.2f}% of logits within 1e-4 of the oracle on the updated weights')
    assert frac > 0.97


def test_model_matches_cpu_oracle_on_fresh_input(dev):
    """Same weights, fresh seeded input: HIP model vs the torch-CPU oracle model (not only the stored fixtures)."""
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True'])
    hip = exp.get_model()
    ref = model_ref.build_model(use_spike='True')
    assert fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS['True']) == fill.procedural_fill_(ref, 2.0, ann_regex=fill.ANN_KEYS['True'])
    x = torch.from_numpy(fill.poisson_events((2, 1, 4, 2, 96, 128), 0.5, seed=77))
    ref.eval(); hip.to(dev).eval()
    with torch.no_grad():
        lr = ref(x).numpy()
        lh = hip(x.to(dev)).cpu().numpy()
    sj_ref.reset_net(ref); functional.reset_net(hip)
    assert _frac_close(lh, lr, RTOL, 1e-4) > 0.97


def test_energy_estimation_matches_the_reference_function(dev):
    """eas_snn_amd.stats.energy_estimation (device taps + eas_spike_sop) against (a) the figures the reference's own
    EventEvaluator.energy_estimation printed for the same weights and batches (tests/golden/energy_estimation.npz, generated by
    running event_evaluator.py:466-565 on the CPU) and (b) the restated procedure on the torch-CPU oracle model.  MAC counts are
    exact; accumulate counts depend on the spikes, which may differ by a few rounding-level flips between a GPU and a CPU run of a
    spiking net: 1 %.  Like the reference, no reset between the batches."""
    from eas_snn_amd import stats
    from oracle import fill, model_ref, sj_ref, stats_ref
    from yolox.exp import get_exp
    g = load_golden('energy_estimation')
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True'])
    hip = exp.get_model()
    ref = model_ref.build_model(use_spike='True')
    assert fill.procedural_fill_(hip, 2.0, ann_regex=fill.ANN_KEYS['True']) == fill.procedural_fill_(ref, 2.0, ann_regex=fill.ANN_KEYS['True']) == int(g['crc'])
    batches = [torch.from_numpy(fill.poisson_events(tuple(int(v) for v in g['shape']), 0.5, seed=int(s_))) for s_ in g['seeds']]
    want = stats_ref.energy_estimation(ref, batches, T=3, reset_fn=None)
    got = stats.energy_estimation(hip.to(dev), [b.to(dev) for b in batches], T=3)
    assert got['num_samples'] == want['num_samples'] == 6
    for k in stats.GROUPS:
        assert got['module_mac'][k] == want['module_mac'][k], k
        np.testing.assert_allclose(got['module_mac'][k] / 6 / 1e9, float(g[f'mac_G/{k}']), rtol=1e-6, err_msg=k)     # fp32 sums in the reference
        np.testing.assert_allclose(got['module_ac'][k], want['module_ac'][k], rtol=1e-2, err_msg=k)
        np.testing.assert_allclose(got['module_ac'][k] / 6 / 1e9, float(g[f'ac_G/{k}']), rtol=1e-2, err_msg=k)
    assert got['module_ac']['embedding'] > 0 and got['module_ac']['backbone'] > 0
    np.testing.assert_allclose(got['ann_energy'], float(g['ann_energy']), rtol=1e-6)
    np.testing.assert_allclose(got['snn_energy'], float(g['snn_energy']), rtol=1e-2)
    np.testing.assert_allclose(got['ann_energy'], want['ann_energy'], rtol=1e-12)
    # every batch from rest: differs from the carried-state figures (the second batch starts from other potentials)
    rest = stats.energy_estimation(hip, [b.to(dev) for b in batches], T=3, reset_between_batches=True)
    assert rest['module_mac'] == got['module_mac'] and rest['module_ac']['backbone'] != got['module_ac']['backbone']
    with pytest.raises(ValueError):                                    # the reference reshapes by T: batch 2 with T = 3 is an error
        stats.energy_estimation(hip, [batches[0][:2].to(dev)], T=3)


@pytest.mark.parametrize('B,G,nc,hw,seed', [(4, 50, 2, (64, 96), 1), (3, 12, 100, (64, 64), 2), (2, 50, 3, (256, 320), 3), (3, 50, 2, (640, 640), 5)])
def test_fused_simota_assignment_equals_tensor_form(dev, B, G, nc, hw, seed):
    """eas_simota_assign (one launch over the valid label rows) against YOLOXHead._assign (the batch-wide tensor formulation
    that the golden train-loss fixtures pin): foreground mask, matched rows and matched IoUs.  Inputs: random decoded
    predictions around a few ground-truth boxes, images with 0, 1 and many labels, clustered so that anchors are claimed by
    several boxes (the least-cost rule) and dynamic k varies."""
    from eas_snn_amd import ops
    from yolox.models.yolo_head import YOLOXHead
    g = torch.Generator().manual_seed(seed)
    H, W = hw
    head = YOLOXHead(nc, width=0.5)
    grids, strides = [], []
    for s_ in (8, 16, 32):
        hs, ws = H // s_, W // s_
        yv, xv = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing='ij')
        grids.append(torch.stack((xv, yv), 2).view(1, -1, 2).float())
        strides.append(torch.full((1, hs * ws), float(s_)))
    grids, strides = torch.cat(grids, 1), torch.cat(strides, 1)
    A = grids.shape[1]
    n_valid = torch.randint(0, min(G, 9), (B,), generator=g)
    n_valid[0] = 0
    if B > 1:
        n_valid[1] = min(G, 8)
    gt_valid = torch.arange(G)[None] < n_valid[:, None]
    centres = torch.rand(B, G, 2, generator=g) * torch.tensor([W * 0.6, H * 0.6]) + torch.tensor([W * 0.2, H * 0.2])
    centres[:, 1::2] = centres[:, 0::2][:, :centres[:, 1::2].shape[1]] + 6.0          # overlapping pairs -> anchors with several claimants
    gt_boxes = torch.cat([centres, torch.rand(B, G, 2, generator=g) * 60 + 12], -1)
    gt_cls = torch.randint(0, nc, (B, G), generator=g).float()
    ctr = (grids[0] + 0.5) * strides[0][:, None]
    bbox = torch.cat([ctr[None].expand(B, -1, -1) + torch.randn(B, A, 2, generator=g) * 4, torch.rand(B, A, 2, generator=g) * 70 + 8], -1)
    obj, cls = torch.randn(B, A, 1, generator=g) * 2, torch.randn(B, A, nc, generator=g) * 2
    args = [t.to(dev) for t in (grids, strides, gt_boxes, gt_cls, gt_valid, bbox, obj, cls)]
    fg_r, m_r, iou_r = head._assign(*args)
    fg_h, m_h, iou_h = ops.simota_assign(*args)
    assert fg_h.dtype == torch.bool and m_h.dtype == torch.int64
    ndiff = int((fg_r != fg_h).sum())                  # (an integer count: a float32 mean of 25 200 ones is 0.99999994)
    assert ndiff == 0, f'foreground masks differ on {ndiff} anchors'
    assert torch.equal(m_r[fg_r], m_h[fg_h]) and int(fg_r.sum()) > 0 and int(fg_r[0].sum()) == 0
    torch.testing.assert_close(iou_h, iou_r, rtol=1e-6, atol=1e-7)
    multi = int((fg_r & (iou_r > 0)).sum())
    assert multi > 0


@pytest.mark.parametrize('nc,use_l1,hw,seed', [(2, True, (64, 96), 1), (2, False, (64, 96), 2), (100, True, (64, 64), 3), (3, True, (256, 320), 4), (2, True, (640, 640), 6)])
def test_fused_detection_loss_equals_tensor_form(dev, nc, use_l1, hw, seed):
    """ops.det_loss (eas_det_decode + eas_simota_assign_rows + eas_det_loss: decode, assignment, loss terms and their
    gradient in five launches) against the tensor-op get_losses of the same head on the same features and labels: the six
    returned numbers and the gradient of the total loss w.r.t. every head parameter and the input features."""
    import copy
    from yolox.models.yolo_head import YOLOXHead
    torch.manual_seed(seed)
    H, W = hw
    head = YOLOXHead(nc, width=0.25).to(dev).train()
    head.use_l1 = use_l1
    B = 4
    feats = [torch.randn(B, c, H // s_, W // s_, device=dev) * 0.5 for c, s_ in zip((64, 128, 256), (8, 16, 32))]
    labels = torch.zeros(B, 50, 5, device=dev)
    g = torch.Generator().manual_seed(seed)
    for b in range(1, B):                                   # image 0 has no label
        n = int(torch.randint(1, 7, (1,), generator=g))
        cxy = torch.rand(n, 2, generator=g) * torch.tensor([W * 0.6, H * 0.6]) + torch.tensor([W * 0.2, H * 0.2])
        labels[b, :n] = torch.cat([torch.randint(0, nc, (n, 1), generator=g).float(), cxy, torch.rand(n, 2, generator=g) * 40 + 10], 1).to(dev)
    res = []
    for fused in (False, True):
        h = copy.deepcopy(head)
        h.fused_loss = h.fused_assign = fused
        xs = [f.clone().requires_grad_(True) for f in feats]
        out = h(xs, labels)
        out[0].backward()
        res.append(([float(v) for v in out], [x.grad.clone() for x in xs], {n_: p.grad.clone() for n_, p in h.named_parameters() if p.grad is not None}))
    (v0, gx0, gp0), (v1, gx1, gp1) = res
    np.testing.assert_allclose(v1, v0, rtol=2e-5, atol=1e-6)
    assert v0[5] > 0 and set(gp0) == set(gp1)
    for a, b_ in zip(gx0, gx1):
        torch.testing.assert_close(b_, a, rtol=1e-4, atol=1e-6 * float(a.abs().max()) + 1e-9)
    for k in gp0:
        torch.testing.assert_close(gp1[k], gp0[k], rtol=1e-4, atol=2e-6 * float(gp0[k].abs().max()) + 1e-9, msg=k)


@pytest.mark.parametrize('train', [True, False])
def test_csp_single_convolution_for_both_branches_is_identical(dev, train, monkeypatch):
    """CSPLayer with conv1 / conv2 computed by ONE 1x1 convolution (concatenated weights, the two BN+LIF layers on channel slices of
    its output, one gradient tensor) against the two-convolution path: spikes and BN running statistics bit-identical (the same
    products are summed in the same order per output channel), gradients equal to rounding."""
    import copy
    from yolox.models.network_blocks import CSPLayer
    from yolox.utils.utils_snn import convert_to_spiking
    from spikingjelly.activation_based import functional, surrogate
    torch.manual_seed(5)
    base = CSPLayer(32, 32, n=2)
    convert_to_spiking(base, spike_fn=surrogate.ATan(2.0))
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
            with torch.no_grad():
                m.running_mean.fill_(-0.8)          # eval mode: neurons must fire for the comparison to mean something
    x = (torch.rand(3, 2, 32, 16, 20) < 0.3).float()
    go = torch.randn(3, 2, 32, 16, 20)
    res = []
    for nodual in ('1', ''):
        if nodual:
            monkeypatch.setenv('EAS_NO_DUAL', '1')
        else:
            monkeypatch.delenv('EAS_NO_DUAL', raising=False)
        net = copy.deepcopy(base).to(dev).train(train)
        assert net._dual_ok() == (not nodual)
        from eas_snn_amd import ops
        xd = ops.mark_small_int(x.to(dev)).requires_grad_(True)
        out = net(xd)
        out.backward(go.to(dev))
        functional.reset_net(net)
        res.append((out.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()},
                    {k: b.clone() for k, b in net.named_buffers()}))
    (o0, g0, p0, b0), (o1, g1, p1, b1) = res
    assert torch.equal(o0, o1) and float(o0.sum()) > 0
    # the input gradient sums the two branches' products in one accumulation instead of two sums and an addition: rounding-level
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-6 * float(g0.abs().max()))
    for k in p0:
        torch.testing.assert_close(p1[k], p0[k], rtol=1e-5, atol=1e-6 * float(p0[k].abs().max()) + 1e-12, msg=k)
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


@pytest.mark.parametrize('n,shortcut,dual', [(2, True, True), (1, False, True), (0, True, True), (3, True, False)])
def test_csp_layer_on_spike_planes_is_bit_identical(dev, monkeypatch, n, shortcut, dual):
    """A converted CSPLayer whose internal spike tensors (branch outputs, bottleneck outputs with their SEW shortcuts, the in-place
    concatenation) are bf16 spike planes, inside a ``packed_weights`` scope as in the model's forward, against the same layer on fp32 spike
    tensors: output, input gradient, parameter gradients and BatchNorm buffers bit-identical; a block called outside such a scope hands out
    fp32 spikes."""
    import copy
    from eas_snn_amd import ops
    from yolox.models.network_blocks import CSPLayer, enable_spike_planes
    from yolox.utils.utils_snn import convert_to_spiking
    from spikingjelly.activation_based import functional, surrogate
    if not dual:
        monkeypatch.setenv('EAS_NO_DUAL', '1')
    torch.manual_seed(7 + n)
    base = CSPLayer(64, 64, n=n, shortcut=shortcut)
    convert_to_spiking(base, spike_fn=surrogate.ATan(2.0))
    x = (torch.rand(3, 2, 64, 16, 20) < 0.3).float()
    go = torch.randn(3, 2, 64, 16, 20)
    res = []
    for planes in (True, False):
        net = enable_spike_planes(copy.deepcopy(base).to(dev).train(), planes)
        if planes:
            probe = net(ops.mark_small_int(x.to(dev)))
            assert ops.planes_of(probe) is None and not bool(torch.isnan(probe).any())       # on its own: fp32 out
            functional.reset_net(net)
            net = enable_spike_planes(copy.deepcopy(base).to(dev).train(), True)
        xd = ops.mark_small_int(x.to(dev)).requires_grad_(True)
        with ops.packed_weights(net):
            out = net(xd)
            assert (ops.planes_of(out) is not None) == planes
            val = ops.dense(out)
        val.backward(go.to(dev))
        functional.reset_net(net)
        res.append((val.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()},
                    {k: b.clone() for k, b in net.named_buffers()}))
    (o0, g0, p0, b0), (o1, g1, p1, b1) = res
    assert torch.equal(o0, o1) and float(o0.sum()) > 0
    assert torch.equal(g0, g1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


@pytest.mark.parametrize('n,H,W', [(1, 16, 20), (2, 8, 10), (1, 32, 40)])
def test_ann_csp_concatenation_in_place_is_bit_identical(dev, monkeypatch, n, H, W):
    """Real-valued CSPLayer (PAFPN neck, shortcut=False): both branches' BN + SiLU kernels write into the two channel halves of one buffer
    (eas_bn_silu_fwd_ex out_ctot) and their backward reads its half of the gradient in place (eas_bn_silu_bwd grad_out_ctot, incl. the
    one-launch small-map form) -- against torch.cat of separate outputs and contiguous gradient copies: same kernels on the same values,
    so outputs, running statistics and every gradient are bit-identical."""
    import copy
    from yolox.models.network_blocks import CSPLayer
    monkeypatch.setenv('EAS_NO_DUAL', '1')      # conv1 / conv2 as separate convolutions in both runs: this test isolates the concatenation
    torch.manual_seed(9)
    base = CSPLayer(128, 128, n=n, shortcut=False)
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    x = torch.randn(8, 128, H, W)
    go = torch.randn(8, 128, H, W)
    res = []
    for inplace in (False, True):
        net = copy.deepcopy(base).to(dev).train()
        if not inplace:
            monkeypatch.setattr(CSPLayer, '_ann_inplace_cat', lambda self, x_: False)
        else:
            monkeypatch.undo()
            monkeypatch.setenv('EAS_NO_DUAL', '1')
            assert net._ann_inplace_cat(x.to(dev))
        xd = x.to(dev).requires_grad_(True)
        out = net(xd)
        out.backward(go.to(dev))
        res.append((out.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()},
                    {k: b.clone() for k, b in net.named_buffers()}))
    (o0, g0, p0, b0), (o1, g1, p1, b1) = res
    assert torch.equal(o0, o1) and torch.equal(g0, g1)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


@pytest.mark.parametrize('which', ['csp', 'head', 'spiking_head'])
def test_real_valued_dual_convolution_matches_separate_convolutions(dev, monkeypatch, which):
    """Two real-valued convolutions that read the same input as ONE convolution (ops.conv2d_dual: weights packed from the two
    parameters; ops.bn_silu_pair on the channel halves): conv1 | conv2 of the PAFPN CSPLayer and the first cls / reg tower convolutions
    of the head.  Forward values and running statistics are bit-identical to the separate convolutions (same products, same order per
    output channel); gradients agree to rounding (one input-gradient accumulation instead of two sums and an addition)."""
    import copy
    from eas_snn_amd import ops
    from yolox.models.network_blocks import CSPLayer
    from yolox.models.yolo_head import YOLOXHead
    torch.manual_seed(11)
    if which == 'csp':
        base = CSPLayer(128, 128, n=1, shortcut=False)
        x = torch.randn(8, 128, 16, 20)
        run = lambda net, xd: net(xd)
    elif which == 'head':
        base = YOLOXHead(3, width=0.5)
        x = torch.randn(8, 128, 8, 10)
        run = lambda net, xd: torch.cat(net._towers(2, xd), 1)
    else:
        # the converted head of the full_spike_v2 models: the two spiking towers' first 3x3 convolutions as one (+ the BN+LIF pair)
        from spikingjelly.activation_based import functional, surrogate
        from yolox.models.yolo_head import SpikingYOLOXHead
        base = SpikingYOLOXHead(3, width=0.5, spike_fn=surrogate.ATan(2.0), full_spike=True)
        x = (torch.rand(3, 4, 128, 8, 10) < 0.3).float()
        ops.mark_small_int(x)

        def run(net, xd):
            ops.mark_small_int(xd)
            a, b = net._towers(2, xd)
            functional.reset_net(net)
            return torch.cat([ops.dense(a), ops.dense(b)], 2)
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
            if which == 'spiking_head':
                with torch.no_grad():
                    m.bias.fill_(0.6)          # neurons must fire for the comparison to mean something
    res = []
    for nodual in ('1', ''):
        if nodual:
            monkeypatch.setenv('EAS_NO_DUAL', '1')
        else:
            monkeypatch.delenv('EAS_NO_DUAL', raising=False)
        net = copy.deepcopy(base).to(dev).train()
        xd = x.to(dev).requires_grad_(True)
        with ops.packed_weights(net):
            out = run(net, xd)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        out.backward(go)
        res.append((out.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                    {k: b.clone() for k, b in net.named_buffers()}))
    (o0, g0, p0, b0), (o1, g1, p1, b1) = res
    assert torch.equal(o0, o1)
    if which == 'spiking_head':
        assert 0.02 < float(o0.mean()) < 0.9
    assert set(p0) == set(p1) and len(p0) > 0
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=2e-6 * float(g0.abs().max()))
    for k in p0:
        torch.testing.assert_close(p1[k], p0[k], rtol=1e-5, atol=2e-6 * float(p0[k].abs().max()) + 1e-12, msg=k)
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


def test_state_dict_roundtrip_and_writeback_switch(dev):
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    g, model = _build('model_s_true_64', dev)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert not any(k.endswith('.v') for k in sd)                      # neuron state is not a checkpoint entry
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = torch.from_numpy(g['x']).to(dev)
    with torch.no_grad():
        a = model(x)
        functional.reset_net(model)
        ops.set_state_writeback(False)
        try:
            b = model(x)
        finally:
            ops.set_state_writeback(True)
        functional.reset_net(model)
    assert torch.equal(a, b)            # every convolution is an own kernel with a fixed summation order: bit-identical


CONFIG_CASES = {
    # BASELINE.json configs[3]: 1Mpx stacked histogram summed to 2 channels, 3 classes, T=3, Tm=Tl micro-steps (canvas 384x640 scaled /4)
    'cfg4_1mpx': (dict(depth=0.33, width=0.5, T=3, Tm=2, Ts=1, num_classes=3, use_spike='full_spike_v2'), (1, 1, 2, 2, 96, 160)),
    # configs[4]: N-Caltech101, 100 classes, T=7, Tm=8 long-sequence adaptive sampling with Ts=T (canvas 192x256 scaled /2)
    'cfg5_ncaltech_ts7': (dict(depth=0.33, width=0.5, T=7, Tm=8, Ts=7, num_classes=100, use_spike='True', alpha=1.5), (1, 1, 8, 2, 96, 128)),
    'cfg5_ncaltech_ts1': (dict(depth=0.33, width=0.5, T=7, Tm=8, Ts=1, num_classes=100, use_spike='True'), (2, 1, 8, 2, 64, 64)),
    # the simpler embeddings in front of the whole model
    'emb_rsnn': (dict(depth=0.33, width=0.5, T=3, Tm=4, embedding='rsnn', use_spike='True'), (2, 1, 4, 2, 64, 96)),
    'emb_snn': (dict(depth=0.33, width=0.5, T=3, Tm=4, embedding='snn', use_spike='True'), (2, 1, 4, 2, 64, 96)),
    'emb_count': (dict(depth=0.33, width=0.5, T=3, Tm=4, embedding='count', use_spike='True'), (2, 1, 4, 2, 64, 96)),
}


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CONFIG_CASES))
def test_config_shaped_cases_vs_cpu_oracle(dev, name):
    """Parity cases shaped like BASELINE.json configs[3], configs[4] (canvases scaled down so the CPU oracle finishes in
    seconds) and the alternative embeddings: same procedural weights, fresh seeded Poisson event volumes, eval logits of
    the HIP model against the torch-CPU oracle model within 1e-4 (relative + absolute) for the bulk of the anchors --
    single rounding-level spike flips are amplified by the spiking backbone, see DESIGN.md section 5."""
    from oracle import fill, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    cfg, shape = CONFIG_CASES[name]
    H, W = shape[-2:]
    exp = get_exp(None, 'e-yolox-s')
    opts = list(BASE_OPTS) + ['input_size', f'({H},{W})', 'test_size', f'({H},{W})']
    for k, v in cfg.items():
        if k in ('depth', 'width'):
            continue
        if k in [o for o in opts[::2]]:
            opts[opts.index(k) + 1] = str(v)
        else:
            opts += [k, str(v)]
    exp.merge(opts)
    hip = exp.get_model()
    ref = model_ref.build_model(**cfg)
    rx = fill.ANN_KEYS[cfg['use_spike']]
    assert fill.procedural_fill_(hip, 2.0, ann_regex=rx) == fill.procedural_fill_(ref, 2.0, ann_regex=rx)
    x = torch.from_numpy(fill.poisson_events(shape, 0.5, seed=123))
    ref.eval(); hip.to(dev).eval()
    with torch.no_grad():
        lr = ref(x).numpy()
        lh = hip(x.to(dev)).cpu().numpy()
    sj_ref.reset_net(ref); functional.reset_net(hip)
    assert lh.shape == lr.shape
    assert _frac_close(lh, lr, RTOL, 1e-4) > 0.95, _frac_close(lh, lr, RTOL, 1e-4)


@pytest.mark.parametrize('batch', [2, 64])
def test_head_with_grouped_launches_equals_the_per_level_head(dev, batch):
    """YOLOXHead in training mode: every stage of the three pyramid levels as one grouped launch (``_levels_grouped``, ops_group) against
    the per-level walk of the reference (yolo_head.py:149-200): loss, input gradients, all parameter gradients and the running statistics.
    The convolutions are bit-identical per problem; BatchNorm sums and weight-gradient slabs are cut by another tile plan (rounding)."""
    from eas_snn_amd import data, ops
    from eas_snn_amd import ops_group as G
    from yolox.models.yolo_head import YOLOXHead
    torch.manual_seed(5)
    hd = YOLOXHead(2, width=0.5).to(dev)
    hd.initialize_biases(1e-2)
    hd.train()
    hd.use_l1 = True
    for m in hd.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    xs0 = [torch.randn(batch, c, h, w, device=dev) for c, (h, w) in zip((128, 256, 512), ((32, 40), (16, 20), (8, 10)))]
    labels = data.synth_targets(batch, (256, 320), dev)
    state = {k: v.clone() for k, v in hd.state_dict().items()}
    res = {}
    prev = G.ENABLED
    try:
        for flag in (False, True):
            G.ENABLED = flag
            hd.load_state_dict(state)
            hd.zero_grad(set_to_none=True)
            xs = [x.clone().requires_grad_(True) for x in xs0]
            with ops.packed_weights(hd), ops.kernel_trace() as tr:
                out = hd(xs, labels)
                out[0].backward()
            grouped = any('group_kernel' in k for k in tr.kernels)
            assert grouped == flag, tr.kernels
            res[flag] = (out[0].detach().clone(), [x.grad.clone() for x in xs], {n: p.grad.clone() for n, p in hd.named_parameters()},
                         {n: b.clone() for n, b in hd.named_buffers()})
    finally:
        G.ENABLED = prev
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=0)
    for a, b in zip(res[True][1], res[False][1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
    for n in res[False][2]:
        torch.testing.assert_close(res[True][2][n], res[False][2][n], rtol=1e-4, atol=2e-5 * float(res[False][2][n].abs().max()) + 1e-12, msg=n)
    for n in res[False][3]:
        torch.testing.assert_close(res[True][3][n].float(), res[False][3][n].float(), rtol=1e-5, atol=1e-7, msg=n)


@pytest.mark.gpu
@pytest.mark.parametrize('B,nc,hws', [(2, 2, ((32, 40), (16, 20), (8, 10))), (3, 100, ((24, 32), (12, 16), (6, 8))), (64, 2, ((80, 80), (40, 40), (20, 20)))])
def test_inference_output_of_the_head_in_one_launch(dev, B, nc, hws):
    """eas_det_decode_eval against the tensor operators of YOLOXHead's inference path (yolo_head.py:187-216 of the reference: sigmoid on
    objectness / classes, cat along the anchors, (xy + grid) * stride, exp(wh) * stride): boxes bit-identical, probabilities within 2 ulp."""
    from eas_snn_amd import ops
    from yolox.models.yolo_head import YOLOXHead
    g = torch.Generator().manual_seed(B + nc)
    raws = [tuple(torch.randn(B, c, h, w, generator=g).to(dev) for c in (4, 1, nc)) for h, w in hws]
    hd = YOLOXHead(nc, width=0.5)
    hd.decode_in_inference = True
    import yolox.models.yolo_head as yh
    prev = yh.FUSED_EVAL_DECODE
    try:
        with torch.no_grad():
            yh.FUSED_EVAL_DECODE = True
            with ops.kernel_trace() as tr:
                got = hd.assemble_eval(raws)
            assert any('det_decode' in k for k in tr.kernels), tr.kernels
            yh.FUSED_EVAL_DECODE = False
            want = hd.assemble_eval(raws)
    finally:
        yh.FUSED_EVAL_DECODE = prev
    assert got.shape == want.shape == (B, sum(h * w for h, w in hws), 5 + nc)
    assert torch.equal(got[..., :2], want[..., :2])
    torch.testing.assert_close(got[..., 2:4], want[..., 2:4], rtol=3e-7, atol=0)
    torch.testing.assert_close(got[..., 4:], want[..., 4:], rtol=3e-7, atol=1e-9)


@pytest.mark.gpu
def test_frozen_weights_scope_reuses_the_packs_and_never_serves_stale_ones(dev):
    """ops.frozen_weights (the evaluator's loop): inside the block an eval forward launches no weight-packing kernel and gives the bits of an
    ordinary forward; a HIP graph captured inside it follows weights AND BatchNorm statistics changed between two blocks (the next
    evaluation after more training), because entering a block refreshes the same buffers; outside a block nothing is cached."""
    from eas_snn_amd import ops
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(BASE_OPTS + ['use_spike', 'True'])          # the 256x320 canvas: every layer on the own kernels (bit-reproducible)
    from oracle import fill
    model = exp.get_model()
    fill.procedural_fill_(model, 2.0, ann_regex=fill.ANN_KEYS['True'])       # weights that make every layer fire
    model = model.to(dev).eval()
    torch.manual_seed(11)
    for m in model.modules():          # statistics that are not the initial (0, 1): the BatchNorm constants matter
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.uniform_(0.5, 2.0)
            m.running_mean.normal_(0.0, 0.1)
    x = torch.from_numpy(np.random.default_rng(3).poisson(0.5, (2, 1, 4, 2, 256, 320)).astype(np.float32)).to(dev)

    def fwd():
        out = model(x)
        functional.reset_net(model)
        return out
    with torch.no_grad():
        ref = fwd().clone()
        assert torch.equal(fwd(), ref)                                   # the eval forward itself is reproducible
        with ops.frozen_weights(model):
            with ops.kernel_trace() as tr:
                a = fwd().clone()
            assert not any('pack_weights_flat' in k or 'pack_weights_many' in k for k in tr.kernels), tr.kernels      # (the model-wide packing launch)
            assert torch.equal(a, ref)
            fwd()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fwd()
            g.replay()
            assert torch.equal(out, ref)
        with ops.kernel_trace() as tr:
            fwd()
        assert any('pack_weights_flat' in k or 'pack_weights_many' in k for k in tr.kernels)              # outside the block: packed per forward again
        for m in model.modules():                                        # "more training": every weight and every statistic moves, in place
            if isinstance(m, torch.nn.Conv2d) and m.kernel_size in ((1, 1), (3, 3)):
                m.weight.mul_(1.25)
            elif isinstance(m, torch.nn.BatchNorm2d):
                m.running_var.mul_(0.5)
        ref2 = fwd().clone()
        assert not torch.equal(ref2, ref)
        with ops.frozen_weights(model):
            g.replay()                                                   # the graph of the first block
            assert torch.equal(out, ref2)
            assert torch.equal(fwd(), ref2)


def _bench_child(extra_env, steps=3, warmup=3, timeout=900):
    """bench.py in a FRESH child process (the pytest process already holds the GPU and must never exec; the child is an ordinary
    subprocess with its own HIP context, like the ranks bench.py --gpus N starts); returns its JSON line"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    # EAS_BENCH_GRAPH=1: graph replay without the eager-against-replay probe of the one-GPU warm-up (eight more steps than the exchange path runs)
    env = dict(os.environ, EAS_BENCH_NO_EVAL='1', EAS_BENCH_NO_640='1', EAS_BENCH_NO_EMA='1', EAS_BENCH_GRAPH='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, f'bench.py exited with {r.returncode}:\n{r.stderr[-2000:]}'
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines, r.stdout[-500:]
    return json.loads(lines[-1])


def test_bench_step_holds_a_live_rccl_communicator(dev):
    """The N > 1 launch form with a real RCCL communicator in the loop (yolox/core/trainer.py:175-176 wraps the model for N ranks,
    yolox/core/launch.py:118-142 makes the process group): one rank, process group on the 'nccl' (= RCCL) backend, the bucketed gradient
    exchange between three HIP-graph replays -- bench.py with EAS_BENCH_FORCE_DDP=1 in a child process.  It must finish, report the
    exchange it ran, and train exactly like the one-graph step (same loss after the same number of steps)."""
    plain = _bench_child({})
    ddp = _bench_child({'EAS_BENCH_FORCE_DDP': '1'})
    cfg = ddp['config']
    assert cfg['rccl_ranks'] == 1 and ddp['n_gpus'] == 1
    assert cfg['launch'].startswith('three hip-graph replays per step'), cfg['launch']
    assert cfg['gradient_exchange'] == '2 flat bucket(s)', cfg['gradient_exchange']
    assert plain['config']['launch'] == 'hip-graph replay of the whole step' and plain['config']['gradient_exchange'] is None
    a, b = plain['final_loss'], ddp['final_loss']
    assert np.isfinite(a) and abs(a - b) <= 1e-6 * abs(a), (a, b)


_LIVE_GROUP_CHILD = r"""
import os, sys, types
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import eas_snn_amd
from eas_snn_amd import ops, workloads
from yolox.utils import wait_process_group_idle
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')
dist.init_process_group('nccl', device_id=dev)                    # the group exists BEFORE anything is recorded (a caller that did not defer it)
torch.cuda.set_stream(torch.cuda.Stream())
ops.set_state_writeback(False)
w = workloads.get(2)
trainer, model, step = workloads.build_trainer(w, 4, dev, events=20000, world=1, rank=0, force_exchange=True, out_dir=%r)
assert trainer.exchange is not None and trainer.exchange.bound
for _ in range(3):
    step()                                                        # eager steps with real all-reduces in flight
assert wait_process_group_idle() is True, 'the flight recorder is not available: captures with a live group would stay eager'
import pickle
from torch._C._distributed_c10d import _dump_nccl_trace
seen = pickle.loads(_dump_nccl_trace(True, False, False))['entries']           # every recorded collective, retired ones included
assert len(seen) >= 6, f'the flight recorder lists {len(seen)} collectives after three exchanged steps: it is not recording'
assert not pickle.loads(_dump_nccl_trace(True, False, True))['entries']        # ... and none of them is still held by the watchdog
form = step.capture(warm=1)
assert form.startswith('three hip-graph replays'), form
losses = []
for _ in range(4):
    losses.append(float(step().detach()))
torch.cuda.synchronize()
assert all(l == l and abs(l) < 1e6 for l in losses), losses
# the evaluator with the group alive: records its two graphs per batch shape after the same condition wait
from yolox.exp import get_exp
exp = workloads.build_exp(w)
exp.eval_samples, exp.eval_events = 8, 20000
ev = exp.get_evaluator(2, False)
dist.all_reduce(torch.ones(4, device=dev))
ev.evaluate(model, False, False, None, None, exp.test_size)
assert ev.use_graph and ev.graphs_recorded >= 1, (ev.use_graph, ev.graphs_recorded)
dist.destroy_process_group()
print('LIVE_GROUP_OK', form[:40], losses)
"""


def test_capture_with_a_live_process_group_waits_for_the_watchdog(dev, tmp_path):
    """The condition wait that replaced the 0.3 s sleeps of round 5: a caller that created the RCCL process group BEFORE recording
    (not bench.py, not the Trainer -- they record first -- but e.g. the evaluator under tools/eval_event.py's DistributedDataParallel
    wrapper) gets its HIP graphs recorded after ``wait_process_group_idle`` has seen the watchdog retire every collective (flight
    recorder), never after a delay.  Child process: one-rank group on 'nccl', eager steps with all-reduces, TrainStep.capture and the
    evaluator's recording with the group alive, replays."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    script = tmp_path / 'live_group.py'
    script.write_text(_LIVE_GROUP_CHILD % (root, str(tmp_path / 'out')))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and 'LIVE_GROUP_OK' in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize('batch', [2, 64])
@pytest.mark.parametrize('fuse', [False, True])
def test_eval_head_with_grouped_launches_is_bit_identical(dev, batch, fuse):
    """eval mode: every stage of the head's three levels as ONE grouped launch of the fused conv -> BatchNorm (running statistics) -> SiLU
    kernel (eas_conv_bn_act_eval_group) and the nine prediction convolutions as one launch, against the per-level forward
    (yolo_head.py:149-200 of the reference; after fuse_model: model_utils.py:35-80): same arithmetic per problem, bit-identical detections"""
    from eas_snn_amd import ops
    from eas_snn_amd import ops_group as G
    from yolox.models.yolo_head import YOLOXHead
    from yolox.utils import fuse_model
    torch.manual_seed(9)
    hd = YOLOXHead(2, width=0.5).to(dev)
    hd.initialize_biases(1e-2)
    with torch.no_grad():
        for m in hd.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    hd.eval()
    if fuse:
        hd = fuse_model(hd)
    xs = [torch.randn(batch, c, h, w, device=dev) for c, (h, w) in zip((128, 256, 512), ((32, 40), (16, 20), (8, 10)))]
    res = {}
    prev = G.ENABLED
    try:
        for flag in (False, True):
            G.ENABLED = flag
            with torch.no_grad(), ops.packed_weights(hd), ops.kernel_trace() as tr:
                res[flag] = hd(xs).clone()
            assert any('group_kernel' in k for k in tr.kernels) == flag, tr.kernels
    finally:
        G.ENABLED = prev
    assert torch.isfinite(res[True]).all() and torch.equal(res[True], res[False])
