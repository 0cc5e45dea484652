#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator (runs ONLY in the build container).

Imports the real reference from /root/reference (read-only, never copied) and
writes small input/output fixtures to tests/golden/*.npz.  Nothing from the
reference travels: the fixtures are data (inputs, expected outputs, gradients).

    python -m oracle.gen_golden            # regenerate everything

What stands in for what (none of it carries hot-path arithmetic except sj_ref):
  * loguru / cv2 / torchvision / pycocotools / h5py / thop -> empty shims written to a temp dir
    (the reference imports them at package import time).
  * spikingjelly.activation_based -> oracle.sj_ref (spikingjelly==0.0.0.0.14 is not vendored;
    parity at that boundary is UNPINNED, see oracle/sj_ref.py).
"""
import os
import sys
import tempfile
import textwrap
import types
import zlib

sys.dont_write_bytecode = True          # never write __pycache__ into /root/reference
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

_SHIMS = {
    'loguru/__init__.py': '''
        class _L:
            def __getattr__(self, k):
                if k == 'catch':
                    return lambda f=None, **kw: f if callable(f) else (lambda g: g)
                return lambda *a, **k: None
        logger = _L()
    ''',
    'cv2/__init__.py': '''
        INTER_LINEAR = 1; INTER_NEAREST = 0; INTER_AREA = 3; INTER_CUBIC = 2; INTER_LANCZOS4 = 4
        def setNumThreads(n): pass
        class ocl:
            @staticmethod
            def setUseOpenCL(b): pass
    ''',
    'torchvision/__init__.py': 'from . import ops\n',
    'torchvision/ops.py': '''
        def nms(*a, **k): raise NotImplementedError
        def batched_nms(*a, **k): raise NotImplementedError
    ''',
    'pycocotools/__init__.py': '',
    'pycocotools/coco.py': 'class COCO: pass\n',
    'pycocotools/cocoeval.py': 'class COCOeval: pass\n',
    'pycocotools/mask.py': '',
    # container stand-in for the HDF5 reader: `File(path)['data']` is the array stored next to it as <path>.npy (no arithmetic)
    'h5py/__init__.py': '''
        import numpy as _np
        class File:
            def __init__(self, path, mode='r'):
                self._d = {'data': _np.load(path + '.npy', mmap_mode='r')}
            def __enter__(self):
                return self._d
            def __exit__(self, *a):
                return False
    ''',
    'thop/__init__.py': 'def profile(*a, **k): raise NotImplementedError\n',
}


def setup_reference_imports():
    shim_dir = tempfile.mkdtemp(prefix='eas_shims_')
    for rel, src in _SHIMS.items():
        path = os.path.join(shim_dir, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, shim_dir)
    from oracle import sj_ref
    sj_ref.install_as_spikingjelly()
    sys.path.insert(1, REF)
    return shim_dir


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'  wrote {os.path.relpath(path)}  {os.path.getsize(path) / 1024:.1f} KiB')


def _np(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- other event representations
def gen_event_reps():
    """voxel cube / time surface of the reference (yolox/utils/event_reps.py:92-160) on seeded synthetic streams."""
    from types import SimpleNamespace
    from yolox.data.datasets.gen1 import GEN1Dataset
    from yolox.utils.event_reps import to_timesurface_numpy, to_voxel_cube_numpy
    from oracle.events_ref import synth_events
    dts = np.dtype([('t', 'i8'), ('x', 'i2'), ('y', 'i2'), ('p', 'i1')])   # signed fields: torch cannot promote a uint32 window
    dtu = np.dtype([('t', 'u4'), ('x', 'u2'), ('y', 'u2'), ('p', 'u1')])

    def struct(t, x, y, p, dt):
        ev = np.zeros(len(t), dtype=dt)
        ev['t'], ev['x'], ev['y'], ev['p'] = t, x, y, p
        return ev

    cases = {}
    for name, (n, ns, tb, H, W, seed) in {'vc_small_n4_tb2': (3000, 4, 2, 24, 32, 21), 'vc_small_n3_tb3': (3000, 3, 3, 24, 32, 22),
                                          'vc_tiny_n4_tb2': (9, 4, 2, 8, 8, 23), 'vc_gen1_n4_tb2': (20000, 4, 2, 240, 304, 24),
                                          'vc_small_n5_tb1': (2000, 5, 1, 24, 32, 25)}.items():
        t, x, y, p = synth_events(n, H, W, seed=seed)
        out = to_voxel_cube_numpy(struct(t, x, y, p, dts), [W, H, 2], ns, tbins=tb)
        assert out.shape == (ns, 2 * tb, H, W) and (out == np.round(out)).all()
        cases[name] = dict(t=t, x=x, y=y, p=p, ns=ns, tbins=tb, H=H, W=W, out=out.astype(np.int32))
    t, x, y, p = synth_events(50, 8, 8, seed=26)
    one = np.full(50, 777, np.uint32)                                       # window == 0: nothing kept
    cases['vc_same_timestamp'] = dict(t=one, x=x, y=y, p=p, ns=4, tbins=2, H=8, W=8,
                                      out=to_voxel_cube_numpy(struct(one, x, y, p, dts), [8, 8, 2], 4, tbins=2).astype(np.int32))
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f'{name}/{k}'] = np.asarray(v)
    save('events_voxel_cube', names=np.array(sorted(cases)), **flat)

    cases = {}
    for name, (n, ns, H, W, tau, seed, dt) in {'ts_small_n4': (3000, 4, 24, 32, 50e3, 31, dtu), 'ts_small_n8': (3000, 8, 24, 32, 10e3, 32, dtu),
                                               'ts_tiny_n4': (9, 4, 8, 8, 50e3, 33, dtu), 'ts_gen1_n4': (20000, 4, 240, 304, 50e3, 34, dtu),
                                               'ts_small_n3_signed': (3000, 3, 24, 32, 50e3, 35, dts)}.items():
        t, x, y, p = synth_events(n, H, W, seed=seed)
        me = SimpleNamespace(img_size=(H, W), slice_args={'micro_slice': ns})
        me.slice_events = lambda e, k, overlap=0: GEN1Dataset.slice_events(me, e, k, overlap)
        out = GEN1Dataset.agrregate(me, struct(t, x, y, p, dt), 'timesurface')       # tau = 50e3 hard-coded at gen1.py:369
        if tau != 50e3:
            slices, d = me.slice_events(struct(t, x, y, p, dt), ns)
            out = to_timesurface_numpy(slices, [W, H, 2], dt=d, tau=tau)
        assert out.shape == (ns, 2, H, W)
        cases[name] = dict(t=t, x=x, y=y, p=p, ns=ns, H=H, W=W, tau=tau, out=out)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f'{name}/{k}'] = np.asarray(v)
    save('events_time_surface', names=np.array(sorted(cases)), **flat)


# ----------------------------------------------------------------------------- augmentation draws + boxes
def gen_augment():
    """GEN1Dataset.get_random_data (gen1.py:433-521) run with a SHAPE-ONLY stand-in for cv2.resize (cv2 is not installed; the
    stand-in returns ones of the requested size and carries no arithmetic).  What the fixtures pin: the order of the random
    draws, (nw, nh, dx, dy) -- read back from the rectangle of ones -- the flip, and the box transform.  The bilinear
    resize itself stays unpinned (oracle/events_ref.resize_linear restates OpenCV's algorithm)."""
    from types import SimpleNamespace
    import cv2
    from yolox.data.datasets.gen1 import GEN1Dataset
    cv2.resize = lambda image, dsize, interpolation: np.ones((dsize[1], dsize[0]) + image.shape[2:])
    cv2.INTER_LINEAR = getattr(cv2, 'INTER_LINEAR', 1)
    cases = {}
    boxes = np.array([[30, 40, 120, 160, 0], [200, 20, 290, 100, 1], [5, 5, 9, 200, 1], [150, 150, 300, 239, 0]], dtype=np.float64)
    for name, (ih, iw, h, w, random, letterbox, center, seed) in {
            'jitter_gen1_s1': (240, 304, 256, 320, True, True, False, 1), 'jitter_gen1_s2': (240, 304, 256, 320, True, True, False, 2),
            'jitter_gen1_s3': (240, 304, 256, 320, True, True, False, 3), 'jitter_gen1_s4': (240, 304, 256, 320, True, True, False, 4),
            'jitter_ncal_s5': (180, 240, 192, 256, True, True, False, 5), 'letterbox_gen1': (240, 304, 256, 320, False, True, False, 6),
            'letterbox_center': (240, 304, 320, 320, False, True, True, 7), 'stretch_gen1': (240, 304, 256, 320, False, False, False, 8)}.items():
        me = SimpleNamespace(letterbox_image=letterbox)
        me.rand = lambda a=0, b=1: np.random.rand() * (b - a) + a
        me.batch_resize = lambda images, dsize, interpolation: GEN1Dataset.batch_resize(me, images, dsize, interpolation)
        np.random.seed(seed)
        frames = np.zeros((2, 2, ih, iw))
        out, bb = GEN1Dataset.get_random_data(me, frames, boxes.copy(), (h, w), random=random, center=center)
        ys, xs = np.nonzero(out[0, 0])
        cases[name] = dict(ih=ih, iw=iw, h=h, w=w, random=int(random), letterbox=int(letterbox), center=int(center), seed=seed,
                           rect=np.array([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1]), boxes_in=boxes, boxes_out=bb)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f'{name}/{k}'] = np.asarray(v)
    save('augment_draws', names=np.array(sorted(cases)), **flat)


# ----------------------------------------------------------------------------- K1 events
def gen_events():
    from types import SimpleNamespace
    from yolox.data.datasets.gen1 import GEN1Dataset
    from yolox.utils.event_reps import to_voxel_grid_numpy

    dt = np.dtype([('t', 'u4'), ('x', 'u2'), ('y', 'u2'), ('p', 'u1')])
    dts = np.dtype([('t', 'i8'), ('x', 'i2'), ('y', 'i2'), ('p', 'i1')])   # signed polarity (ncaltech.py:44)

    def ref_micro_sum(t, x, y, p, Tm, H, W):
        ev = np.zeros(len(t), dtype=dt)
        ev['t'], ev['x'], ev['y'], ev['p'] = t, x, y, p
        me = SimpleNamespace(img_size=(H, W), slice_args={'micro_slice': Tm})
        me.slice_events = lambda e, n, overlap=0: GEN1Dataset.slice_events(me, e, n, overlap)
        me.agrregate = lambda e, method: GEN1Dataset.agrregate(me, e, method)
        return GEN1Dataset.agrregate(me, ev if len(t) else None, 'micro_sum')

    def ref_voxel(t, x, y, p, nb, H, W):
        ev = np.zeros(len(t), dtype=dts)
        ev['t'], ev['x'], ev['y'], ev['p'] = t, x, y, p
        return to_voxel_grid_numpy(ev, [W, H, 2], nb)

    from oracle.events_ref import synth_events
    rng = np.random.default_rng(7)
    cases = {}

    def add(name, t, x, y, p, Tm, H, W):
        out = ref_micro_sum(t, x, y, p, Tm, H, W)
        assert out.dtype == np.float64 and (out == np.round(out)).all()
        cases[name] = dict(t=np.asarray(t, np.uint32), x=np.asarray(x, np.uint16), y=np.asarray(y, np.uint16),
                           p=np.asarray(p, np.uint8), Tm=Tm, H=H, W=W, out=out.astype(np.int32))

    t, x, y, p = synth_events(20000, 240, 304, seed=0)
    add('gen1_20k_tm4', t, x, y, p, 4, 240, 304)
    t, x, y, p = synth_events(3000, 24, 32, seed=1)
    add('small_tm4', t, x, y, p, 4, 24, 32)
    add('small_tm8', t, x, y, p, 8, 24, 32)
    add('small_tm3', t, x, y, p, 3, 24, 32)                       # span not divisible -> tail dropped
    add('small_tm1', t, x, y, p, 1, 24, 32)
    add('empty', t[:0], x[:0], y[:0], p[:0], 4, 24, 32)
    add('single_event', t[:1], x[:1], y[:1], p[:1], 4, 24, 32)    # window == 0
    add('same_timestamp', np.full(50, 1234567, np.uint32), x[:50], y[:50], p[:50], 4, 24, 32)
    add('fewer_than_tm', t[:3], x[:3], y[:3], p[:3], 4, 24, 32)
    tt = t.copy()
    tt[-5:] = tt[-1]                                              # several events on the last microsecond
    add('last_us_cluster', tt, x, y, p, 4, 24, 32)
    add('all_negative', t, x, y, np.zeros_like(p), 4, 24, 32)
    add('all_positive', t, x, y, np.ones_like(p), 4, 24, 32)
    add('one_pixel', t, np.full_like(x, 31), np.full_like(y, 23), p, 4, 24, 32)   # heavy collisions, last pixel
    t2 = np.sort(rng.integers(0, 2 ** 32 - 1, 4000, dtype=np.uint64)).astype(np.uint32)
    add('wide_timestamps', t2, x[:3000].repeat(2)[:4000], y[:3000].repeat(2)[:4000], p[:3000].repeat(2)[:4000],
        4, 24, 32)
    t3, x3, y3, p3 = synth_events(5000, 180, 240, seed=3)         # N-Caltech sensor
    add('ncaltech_tm8', t3, x3, y3, p3, 8, 180, 240)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f'{name}/{k}'] = np.asarray(v)
    save('events_micro_sum', names=np.array(sorted(cases)), **flat)

    # voxel grid
    vox = {}
    for name, (n, nb, H, W, seed) in {'vg_small_nb5': (3000, 5, 24, 32, 11), 'vg_small_nb10': (3000, 10, 24, 32, 12),
                                      'vg_tiny_nb4': (7, 4, 8, 8, 13)}.items():
        t, x, y, p = synth_events(n, H, W, seed=seed)
        ps = p.astype(np.int8)
        out = ref_voxel(t.astype(np.int64), x, y, ps.copy(), nb, H, W)
        vox[name] = dict(t=t, x=x, y=y, p=p, nb=nb, H=H, W=W, out=out)
    vox['vg_empty'] = dict(t=t[:0], x=x[:0], y=y[:0], p=p[:0], nb=4, H=8, W=8,
                           out=ref_voxel(t[:0].astype(np.int64), x[:0], y[:0], p[:0].astype(np.int8), 4, 8, 8))
    flat = {}
    for name, c in vox.items():
        for k, v in c.items():
            flat[f'{name}/{k}'] = np.asarray(v)
    save('events_voxel_grid', names=np.array(sorted(vox)), **flat)

    # .dat records decoded by the reference's own reader (psee_loader/io/dat_events_tools.py), then binned by GEN1Dataset
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_dat_tools', os.path.join(REF, 'yolox/utils/psee_loader/io/dat_events_tools.py'))
    dat_tools = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dat_tools)
    from oracle.events_ref import encode_dat_file
    t, x, y, p = synth_events(6000, 240, 304, seed=31)
    x[100:107] = 16383                 # 14-bit extremes in mid-stream (first/last event define the slicing window); bits above bit 28 are ignored
    image = bytearray(encode_dat_file(t, x, y, p))
    body = len(image) - 8 * len(t)
    for i in (3, 11, 200):
        image[body + 8 * i + 7] |= 0xE0
    with tempfile.NamedTemporaryFile(suffix='.dat', delete=False) as fh:
        fh.write(bytes(image))
        path = fh.name
    ev = dat_tools.load_td_data(path)
    os.unlink(path)
    inside = (ev['x'] < 304) & (ev['y'] < 240)
    ms = ref_micro_sum(ev['t'][inside], ev['x'][inside], ev['y'][inside], ev['p'][inside].astype('u1'), 4, 240, 304)
    save('events_dat', file_image=np.frombuffer(bytes(image), dtype=np.uint8), t=ev['t'], x=ev['x'], y=ev['y'], p=ev['p'],
         micro_sum=ms.astype(np.int32), oob=np.int64((~inside).sum()))


# ----------------------------------------------------------------------------- a5: the v_record output (embedding.py:141,180,223-224)
def gen_vrecord():
    """AdaptiveRSNNEmbedding.forward(..., v_record=True): the pre-reset potentials of the neurons that did NOT fire, step after step (the
    reference's debugging output) for three of the arsnn_* cases (same inputs and weights as those fixtures: only the new output is stored)"""
    from yolox.models.embedding import AdaptiveRSNNEmbedding
    from yolox.models.activation import Rectangle
    from yolox.utils.util import warp_decay
    from oracle.fill import poisson_events, procedural_fill_
    cases = {'readme': (5, 2, 1, 'sum', True, True, False, 4, 0.0, 2.0), 'soft_reset': (5, 2, 2, 'sum', True, True, False, 4, None, 3.0),
             'ts1_plain': (5, 1, 1, 'sum', False, False, False, 4, 0.0, 2.0), 'ts7_tm8': (5, 2, 7, 'sum', True, True, False, 8, 0.0, 4.0)}
    arrays = {}
    for name, (k, depth, Ts, readout, sat, wz, ab, Tm, vreset, gain) in cases.items():
        m = AdaptiveRSNNEmbedding(kernel_size=k, in_channel=2, out_channel=2, Ts=Ts, spike_attach=sat, write_zero=wz, abs=ab, depth=depth,
                                  readout=readout, nb_steps=Tm, vreset=vreset, thresh=1.0, spike_fn=Rectangle,
                                  decay=torch.nn.Parameter(warp_decay(0.5)))
        crc = procedural_fill_(m, conv_gain=gain)
        x = torch.from_numpy(poisson_events((2, 1, Tm, 2, 24, 32), 0.6, seed=zlib.crc32(name.encode()) % 1000))
        with torch.no_grad():
            out, v_rec = m(x, v_record=True)
        prev = np.load(os.path.join(OUT, f'arsnn_{name}.npz'))
        assert int(prev['crc']) == crc and np.array_equal(prev['x'], _np(x)) and np.array_equal(prev['out'], _np(out))
        arrays[f'{name}/v_record'] = _np(v_rec)
        print(f'    {name}: {v_rec.numel()} recorded potentials')
    save('vrecord_arsnn', **arrays)


# ----------------------------------------------------------------------------- a5/a7 embeddings
def gen_embeddings():
    from yolox.models.embedding import (AdaptiveRSNNEmbedding, LIFEmbedding, SpikeCountEmbedding,
                                        SpikingEmbedding)
    from yolox.models.activation import Rectangle
    from yolox.utils.util import warp_decay
    from oracle.fill import poisson_events, procedural_fill_

    def kwargs(Tm, thresh=1.0, vreset=0.0):
        return {'nb_steps': Tm, 'vreset': vreset, 'thresh': thresh, 'spike_fn': Rectangle,
                'decay': torch.nn.Parameter(warp_decay(0.5))}

    variants = {
        # name: (k, depth, Ts, readout, spike_attach, write_zero, abs, Tm, vreset, gain)
        'readme': (5, 2, 1, 'sum', True, True, False, 4, 0.0, 2.0),
        'ts3_sum': (5, 2, 3, 'sum', True, True, False, 4, 0.0, 3.0),
        'ts3_last': (5, 1, 3, 'last', False, False, False, 4, 0.0, 3.0),
        'ts2_avg': (7, 1, 2, 'avg', True, False, True, 6, 0.0, 3.0),
        'ts1_plain': (5, 1, 1, 'sum', False, False, False, 4, 0.0, 2.0),
        'soft_reset': (5, 2, 2, 'sum', True, True, False, 4, None, 3.0),
        'ts7_tm8': (5, 2, 7, 'sum', True, True, False, 8, 0.0, 4.0),
    }
    for name, (k, depth, Ts, readout, sat, wz, ab, Tm, vreset, gain) in variants.items():
        m = AdaptiveRSNNEmbedding(kernel_size=k, in_channel=2, out_channel=2, Ts=Ts, spike_attach=sat,
                                  write_zero=wz, abs=ab, depth=depth, readout=readout, **kwargs(Tm, 1.0, vreset))
        crc = procedural_fill_(m, conv_gain=gain)
        x = torch.from_numpy(poisson_events((2, 1, Tm, 2, 24, 32), 0.6, seed=zlib.crc32(name.encode()) % 1000))
        x.requires_grad_(True)
        out, t_rec = m(x, record=True)
        out = m(x)
        gw = torch.from_numpy(np.random.default_rng(5).standard_normal(out.shape).astype(np.float32))
        out.backward(gw)
        grads = {f'grad/{n}': _np(p.grad) for n, p in m.named_parameters() if p.grad is not None}
        save(f'arsnn_{name}', x=_np(x), out=_np(out), gout=_np(gw), gx=_np(x.grad), crc=np.uint32(crc),
             t_record=_np(t_rec).astype(np.int8),
             cfg=np.array([k, depth, Ts, int(sat), int(wz), int(ab), Tm, -1 if vreset is None else 0]),
             readout=np.array(readout), gain=np.float32(gain), **grads)

    # count / rsnn / snn
    Tm = 4
    x = torch.from_numpy(poisson_events((2, 1, Tm, 2, 24, 32), 0.6, seed=21))
    save('emb_count', x=_np(x), out=_np(SpikeCountEmbedding(Tm)(x)))
    for ro in ('sum', 'last'):
        m = SpikingEmbedding(kernel_size=5, in_channel=2, out_channel=2, readout=ro, relu=(ro == 'last'), depth=2,
                             **kwargs(Tm))
        crc = procedural_fill_(m, conv_gain=2.0)
        xx = x.clone().requires_grad_(True)
        out = m(xx)
        gw = torch.from_numpy(np.random.default_rng(6).standard_normal(out.shape).astype(np.float32))
        out.backward(gw)
        grads = {f'grad/{n}': _np(p.grad) for n, p in m.named_parameters() if p.grad is not None}
        save(f'emb_rsnn_{ro}', x=_np(x), out=_np(out), gout=_np(gw), gx=_np(xx.grad), crc=np.uint32(crc), **grads)
        m = LIFEmbedding(kernel_size=5, in_channel=2, out_channel=2, readout=ro, depth=2, **kwargs(Tm))
        crc = procedural_fill_(m, conv_gain=2.0)
        xx = x.clone().requires_grad_(True)
        out = m(xx)
        gw = torch.from_numpy(np.random.default_rng(6).standard_normal(out.shape).astype(np.float32))
        out.backward(gw)
        grads = {f'grad/{n}': _np(p.grad) for n, p in m.named_parameters() if p.grad is not None}
        save(f'emb_snn_{ro}', x=_np(x), out=_np(out), gout=_np(gw), gx=_np(xx.grad), crc=np.uint32(crc), **grads)


# ----------------------------------------------------------------------------- a9 cross-check from the in-repo LIFLayer
def gen_lif_layer():
    """The reference's own LIFLayer (layer.py:38-69 + cell.py:37-65) with the configuration quoted at
    utils_snn.py:41-43: the same recurrence as the PLIF node (sigmoid(decay)=0.5, soft reset, thresh 1)
    except '>' instead of '>=' at exact threshold -- inputs below avoid v == 1 exactly."""
    from yolox.models.layer import LIFLayer
    from yolox.models.activation import Rectangle
    from yolox.utils.util import warp_decay
    rng = np.random.default_rng(3)
    T, M = 5, 4096
    x = torch.from_numpy((rng.standard_normal((T, M)) * 1.0 + 0.5).astype(np.float32)).requires_grad_(True)
    lay = LIFLayer(retain_v=True, nb_steps=T, vreset=None, thresh=1.0, spike_fn=Rectangle,
                   decay=torch.nn.Parameter(warp_decay(0.5)))
    s = lay(x)
    g = torch.from_numpy(rng.standard_normal((T, M)).astype(np.float32))
    s.backward(g)
    save('lif_layer_inrepo', x=_np(x), spikes=_np(s), v_final=_np(lay.vmem), gout=_np(g), gx=_np(x.grad),
         gdecay=_np(lay.cell.decay.grad))


# ----------------------------------------------------------------------------- 8f rank 1: timestamp window search
def gen_window_search():
    """GEN1Dataset.search_events (gen1.py:217-232) with the real PSEELoader (psee_loader.py) on synthetic .dat recordings written
    to a temp dir: the event index range it returns for a set of label timestamps, incl. the cases its bisection treats specially
    (> 100 000 events: probes that hit a timestamp exactly; windows before the first / after the last event; empty windows that
    make it step back).  Stored: the timestamps of each recording (its x / y / p are regenerated from the seed by the tests) and
    (timestamp -> first event index, event count, first and last timestamp of the returned events)."""
    import types
    from yolox.data.datasets.gen1 import GEN1Dataset
    from yolox.utils.psee_loader.io import dat_events_tools as dat_tools
    from oracle import events_ref
    # NumPy 2 compatibility only: parse_header hands back the event size as a numpy uint8 scalar, and PSEELoader's
    # ``(end - start) // ev_size`` then raises under NumPy >= 2 (the reference pins 1.26, where the scalar is promoted).  The wrapper
    # converts the two header bytes to Python ints; no arithmetic changes.
    real_parse = dat_tools.parse_header
    dat_tools.parse_header = lambda f: tuple(int(v) if isinstance(v, np.integer) else v for v in real_parse(f))
    rng = np.random.default_rng(77)
    arrays, names = {}, []
    recs = {
        # name: (timestamps, window, num_slice)
        'long': ((5_000 + np.cumsum(rng.integers(0, 25, 250_000))).astype(np.uint32), (-200000, 0), 1),     # gaps 0..24 us: many equal timestamps
        'gappy': (np.concatenate([np.sort(rng.integers(1_000, 50_000, 3000)), np.sort(rng.integers(2_000_000, 2_050_000, 3000))]).astype(np.uint32),
                  (-100000, 0), 4),
        'short': (np.sort(rng.integers(10, 90_000, 500)).astype(np.uint32), (-50000, 0), 1),
    }
    for name, (t, window, num_slice) in recs.items():
        root = tempfile.mkdtemp(prefix='eas_dat_')
        n = len(t)
        x = rng.integers(0, 304, n).astype(np.uint16)
        y = rng.integers(0, 240, n).astype(np.uint16)
        p = rng.integers(0, 2, n).astype(np.uint8)
        with open(os.path.join(root, f'{name}_td.dat'), 'wb') as fh:
            fh.write(events_ref.encode_dat_file(t, x, y, p))
        me = types.SimpleNamespace(files=[os.path.join(root, f'{name}_bbox.npy')], slice_policy='fix_t',
                                   slice_args={'window': window, 'num_slice': num_slice})
        stamps = [int(t[0]), int(t[-1]), int(t[-1]) + 1, int(t[-1]) + 150_000, int(t[-1]) + 10_000_000, 1, 50, int(t[n // 2]),
                  int(t[n // 2]) + 1, int(t[n // 3]) - window[0], 1_000_000, 1_234_567, 2_500_000, 3_200_000]
        if n > 200_000:
            # a label whose window start equals the timestamp of a bisection probe: middle = n // 2 is the first probe
            stamps += [int(t[n // 2]) - window[0], int(t[n // 4]) - window[0], int(t[n // 2 + n // 4]) - window[0]]
        out = []
        for ts in stamps:
            ev = GEN1Dataset.search_events(me, 0, ts)
            if len(ev):
                first = int(np.searchsorted(t, ev['t'][0], side='left'))
                while not (np.array_equal(t[first:first + len(ev)], ev['t']) and np.array_equal(x[first:first + len(ev)], ev['x'])):
                    first += 1                      # equal timestamps: find the run by content
                out.append((ts, first, len(ev), int(ev['t'][0]), int(ev['t'][-1])))
            else:
                out.append((ts, -1, 0, 0, 0))
        arrays[f'{name}/t0'] = np.uint32(t[0])
        arrays[f'{name}/gaps'] = np.diff(t).astype(np.uint32 if np.diff(t).max() > 255 else np.uint8)     # timestamps as gaps (compresses)
        arrays[f'{name}/cfg'] = np.array([window[0], window[1], num_slice], np.int64)
        arrays[f'{name}/result'] = np.array(out, np.int64)
        names.append(name)
        print(f'    {name}: {n} events, {sum(1 for o in out if o[2])} non-empty of {len(out)} searches')
    dat_tools.parse_header = real_parse
    arrays['cases'] = np.array(names)
    save('window_search', **arrays)


# ----------------------------------------------------------------------------- 8f rank 3: SOP / energy estimate
def gen_energy():
    """EventEvaluator.energy_estimation (yolox/evaluators/event_evaluator.py:466-565) itself, run on the CPU: its CUDA-only tensor
    moves are patched to the identity for the duration of the call (Tensor.cuda, torch.cuda.FloatTensor -> the CPU type; no
    arithmetic in them) and the figures it prints are captured at full precision (Tensor.__format__ -> repr of the float).  The model
    is the reference's own SYOLOX-S from EventExp.get_model with the procedural fill; two batches, NO reset between them (the
    method does not reset the neurons between batches)."""
    import re
    import types
    import yolox.evaluators.event_evaluator as ee
    from yolox.exp import get_exp
    from spikingjelly.activation_based import functional
    from oracle.fill import ANN_KEYS, poisson_events, procedural_fill_
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(['T', '3', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
               'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'spike_fn', 'atan', 'use_spike', 'True'])
    torch.manual_seed(80)
    model = exp.get_model()
    crc = procedural_fill_(model, 2.0, ann_regex=ANN_KEYS['True'])
    seeds = (5, 6)
    batches = [(torch.from_numpy(poisson_events((3, 1, 4, 2, 64, 96), 0.5, seed=s_)), None, None, None) for s_ in seeds]
    lines = []
    saved = (torch.Tensor.cuda, torch.cuda.FloatTensor, torch.Tensor.__format__, getattr(ee, 'print', None))
    try:
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.cuda.FloatTensor = torch.FloatTensor
        torch.Tensor.__format__ = lambda self, spec: repr(float(self)) if self.dim() == 0 else object.__format__(self, spec)
        ee.print = lambda *a, **k: lines.append(' '.join(str(v) for v in a))
        ee.EventEvaluator.energy_estimation(types.SimpleNamespace(dataloader=batches), model, exp)
    finally:
        torch.Tensor.cuda, torch.cuda.FloatTensor, torch.Tensor.__format__ = saved[:3]
        if saved[3] is None:
            del ee.print
        else:
            ee.print = saved[3]
    functional.reset_net(model)
    text = '\n'.join(lines)
    num = r'([-+0-9.eE]+|inf|nan)'
    tot = re.search(r'SOP in SNN: ' + num + r', SOP in ANN ' + num, text)
    arrays = {'sop_snn_G': np.float64(tot.group(1)), 'sop_ann_G': np.float64(tot.group(2)),
              'snn_energy': np.float64(re.search(r'SNN Energy:' + num, text).group(1)),
              'ann_energy': np.float64(re.search(r'ANN Energy:' + num, text).group(1))}
    for key in ('embedding', 'backbone', 'fpn', 'head'):
        m = re.search(key + r': SOP in SNN: ' + num + r', SOP in ANN ' + num, text)
        arrays[f'ac_G/{key}'] = np.float64(m.group(1))
        arrays[f'mac_G/{key}'] = np.float64(m.group(2))
    save('energy_estimation', seeds=np.array(seeds), shape=np.array((3, 1, 4, 2, 64, 96)), crc=np.uint32(crc), T=np.int64(exp.T),
         printed=np.array(text), **arrays)
    print(text)


# ----------------------------------------------------------------------------- config-4 input: RVT stacked histogram
def gen_stacked_hist():
    """RVTGEN4Dataset.generate_slices(file, time, num_slice, 'event_sum') (rvt_gen4.py:109-125) called unbound on a synthetic
    representation directory: the index arrays are real .npy files, the HDF5 container is stood in for by an .npy next to it."""
    import types
    from yolox.data.datasets.rvt_gen4 import RVTGEN4Dataset
    rng = np.random.default_rng(44)
    rep_name = 'stacked_histogram_dt=50_nbins=10'
    arrays = {}
    names = []
    for name, (H, W, n_repr, nbins, hi) in {'small': (12, 16, 9, 10, 4), 'saturated': (6, 32, 6, 10, 256), 'ragged_w': (5, 21, 5, 10, 7),
                                            'bins3': (4, 16, 5, 3, 50)}.items():
        root = tempfile.mkdtemp(prefix='eas_rvt_')
        rep_dir = os.path.join(root, 'event_representations_v2', rep_name)
        os.makedirs(rep_dir)
        data = rng.integers(0, hi, (n_repr, 2 * nbins, H, W)).astype(np.uint8)
        if name == 'saturated':
            data[2:] = 255                                       # sums of 2550 must not wrap around in 8 bits
        obj2repr = np.arange(n_repr, dtype=np.int64)
        np.save(os.path.join(rep_dir, 'objframe_idx_2_repr_idx.npy'), obj2repr)
        np.save(os.path.join(rep_dir, 'timestamps_us.npy'), (np.arange(n_repr) + 1) * 50000)
        np.save(os.path.join(rep_dir, 'event_representations_ds2_nearest.h5.npy'), data)
        me = types.SimpleNamespace(files=[root], rep_name=rep_name, img_size=(H, W))
        arrays[f'{name}/data'] = data
        for time, num_slice in ((n_repr - 1, 4), (1, 4), (0, 3), (3, 1), (2, 3)):
            out = RVTGEN4Dataset.generate_slices(me, 0, time, num_slice, 'event_sum')
            key = f'{name}/t{time}_n{num_slice}'
            arrays[key] = out
            names.append(key)
            assert out.shape == (1, num_slice, 2, H, W) and out.dtype == np.float64
    arrays['cases'] = np.array(names)
    save('stacked_hist', **arrays)


# ----------------------------------------------------------------------------- a10 'patan'
def gen_patan():
    """The reference's own EfficientNoisySpikeII(InvArcTanh(alpha), p=0) (activation.py:121-130,181-205; what 'patan' selects at
    event_yolox_base.py:145-150) as the surrogate of a ParametricLIFNode configured as utils_snn.py:44-53: spikes, input gradient,
    dL/dw and dL/dalpha for train mode (positive and negative alpha), spikes for eval mode; then one converted BaseConv and a
    whole SYOLOX-S built by the reference's EventExp with spike_fn='patan'."""
    from yolox.models.activation import EfficientNoisySpikeII, InvArcTanh
    from yolox.models.network_blocks import BaseConv
    from yolox.utils.utils_snn import convert_to_spiking
    from spikingjelly.activation_based import functional, neuron
    from oracle.fill import procedural_fill_
    rng = np.random.default_rng(21)
    T, M = 4, 4096
    xn = (rng.standard_normal((T, M)) * 0.9 + 0.45).astype(np.float32)
    xn[0, :64] = 1.0                      # exact-threshold inputs (v = 0 * 0.5 + 1.0 -> u = 0 -> fires under '>=')
    xn[1, 64:128] = 0.5
    g = rng.standard_normal((T, M)).astype(np.float32)
    for tag, alpha, detach in (('pos', 2.0, False), ('neg', -1.5, False), ('detach', 2.0, True)):
        sf = EfficientNoisySpikeII(InvArcTanh(alpha), p=0)
        node = neuron.ParametricLIFNode(init_tau=2.0, decay_input=False, v_threshold=1.0, v_reset=None, surrogate_function=sf,
                                        detach_reset=detach, step_mode='m', backend='torch')
        node.train()
        x = torch.from_numpy(xn).requires_grad_(True)
        s = node(x)
        s.backward(torch.from_numpy(g))
        arrays = dict(x=xn, gout=g, spikes=_np(s), gx=_np(x.grad), gw=_np(node.w.grad), galpha=_np(sf.inv_sg.alpha.grad),
                      alpha=np.float32(alpha), detach=np.bool_(detach), v_final=_np(node.v))
        functional.reset_net(node)
        node.eval()
        with torch.no_grad():
            arrays['spikes_eval'] = _np(node(torch.from_numpy(xn)))
        assert np.array_equal(arrays['spikes_eval'], arrays['spikes']), 'train-mode forward of patan is not the exact Heaviside step'
        save(f'lif_patan_{tag}', **arrays)

    # one converted block, train mode
    mod = convert_to_spiking(BaseConv(8, 16, 1, 1), EfficientNoisySpikeII(InvArcTanh(2.0), p=0))
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    crc = procedural_fill_(mod, conv_gain=2.5)
    mod.train()
    xb = (rng.random((3, 2, 8, 12, 16)) < 0.3).astype(np.float32)
    xx = torch.from_numpy(xb).requires_grad_(True)
    out = mod(xx)
    gb = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(gb))
    extra = {f'grad/{n}': _np(p.grad) for n, p in mod.named_parameters()}
    extra.update({f'buf/{n}': _np(b) for n, b in mod.named_buffers()})
    save('block_baseconv1x1_patan_train', x=xb, out=_np(out), gout=gb, gx=_np(xx.grad), crc=np.uint32(crc),
         keys=np.array(list(mod.state_dict().keys())), **extra)

    # whole model through the reference's EventExp
    from yolox.exp import get_exp
    from oracle.fill import ANN_KEYS, poisson_events
    exp = get_exp(None, 'e-yolox-s')
    exp.merge(['T', '3', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
               'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'spike_fn', 'patan', 'use_spike', 'True'])
    torch.manual_seed(80)
    model = exp.get_model()
    crc = procedural_fill_(model, conv_gain=2.0, ann_regex=ANN_KEYS['True'])
    x = torch.from_numpy(poisson_events((2, 1, exp.Tm, 2, 64, 64), 0.5, seed=17))
    model.eval()
    with torch.no_grad():
        logits = model(x)
    functional.reset_net(model)
    arrays = dict(x=_np(x), logits=_np(logits), crc=np.uint32(crc), keys=np.array(list(model.state_dict().keys())), gain=np.float32(2.0),
                  nparam=np.int64(sum(p.numel() for p in model.parameters())))
    model.train()
    model.head.use_l1 = True
    tg = torch.zeros(2, 50, 5)
    for b in range(2):
        tg[b, 0] = torch.tensor([0, 64 * 0.3, 64 * 0.4, 64 * 0.25, 64 * 0.3])
        tg[b, 1] = torch.tensor([1, 64 * 0.7, 64 * 0.6, 64 * 0.2, 64 * 0.35])
    out = model(x, tg)
    out['total_loss'].backward()
    functional.reset_net(model)
    for k in ('total_loss', 'iou_loss', 'l1_loss', 'conf_loss', 'cls_loss'):
        arrays[f'loss/{k}'] = np.float32(float(out[k]))
    arrays['loss/num_fg'] = np.float32(out['num_fg'])
    arrays['targets'] = _np(tg)
    gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
    arrays['gradnorm_keys'] = np.array(list(gn))
    arrays['gradnorm_vals'] = np.array(list(gn.values()), np.float32)
    for n, p in model.named_parameters():
        if n.endswith('inv_sg.alpha') or n.endswith('act.w'):
            arrays[f'grad/{n}'] = _np(p.grad)
    opt_groups = [len(gr['params']) for gr in exp.get_optimizer(2).param_groups]
    arrays['optimizer_group_sizes'] = np.array(opt_groups)
    save('model_s_true_patan_64', **arrays)
    print('    optimizer groups', opt_groups, ' alpha grads', sum(1 for k in arrays if k.startswith('grad/') and k.endswith('alpha')))


# ----------------------------------------------------------------------------- a12 blocks
def gen_blocks():
    from yolox.models.network_blocks import BaseConv, CSPLayer, SPPBottleneck
    from yolox.utils.utils_snn import convert_to_spiking
    from spikingjelly.activation_based import surrogate, functional
    from oracle.fill import procedural_fill_
    rng = np.random.default_rng(9)

    def run(name, mod, x, train):
        convert_to_spiking(mod, surrogate.ATan(2.0))
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03
        crc = procedural_fill_(mod, conv_gain=2.5)
        mod.train(train)
        xx = torch.from_numpy(x).requires_grad_(True)
        out = mod(xx)
        extra = {}
        if train:
            g = torch.from_numpy(rng.standard_normal(out.shape).astype(np.float32))
            out.backward(g)
            extra = {'gout': _np(g), 'gx': _np(xx.grad)}
            extra.update({f'grad/{n}': _np(p.grad) for n, p in mod.named_parameters()})
            extra.update({f'buf/{n}': _np(b) for n, b in mod.named_buffers()})
        functional.reset_net(mod)
        save(name, x=x, out=_np(out), crc=np.uint32(crc), **extra)

    T, N = 3, 2
    spikes_in = (rng.random((T, N, 8, 12, 16)) < 0.3).astype(np.float32)
    real_in = rng.standard_normal((T, N, 8, 12, 16)).astype(np.float32)
    for train in (True, False):
        tag = 'train' if train else 'eval'
        run(f'block_baseconv1x1_{tag}', BaseConv(8, 16, 1, 1), spikes_in, train)
        run(f'block_baseconv3x3s2_{tag}', BaseConv(8, 16, 3, 2), real_in, train)
        run(f'block_csp_{tag}', CSPLayer(8, 8, n=2), spikes_in, train)
        run(f'block_spp_{tag}', SPPBottleneck(8, 8), spikes_in, train)


# ----------------------------------------------------------------------------- a8/a12/a13 whole model
def gen_models():
    from yolox.exp import get_exp
    from spikingjelly.activation_based import functional
    from oracle.fill import ANN_KEYS, poisson_events, procedural_fill_

    base_opts = ['T', '3', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1',
                 'readout', 'sum', 'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True',
                 'spike_fn', 'atan']

    def build(exp_name, use_spike, extra=()):
        exp = get_exp(None, exp_name)
        exp.merge(base_opts + ['use_spike', use_spike] + list(extra))
        torch.manual_seed(80)
        return exp, exp.get_model()

    def targets_for(B, H, W):
        t = torch.zeros(B, 50, 5)
        for b in range(B):
            t[b, 0] = torch.tensor([0, W * 0.3, H * 0.4, W * 0.25, H * 0.3])
            t[b, 1] = torch.tensor([1, W * 0.7, H * 0.6, W * 0.2, H * 0.35])
        return t

    runs = [
        # name, exp, use_spike, extra opts, (B, H, W), conv gain
        ('model_s_true_64', 'e-yolox-s', 'True', (), (2, 64, 64), 2.0),
        ('model_s_full_64', 'e-yolox-s', 'full_spike', (), (2, 64, 64), 2.0),
        ('model_s_fullv2_64', 'e-yolox-s', 'full_spike_v2', (), (2, 64, 64), 2.0),
        ('model_s_false_64', 'e-yolox-s', 'False', (), (1, 64, 64), 2.0),
        ('model_s_true_256x320', 'e-yolox-s', 'True', (), (1, 256, 320), 2.0),
        ('model_m_fullv2_t5_64x96', 'e-yolox-m', 'full_spike_v2', ('T', '5'), (1, 64, 96), 2.0),
        ('model_s_true_ts3_64', 'e-yolox-s', 'True', ('Ts', '3'), (1, 64, 64), 2.0),
    ]
    for name, exp_name, us, extra, (B, H, W), gain in runs:
        exp, model = build(exp_name, us, extra)
        nparam = sum(p.numel() for p in model.parameters())
        crc = procedural_fill_(model, conv_gain=gain, ann_regex=ANN_KEYS[us])
        x = torch.from_numpy(poisson_events((B, 1, exp.Tm, 2, H, W), 0.5, seed=zlib.crc32(name.encode()) % 1000))
        keys = np.array(list(model.state_dict().keys()))
        model.eval()
        with torch.no_grad():
            logits = model(x)
        functional.reset_net(model)
        arrays = dict(x=_np(x), logits=_np(logits), crc=np.uint32(crc), nparam=np.int64(nparam), keys=keys,
                      gain=np.float32(gain))
        if name in ('model_s_true_64', 'model_s_fullv2_64'):
            model.train()
            model.head.use_l1 = True
            tg = targets_for(B, H, W)
            out = model(x, tg)
            out['total_loss'].backward()
            functional.reset_net(model)
            for k in ('total_loss', 'iou_loss', 'l1_loss', 'conf_loss', 'cls_loss'):
                arrays[f'loss/{k}'] = np.float32(float(out[k]))
            arrays['loss/num_fg'] = np.float32(out['num_fg'])
            arrays['targets'] = _np(tg)
            picks = [n for n, p in model.named_parameters() if p.grad is not None and (
                n.endswith('act.w') or 'embedding' in n or n.endswith('bn.weight') or p.numel() <= 4096)]
            for n, p in model.named_parameters():
                if n in picks:
                    arrays[f'grad/{n}'] = _np(p.grad)
            # every grad as a norm: cheap whole-model coverage
            gn = {n: float(p.grad.norm()) for n, p in model.named_parameters() if p.grad is not None}
            arrays['gradnorm_keys'] = np.array(list(gn))
            arrays['gradnorm_vals'] = np.array(list(gn.values()), np.float32)
            # BN running stats after one train step
            for n, b in model.named_buffers():
                if n.endswith('running_mean') and ('dark2' in n or 'dark5' in n):
                    arrays[f'buf/{n}'] = _np(b)
        save(name, **arrays)
        print(f'    {name}: params {nparam / 1e6:.3f} M, logits {tuple(logits.shape)} absmax {float(logits.abs().max()):.3g} '
              f'finite {bool(torch.isfinite(logits).all())}')


def gen_model_train_m():
    """BASELINE configs[2] shaped train step (SYOLOX-M, full_spike_v2, T=5, RPD) of the UNMODIFIED reference classes on a 64x96 canvas:
    losses, the norm of every parameter gradient, every small gradient in full and a strided sample (<= 1024 elements) of every
    gradient for the elementwise comparison in tests/test_gpu_model.py::test_model_train_step_golden."""
    from yolox.exp import get_exp
    from spikingjelly.activation_based import functional
    from oracle.fill import ANN_KEYS, poisson_events, procedural_fill_
    name, us, (B, H, W) = 'model_m_fullv2_t5_64x96_train', 'full_spike_v2', (2, 64, 96)
    exp = get_exp(None, 'e-yolox-m')
    exp.merge(['T', '5', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
               'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'spike_fn', 'atan', 'use_spike', us])
    torch.manual_seed(80)
    model = exp.get_model()
    crc = procedural_fill_(model, conv_gain=2.0, ann_regex=ANN_KEYS[us])
    x = torch.from_numpy(poisson_events((B, 1, exp.Tm, 2, H, W), 0.5, seed=zlib.crc32(name.encode()) % 1000))
    tg = torch.zeros(B, 50, 5)
    for b in range(B):
        tg[b, 0] = torch.tensor([0, W * 0.3, H * 0.4, W * 0.25, H * 0.3])
        tg[b, 1] = torch.tensor([1, W * 0.7, H * 0.6, W * 0.2, H * 0.35])
    model.train()
    model.head.use_l1 = True
    out = model(x, tg)
    out['total_loss'].backward()
    functional.reset_net(model)
    arrays = dict(x=_np(x), targets=_np(tg), crc=np.uint32(crc), keys=np.array(list(model.state_dict().keys())), gain=np.float32(2.0))
    for k in ('total_loss', 'iou_loss', 'l1_loss', 'conf_loss', 'cls_loss'):
        arrays[f'loss/{k}'] = np.float32(float(out[k]))
    arrays['loss/num_fg'] = np.float32(out['num_fg'])
    gn = {}
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        g = p.grad.detach().reshape(-1)
        gn[n] = float(g.norm())
        if g.numel() <= 1024:
            arrays[f'grad/{n}'] = _np(p.grad)
        else:
            arrays[f'gradsample/{n}'] = _np(g[::g.numel() // 1024][:1024])
    arrays['gradnorm_keys'] = np.array(list(gn))
    arrays['gradnorm_vals'] = np.array(list(gn.values()), np.float32)
    for n, b in model.named_buffers():
        if n.endswith('running_mean') and ('dark2' in n or 'dark5' in n or 'head.stems' in n):
            arrays[f'buf/{n}'] = _np(b)
    save(name, **arrays)
    print(f'    {name}: loss {float(out["total_loss"]):.5f}, {len(gn)} parameter gradients')


def main():
    torch.set_num_threads(8)
    setup_reference_imports()
    which = sys.argv[1:] or ['events', 'reps', 'augment', 'embeddings', 'lif', 'patan', 'stacked_hist', 'energy', 'window', 'blocks', 'models']
    for w in which:
        print(f'[{w}]')
        {'events': gen_events, 'embeddings': gen_embeddings, 'lif': gen_lif_layer, 'blocks': gen_blocks,
         'models': gen_models, 'reps': gen_event_reps, 'augment': gen_augment, 'patan': gen_patan, 'stacked_hist': gen_stacked_hist, 'energy': gen_energy, 'window': gen_window_search, 'model_train_m': gen_model_train_m, 'vrecord': gen_vrecord}[w]()
    assert not os.path.exists(os.path.join(REF, 'yolox', '__pycache__')), 'bytecode leaked into the reference'


if __name__ == '__main__':
    main()
