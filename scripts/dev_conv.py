#!/usr/bin/env python3
"""Development check of the MFMA convolution kernels: accuracy against an fp64 reference and timing against
torch (MIOpen) at the bench shapes.  usage: dev_conv.py [acc|time|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import eas_snn_amd  # noqa
from eas_snn_amd import ops

dev = torch.device('cuda:0')


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def make(NI, Cin, Cout, H, W, k, spikes, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    if spikes:
        x = torch.randint(0, 3, (NI, Cin, H, W), generator=g).float()
    else:
        x = torch.randn(NI, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    return x.to(dev), w.to(dev)


def acc():
    cases = [  # NI, Cin, Cout, H, W, k, stride, spikes
        (2, 16, 32, 8, 10, 3, 1, True), (2, 16, 32, 8, 10, 3, 1, False), (3, 32, 64, 16, 20, 3, 1, True), (1, 64, 64, 32, 40, 3, 1, False),
        (2, 24, 40, 12, 16, 3, 1, False), (2, 8, 32, 16, 16, 3, 1, False), (2, 64, 128, 16, 20, 3, 2, True), (2, 32, 64, 16, 20, 3, 2, False),
        (3, 64, 32, 8, 10, 1, 1, True), (2, 128, 64, 16, 20, 1, 1, False), (2, 32, 32, 64, 80, 3, 1, True), (5, 128, 128, 16, 20, 3, 1, True),
        (4, 256, 256, 8, 10, 3, 1, True), (1, 8, 5, 7, 10, 3, 1, False), (2, 128, 2, 8, 10, 1, 1, False), (3, 40, 72, 10, 12, 3, 2, False)]
    for (NI, Cin, Cout, H, W, k, s, sp) in cases:
        x, w = make(NI, Cin, Cout, H, W, k, sp)
        ref = F.conv2d(x.double().cpu(), w.double().cpu(), stride=s, padding=k // 2)
        pk = ops.conv_pack_weights(w, 0)
        y = ops.conv_fwd_packed(x, pk, None, Cout, k, s, 1 if sp else 3)
        y32 = F.conv2d(x, w, stride=s, padding=k // 2)
        scale = ref.abs().max().item()
        e_own = (y.double().cpu() - ref).abs().max().item() / scale
        e_mio = (y32.double().cpu() - ref).abs().max().item() / scale
        print(f'fwd NI={NI} {Cin:3d}->{Cout:3d} {H}x{W} k{k} s{s} {"spk" if sp else "f32"}: max err / max|y|  own {e_own:.2e}   miopen-fp32 {e_mio:.2e}',
              flush=True)
    print('inexact flag:', int(ops.conv_inexact_flag(dev)))


def wacc():
    cases = [(2, 16, 32, 8, 10, 1, True), (2, 32, 32, 16, 20, 1, False), (3, 64, 64, 32, 40, 1, True), (2, 64, 128, 16, 20, 1, True),
             (2, 32, 64, 64, 80, 1, True), (4, 128, 128, 8, 10, 1, True), (2, 64, 128, 32, 40, 2, True), (2, 32, 64, 16, 20, 2, False),
             (5, 24, 40, 12, 16, 1, False), (3, 128, 64, 16, 20, 1, False), (3, 128, 64, 16, 20, 1, False, 1), (2, 64, 32, 64, 80, 1, True, 1),
             (4, 1024, 512, 8, 10, 1, True, 1), (2, 40, 24, 8, 12, 1, False, 1)]
    for case in cases:
        (NI, Cin, Cout, H, W, s, sp), kk = case[:7], (case[7] if len(case) > 7 else 3)
        x, w = make(NI, Cin, Cout, H, W, kk, sp)
        Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
        gy = torch.randn(NI, Cout, Ho, Wo, device=dev)
        ref = torch.nn.grad.conv2d_weight(x.double().cpu(), w.shape, gy.double().cpu(), stride=s, padding=kk // 2)
        gw = ops.conv_wgrad(x, gy, kk, s, 1 if sp else 3)
        g32 = torch.nn.grad.conv2d_weight(x, w.shape, gy, stride=s, padding=kk // 2)
        scale = ref.abs().max().item()
        print(f'wgrad k{kk} NI={NI} {Cin:3d}->{Cout:3d} {H}x{W} s{s} {"spk" if sp else "f32"}: max err / max|gw|  own '
              f'{(gw.double().cpu() - ref).abs().max().item() / scale:.2e}   miopen-fp32 {(g32.double().cpu() - ref).abs().max().item() / scale:.2e}', flush=True)


def wtime():
    for (name, NI, Cin, Cout, H, W, k, s, sp) in bench_shapes():
        x, w = make(NI, Cin, Cout, H, W, k, sp)
        Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
        gy = torch.randn(NI, Cout, Ho, Wo, device=dev)
        t_own = timeit(lambda: ops.conv_wgrad(x, gy, k, s, 1 if sp else 3))
        t_mio = timeit(lambda: torch.nn.grad.conv2d_weight(x, w.shape, gy, stride=s, padding=k // 2))
        fl = 2.0 * NI * Cout * Ho * Wo * Cin * k * k
        print(f'wgrad {name:22s} own {t_own:7.3f} ms ({fl / t_own / 1e9:7.1f} TF fp32-equiv) | miopen {t_mio:7.3f} ms ({fl / t_mio / 1e9:6.1f} TF)', flush=True)


def bench_shapes():
    return [  # name, NI, Cin, Cout, H, W, k, s, spikes
        ('dark2.m.conv2', 192, 32, 32, 64, 80, 3, 1, True), ('dark3.m.conv2', 192, 64, 64, 32, 40, 3, 1, True),
        ('dark4.m.conv2', 192, 128, 128, 16, 20, 3, 1, True), ('dark5.m.conv2', 192, 256, 256, 8, 10, 3, 1, True),
        ('head.cls0 (ann)', 64, 128, 128, 32, 40, 3, 1, False), ('C3_p4.m (ann)', 64, 128, 128, 16, 20, 3, 1, False),
        ('dark3.0 s2', 192, 64, 128, 64, 80, 3, 2, True), ('dark4.0 s2', 192, 128, 256, 32, 40, 3, 2, True),
        ('dark5.0 s2', 192, 256, 512, 16, 20, 3, 2, True),
        ('dark3.conv3 1x1', 192, 128, 128, 32, 40, 1, 1, True), ('dark2.conv1 1x1', 192, 64, 32, 64, 80, 1, 1, True),
        ('dark5.spp.conv2 1x1', 192, 1024, 512, 8, 10, 1, 1, True),
        ('dgrad dark2.m', 192, 32, 32, 64, 80, 3, 1, False), ('dgrad dark3.m', 192, 64, 64, 32, 40, 3, 1, False),
        ('dgrad dark4.m', 192, 128, 128, 16, 20, 3, 1, False), ('dgrad dark5.m', 192, 256, 256, 8, 10, 3, 1, False),
        ('head 8x10 (ann)', 64, 128, 128, 8, 10, 3, 1, False), ('neck 8x10 (ann)', 64, 256, 256, 8, 10, 3, 1, False),
        ('dark2.0 s2 (real x)', 64, 32, 64, 128, 160, 3, 2, False), ('stem (real x)', 64, 8, 32, 128, 160, 3, 1, False)]


def time_():
    only = [int(a) for a in os.environ['EAS_SHAPES'].split(',')] if os.environ.get('EAS_SHAPES') else None
    for idx, (name, NI, Cin, Cout, H, W, k, s, sp) in enumerate(bench_shapes()):
        if only is not None and idx not in only:
            continue
        x, w = make(NI, Cin, Cout, H, W, k, sp)
        pk = ops.conv_pack_weights(w, 0)
        t_own = timeit(lambda: ops.conv_fwd_packed(x, pk, None, Cout, k, s, 1 if sp else 3))
        t_pack = timeit(lambda: ops.conv_pack_weights(w, 0))
        t_mio = timeit(lambda: F.conv2d(x, w, stride=s, padding=k // 2))
        Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
        fl = 2.0 * NI * Cout * Ho * Wo * Cin * k * k
        print(f'{name:22s} own {t_own:7.3f} ms ({fl / t_own / 1e9:7.1f} TF fp32-equiv) pack {t_pack:6.3f} ms | miopen {t_mio:7.3f} ms '
              f'({fl / t_mio / 1e9:6.1f} TF)', flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('acc', 'all'):
        acc()
    if what in ('time', 'all'):
        time_()
    if what in ('wacc', 'wall'):
        wacc()
    if what in ('wtime', 'wall'):
        wtime()


def one(idx, reps=5):
    name, NI, Cin, Cout, H, W, k, s, sp = bench_shapes()[idx]
    x, w = make(NI, Cin, Cout, H, W, k, sp)
    pk = ops.conv_pack_weights(w, 0)
    for _ in range(reps):
        ops.conv_fwd_packed(x, pk, None, Cout, k, s, 1 if sp else 3)
    torch.cuda.synchronize()
    print('ran', name)


def ablate():
    for idx in [int(a) for a in os.environ.get('EAS_SHAPES', '0,1,2,4').split(',')]:
        name, NI, Cin, Cout, H, W, k, s, sp = bench_shapes()[idx]
        x, w = make(NI, Cin, Cout, H, W, k, sp)
        pk = ops.conv_pack_weights(w, 0)
        try:
            t = timeit(lambda: ops.conv_fwd_packed(x, pk, None, Cout, k, s, 1 if sp else 3), reps=20)
        except Exception:
            t = float('nan')
        print(f'dbg={os.environ.get("EAS_CONV_DBG", "0")} {name:20s} {t * 1e3:8.1f} us', flush=True)


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[1] == 'one':
    one(int(sys.argv[2]))
if __name__ == '__main__' and sys.argv[1:] == ['ablate']:
    ablate()


def c1time():
    """1x1 forward / input-gradient shapes with many channels (dark4 / dark5)"""
    shapes = [('dark5.spp.conv2 fwd', 192, 1024, 512, 8, 10, True), ('dark5.conv3 fwd', 192, 512, 512, 8, 10, True),
              ('dark5.conv1 fwd', 192, 512, 256, 8, 10, True), ('dark5.m.conv1 fwd', 192, 256, 256, 8, 10, True),
              ('dark5.spp.conv2 dgrad', 192, 512, 1024, 8, 10, False), ('dark5.conv3 dgrad', 192, 512, 512, 8, 10, False),
              ('dark4.conv3 fwd', 192, 256, 256, 16, 20, True), ('dark4.conv3 dgrad', 192, 256, 256, 16, 20, False),
              ('dark4.conv1 fwd', 192, 256, 128, 16, 20, True)]
    for name, NI, Cin, Cout, H, W, sp in shapes:
        x, w = make(NI, Cin, Cout, H, W, 1, sp)
        pk = ops.conv_pack_weights(w, 0)
        t = timeit(lambda: ops.conv_fwd_packed(x, pk, None, Cout, 1, 1, 1 if sp else 3))
        mb = 4 * NI * H * W * (Cin + Cout) / 1e6
        print(f'{name:24s} {t * 1e3:7.1f} us   {mb / t / 1e3:6.2f} TB/s algorithmic ({mb:.0f} MB)', flush=True)


if __name__ == '__main__' and sys.argv[1:] == ['c1time']:
    c1time()


def tile_acc():
    """development: forward of the EAS_SHAPES layers with the tile forced by EAS_CONV_TILE against fp64 (max error / max |y|)"""
    for idx in [int(a) for a in os.environ.get('EAS_SHAPES', '16,5,17,2').split(',')]:
        name, NI, Cin, Cout, H, W, k, s, sp = bench_shapes()[idx]
        x, w = make(NI, Cin, Cout, H, W, k, sp)
        pk = ops.conv_pack_weights(w, 0)
        try:
            y = ops.conv_fwd_packed(x, pk, None, Cout, k, s, 1 if sp else 3)
        except Exception as e:
            print(f'tile={os.environ.get("EAS_CONV_TILE")} {name:20s} unsupported ({type(e).__name__})')
            continue
        ref = F.conv2d(x[:8].double(), w.double(), stride=s, padding=k // 2)
        err = (y[:8].double() - ref).abs().max().item() / ref.abs().max().item()
        t = timeit(lambda: ops.conv_fwd_packed(x, pk, None, Cout, k, s, 1 if sp else 3), reps=20)
        print(f'tile={os.environ.get("EAS_CONV_TILE")} {name:20s} {t * 1e3:8.1f} us  err {err:.1e}', flush=True)


if __name__ == '__main__' and sys.argv[1:] == ['tile_acc']:
    tile_acc()
