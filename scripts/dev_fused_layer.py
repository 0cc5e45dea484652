"""development: one converted BaseConv in eval mode on spike planes -- the fused conv -> BN -> LIF kernel against conv + BN+LIF (time per call)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eas_snn_amd  # noqa: E402
from eas_snn_amd import ops  # noqa: E402
from spikingjelly.activation_based import functional, surrogate  # noqa: E402
from yolox.models.network_blocks import BaseConv  # noqa: E402
from yolox.utils.utils_snn import convert_to_spiking  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [(3, 64, 64, 64, 32, 40, 3, 1), (3, 64, 128, 128, 16, 20, 3, 1), (3, 64, 64, 128, 64, 80, 3, 2), (3, 64, 32, 32, 64, 80, 3, 1),
          (3, 64, 64, 64, 64, 80, 1, 1), (3, 64, 128, 128, 32, 40, 1, 1), (3, 64, 512, 256, 8, 10, 1, 1), (3, 64, 256, 256, 8, 10, 3, 1)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in sys.argv[1].split(','))]


def clock(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


for T, N, Cin, Cout, H, W, k, s in SHAPES:
    blk = convert_to_spiking(BaseConv(Cin, Cout, k, s), surrogate.ATan(2.0)).to(dev).eval()
    blk.planes_out = True
    x = (torch.rand(T, N, Cin, H, W, device=dev) < 0.2).float()
    xg = ops.ghost(x.shape, dev, ops.to_planes(x))
    ops._PLANES_SCOPE = True
    res = {}
    with torch.no_grad(), ops.no_state_writeback():
        for fused in (True, False):
            ops.FUSED_EVAL = 'all' if fused else False

            def f():
                blk(xg)
                functional.reset_net(blk)
            g = torch.cuda.CUDAGraph()
            f(); f()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                f()
            res[fused] = clock(g.replay)
    print(f'T{T} N{N} {Cin}->{Cout} {H}x{W} k{k} s{s}: fused {res[True]:.1f} us, conv + bn_lif {res[False]:.1f} us')
