#!/usr/bin/env python3
"""One training step (forward + backward + Adam + reset) of BASELINE.json configs[2..4] at their full per-GPU shapes on one
GPU (development tool: checks that every kernel's geometry limits hold at those sizes and prints the step time and how much
of the GPU time is spent in library kernels instead of libeas_hip.so).  Usage: python scripts/config_smoke.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import eas_snn_amd
from eas_snn_amd import data, ops
from spikingjelly.activation_based import functional
from yolox.exp import get_exp

CONFIGS = {
    # BASELINE.json configs[2]: SYOLOX-M Gen1 304x240, T=5 with RPD (write_zero), full_spike_v2
    'cfg3_m_gen1_t5': ('e-yolox-m', ['T', '5', 'Tm', '4', 'use_spike', 'full_spike_v2', 'num_classes', '2'], (256, 320), 4),
    # configs[3]: SYOLOX-M 1Mpx 640x360 stacked histogram summed to 2 channels, T=3, 3 classes, canvas 384x640
    'cfg4_m_1mpx_t3': ('e-yolox-m', ['T', '3', 'Tm', '4', 'use_spike', 'full_spike_v2', 'num_classes', '3'], (384, 640), 4),
    # configs[4]: SYOLOX-M N-Caltech101 180x240 -> canvas 192x256, 100 classes, T=7, Tm=8, Ts=7
    'cfg5_m_ncaltech_t7': ('e-yolox-m', ['T', '7', 'Tm', '8', 'Ts', '7', 'use_spike', 'full_spike_v2', 'num_classes', '100'], (192, 256), 8),
}
COMMON = ['embedding', 'arsnn', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum', 'embedding_depth', '2', 'embedding_ksize', '5',
          'write_zero', 'True', 'spike_fn', 'atan']


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device('cuda:0')
    eas_snn_amd.hip_library()
    ops.set_state_writeback(False)           # whole process is a train loop that resets after every step
    for name, (exp_name, opts, canvas, Tm) in CONFIGS.items():
        exp = get_exp(None, exp_name)
        exp.merge(COMMON + opts + ['input_size', str(canvas), 'test_size', str(canvas)])
        torch.manual_seed(80)
        model = exp.get_model().to(dev)
        model.head.use_l1 = True
        model.head.fused_assign = os.environ.get('EAS_FUSED_ASSIGN', '1') == '1'
        if os.environ.get('EAS_SMOKE_ONLY') and os.environ['EAS_SMOKE_ONLY'] not in name:
            continue
        opt = exp.get_optimizer(batch)
        g = torch.Generator().manual_seed(1)
        hist = None
        if 'cfg4' in name:
            # configs[3]'s real input: the RVT stacked histogram [B, Tm, 2*10, 360, 640] u8, summed over its 10 time bins and padded
            # to the canvas on the device inside the step (eas_stacked_hist_event_sum; rvt_gen4.py:118-125)
            hist = torch.poisson(torch.full((batch, Tm, 20, 360, 640), 0.03), generator=g).clamp_(max=255).to(torch.uint8).to(dev)
            frames = None
        else:
            frames = torch.poisson(torch.full((batch, 1, Tm, 2) + canvas, 0.3), generator=g).to(dev)
        targets = data.synth_targets(batch, canvas, dev)

        def step():
            x_in = frames if hist is None else ops.stacked_hist_event_sum(hist, *canvas)
            out = model(x_in, targets)
            opt.zero_grad(set_to_none=True)
            out['total_loss'].backward()
            opt.step()
            functional.reset_net(model)
            return out['total_loss']

        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        own = lib = 0.0
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                d = e.device_time if hasattr(e, 'device_time') else e.cuda_time
                if '(anonymous namespace)' in e.name and not e.name.startswith('void at::'):
                    own += d
                elif any(k in e.name for k in ('igemm', 'Sp3AsmConv', 'Cijk', 'transpose', 'ck::', 'naive_conv', 'Im2d2Col', 'miopen')):
                    lib += d
        tot = sum((e.device_time if hasattr(e, 'device_time') else e.cuda_time) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
        print(f'{name:22s} batch {batch}: {ms:8.1f} ms/step = {batch / ms * 1e3:7.1f} event-frames/s, loss {float(loss):.4f} finite {bool(torch.isfinite(loss))}; '
              f'GPU time: own kernels {own / tot:.0%}, MIOpen/rocBLAS {lib / tot:.1%}, max memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
        del model, opt, frames, hist
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
