"""The BASELINE.json configurations as runnable workloads: experiment options, input pipeline on the device, batch per GPU.

Used by bench.py (``--config``), scripts/ and the tests, so that "config 3" means the same thing everywhere:

  2  SYOLOX-S Gen1 304x240 (canvas 256x320), T=3, Tm=4, use_spike=True, batch 64/GPU            -- the headline metric
  3  SYOLOX-M Gen1 304x240 (canvas 256x320), T=5 with RPD (write_zero), full_spike_v2, batch 32/GPU (2 GPUs in BASELINE)
  4  SYOLOX-M 1 Mpx 640x360: RVT stacked histogram u8 [B,Tm,2*10,360,640] -> event_sum -> canvas 384x640, T=3, 3 classes (8 GPUs)
  5  SYOLOX-M N-Caltech101 180x240 (canvas 192x256), 100 classes, T=7, Tm=8, Ts=7 long-sequence adaptive sampling (4 GPUs)
  2b config 2's model at the README recipe's canvas: the published commands (readme.md:124-146) never set input_size, so they train at the
     default (640, 640) (yolox/exp/event_yolox_base.py:66,137) behind the dataset's letterbox resize (gen1.py:433-455): events -> counts
     240x304 -> eas_counts_letterbox (cv2 INTER_LINEAR to 640x505, pasted top-left) -> 640x640.  A side figure of bench.py, never `value`.

Reference: exps/default/e_yolox_{s,m}.py, yolox/exp/event_yolox_base.py (options), yolox/data/datasets/gen1.py:313-360 (micro_sum),
rvt_gen4.py:109-125 (stacked histogram, event_sum), ncaltech.py:26 (sensor size).  All inputs are synthetic (no datasets here)."""
import torch

from . import data, ops

COMMON = ['embedding', 'arsnn', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum', 'embedding_depth', '2', 'embedding_ksize', '5',
          'write_zero', 'True', 'spike_fn', 'atan']

WORKLOADS = {
    2: dict(name='SYOLOX-S Gen1 304x240 (canvas 256x320), T=3, Tm=4, arsnn sampler, use_spike=True', exp='e-yolox-s', baseline_gpus=1,
            opts=['T', '3', 'Tm', '4', 'num_classes', '2', 'use_spike', 'True'], sensor=(240, 304), canvas=(256, 320), Tm=4, batch=64,
            input='events', oracle=dict(use_spike='True')),
    '2b': dict(name='SYOLOX-S Gen1 304x240 letterboxed to the README canvas 640x640 (resize 640x505 + padding), T=3, Tm=4, arsnn sampler, use_spike=True',
               exp='e-yolox-s', baseline_gpus=1, opts=['T', '3', 'Tm', '4', 'num_classes', '2', 'use_spike', 'True'], sensor=(240, 304),
               canvas=(640, 640), Tm=4, batch=32, input='events_letterbox', oracle=dict(use_spike='True')),
    3: dict(name='SYOLOX-M Gen1 304x240 (canvas 256x320), T=5 with RPD, Tm=4, full_spike_v2', exp='e-yolox-m', baseline_gpus=2,
            opts=['T', '5', 'Tm', '4', 'num_classes', '2', 'use_spike', 'full_spike_v2'], sensor=(240, 304), canvas=(256, 320), Tm=4, batch=32,
            input='events', oracle=dict(depth=0.67, width=0.75, use_spike='full_spike_v2', T=5, Tm=4, num_classes=2)),
    4: dict(name='SYOLOX-M 1Mpx 640x360 stacked histogram nbins=10 (u8 [B,Tm,20,360,640] -> event_sum -> canvas 384x640), T=3, 3 classes, full_spike_v2',
            exp='e-yolox-m', baseline_gpus=8, opts=['T', '3', 'Tm', '4', 'num_classes', '3', 'use_spike', 'full_spike_v2'], sensor=(360, 640),
            canvas=(384, 640), Tm=4, batch=32, input='stacked_hist', oracle=dict(depth=0.67, width=0.75, use_spike='full_spike_v2', T=3, Tm=4, num_classes=3)),
    5: dict(name='SYOLOX-M N-Caltech101 180x240 (canvas 192x256), 100 classes, T=7, Tm=8, Ts=7 long-sequence adaptive sampling, full_spike_v2',
            exp='e-yolox-m', baseline_gpus=4, opts=['T', '7', 'Tm', '8', 'Ts', '7', 'num_classes', '100', 'use_spike', 'full_spike_v2'],
            sensor=(180, 240), canvas=(192, 256), Tm=8, batch=32, input='events',
            oracle=dict(depth=0.67, width=0.75, use_spike='full_spike_v2', T=7, Tm=8, Ts=7, num_classes=100)),
}


def get(config):
    key = config if config in WORKLOADS else (int(config) if str(config).isdigit() else config)
    if key not in WORKLOADS:
        raise KeyError(f'config {config}: BASELINE.json configs 2..5 (and 2b, config 2 at the 640x640 canvas) are runnable workloads '
                       '(config 1 is the CPU plumbing case)')
    return dict(WORKLOADS[key], config=key)


def exp_opts(w):
    return COMMON + list(w['opts']) + ['input_size', str(tuple(w['canvas'])), 'test_size', str(tuple(w['canvas']))]


def build_exp(w):
    from yolox.exp import get_exp
    exp = get_exp(None, w['exp'])
    exp.merge(exp_opts(w))
    return exp


def device_inputs(w, batch, n_events, device, seed=0):
    """Synthetic raw input of one batch, resident on the device, and ``inputs_fn() -> (frames [B,1,Tm,2,Hc,Wc] fp32, targets)``, the
    per-step device pipeline from that raw input (the histogram / the stacked-histogram reduction are part of the step)."""
    canvas, Tm = tuple(w['canvas']), w['Tm']
    targets = data.synth_targets(batch, canvas, device)
    if w['input'] == 'events':
        ev = data.events_to_device(data.synth_event_batch(batch, n_events, *w['sensor'], seed=seed), device)
        return ev, (lambda: (data.events_to_frames(ev, Tm, tuple(w['sensor']), canvas), targets))
    if w['input'] == 'events_letterbox':
        # the deterministic branch of GEN1Dataset.get_random_data (gen1.py:438-455): one (nw, nh, dx, dy, flip) row per sample, on the device
        # once -- the per-step pipeline is histogram + resize / paste, no host work
        ev = data.events_to_device(data.synth_event_batch(batch, n_events, *w['sensor'], seed=seed), device)
        par = torch.tensor([data.letterbox_params(*w['sensor'], *canvas)] * batch, dtype=torch.int32).to(device)
        H, W = w['sensor']

        def frames():
            counts = ops.event_histogram(ev['t'], ev['x'], ev['y'], ev['p'], ev['offsets'], Tm, H, W)
            return ops.counts_letterbox(counts, par, canvas[0], canvas[1]).unsqueeze(1)
        return ev, (lambda: (frames(), targets))
    if w['input'] == 'stacked_hist':
        g = torch.Generator().manual_seed(1 + seed)
        H, W = w['sensor']
        hist = torch.poisson(torch.full((batch, Tm, 20, H, W), 0.03), generator=g).clamp_(max=255).to(torch.uint8).to(device)
        return hist, (lambda: (ops.stacked_hist_event_sum(hist, *canvas), targets))
    raise KeyError(w['input'])


def build_trainer(w, batch, device, events=200_000, world=1, rank=0, force_exchange=False, out_dir='/tmp/eas_bench_out', ema=True):
    """The bench's step object for workload ``w``: the reference-shaped ``yolox.core.Trainer`` (model on the device, optimizer, gradient
    exchange when ``world`` > 1) and its ``TrainStep`` fed by the workload's device pipeline -- what ``bench.py`` times and what
    tests/test_gpu_bench_shapes.py traces.  Returns (trainer, model, step).

    ema: the weight average the reference keeps by default (``exp.ema = True``, event_yolox_base.py:116; ``ModelEMA.update`` after every
    ``optimizer.step()``, trainer.py:120-121) is part of the step -- made inside the optimizer's one launch (FusedAdam.attach_ema)."""
    import os
    import types
    exp = build_exp(w)
    exp.ema = bool(ema)                      # the measured step: forward + backward + exchange + Adam (+ weight average) + reset_net
    exp.output_dir = os.environ.get('EAS_BENCH_OUT', out_dir)
    torch.manual_seed(80)
    trainer = exp.get_trainer(types.SimpleNamespace(batch_size=batch * world, fp16=False, experiment_name=f'bench_config{w["config"]}',
                                                     ckpt=None, resume=False))
    model = trainer.setup(force_exchange=force_exchange)
    model.head.use_l1 = True                 # no_aug from epoch 0 (trainer.py:157, 231-238)
    model.head.fused_assign = os.environ.get('EAS_FUSED_ASSIGN', '1') == '1'    # development switch: 0 = tensor-op SimOTA
    model.head.fused_loss = os.environ.get('EAS_FUSED_LOSS', '1') == '1'        # development switch: 0 = tensor-op loss terms
    raw, inputs_fn = device_inputs(w, batch, events, device, seed=rank)
    trainer.ema_model = trainer.make_ema(0.9998, 0) if ema else None       # (what Trainer.before_train does; join_ranks re-copies it after the broadcast)
    step = trainer.step_fn(inputs_fn, ema=trainer.ema_model)
    step.raw_inputs = raw
    return trainer, model, step


def algorithmic_input_bytes(w, batch, n_events):
    """bytes of raw input one step reads: 9 B per event, or the u8 stacked histogram"""
    if w['input'] in ('events', 'events_letterbox'):
        return 9 * batch * n_events
    H, W = w['sensor']
    return batch * w['Tm'] * 20 * H * W
