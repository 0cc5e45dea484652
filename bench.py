#!/usr/bin/env python3
"""bench.py -- event-frames/sec of the SYOLOX-S (T=3) training step on MI355X.

One "step" = one pass of the hot path over one batch of synthetic raw events already resident in HBM:
  K1 event histogram (4 micro-slices) -> canvas -> adaptive sampler (arsnn) -> spiking backbone (T=3) -> PAFPN ->
  head -> SimOTA loss -> backward -> (N>1: one packed gradient all-reduce over RCCL, eas_snn_amd/parallel.py) -> Adam step ->
  reset_net.  The warm-up also times eager launches against HIP-graph replay of the same step and the timed region uses the faster.
Workload = BASELINE.json configs[1]: SYOLOX-S, Gen1 304x240 sensor (256x320 canvas), T=3, Tm=4, batch 64 per GPU.
Multi-GPU: data parallel, weak scaling (64 samples per rank), one process per GPU over RCCL.  ``python bench.py --gpus N`` with
no WORLD_SIZE in the environment starts the N ranks itself (children are started before the parent touches the GPU, like the
reference's yolox/core/launch.py:59-98 spawns its workers; the parent only waits and passes rank 0's line through); under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` the ranks are already there.  A rank count different
from --gpus is an error.  At N > 1 every rank replays TWO HIP graphs per step -- (histogram, forward, backward, gradient
packing) and (Adam, reset) -- with the one flat RCCL all-reduce launched eagerly between them: three host calls per step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      achieved algorithmic GB/s of the dominant hand-written HIP kernel family, timed with HIP events
                on the launch stream inside the timed region;
  cpu_baseline  the CPU oracle (torch fp32 restatement of the reference) timed on this host, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
F32_MFMA_PEAK_TF = 157.3    # dense f32-input MFMA peak (MI355X_MICROARCH.md, Matrix cores)
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak
# HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same
# command, scripts/gpu_profile.sh; gfx950 corrections applied by scripts/pmc_summary.py).  C-ABI entry -> device kernels.
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.json')
PMC_KERNELS = {'eas_bn_lif_bwd': ['bn_lif_bwd_kernel'], 'eas_bn_lif_fwd': ['bn_lif_fwd_kernel'], 'eas_bn_stats': ['bn_stats_partial'],
               'eas_bn_silu_bwd': ['bn_silu_bwd_kernel'], 'eas_bn_silu_fwd': ['bn_silu_fwd_kernel'],
               'eas_event_histogram': ['event_hist_kernel', 'event_hist_banded_kernel'], 'eas_arsnn_step_fwd': ['arsnn_step_fwd_kernel'],
               'eas_smallconv_fwd': ['smallconv_kernel'], 'eas_smallconv_bwd_weight': ['smallconv_wgrad_kernel', 'smallconv_wgrad_mfma_kernel'],
               'eas_conv_fwd': ['conv_fwd_mfma_kernel', 'conv1x1_mfma_kernel'], 'eas_conv_wgrad': ['conv_wgrad_mfma_kernel', 'conv1x1_wgrad_kernel']}


def pmc_traffic(entry, launches_per_call):
    """Average HBM bytes one call of ``entry`` moves, from the committed PMC summary: launch-weighted mean over the device
    kernels that implement it, times the launches one call makes."""
    try:
        with open(PMC_FILE) as fh:
            pmc = json.load(fh)
        ks = [pmc[k] for k in PMC_KERNELS[entry] if k in pmc]
        n = sum(k['launches'] for k in ks)
        return round(sum(k['hbm_bytes_per_launch'] * k['launches'] for k in ks) / n * launches_per_call)
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


SENSOR = (240, 304)
CANVAS = (256, 320)
OPTS = ['T', '3', 'Tm', '4', 'embedding', 'arsnn', 'num_classes', '2', 'spike_attach', 'True', 'thresh', '1', 'readout', 'sum',
        'embedding_depth', '2', 'embedding_ksize', '5', 'write_zero', 'True', 'use_spike', 'True', 'spike_fn', 'atan',
        'input_size', '(256,320)', 'test_size', '(256,320)']


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='samples per GPU')
    ap.add_argument('--events', type=int, default=200_000, help='events per sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--selftest-cpu', action='store_true',
                    help='no GPU: run the launcher, rendezvous (gloo), barrier / max-over-ranks timing and the flat gradient '
                         'exchange on a stand-in CPU module (tests/test_cpu_host.py)')
    return ap.parse_args()


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for ln in fh:
                if ln.lower().startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(cpu_batch, n_events):
    """The oracle (a port: torch-CPU restatement validated against the reference) on a bounded sample of the same
    workload: same model/config, batch ``cpu_batch``, numpy event binning + fwd + bwd + Adam + reset per iteration."""
    from oracle import events_ref, model_ref, sj_ref
    # intra-op threads: all host cores up to 32 (beyond that ATen's CPU convs at these sizes get slower, not faster)
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    torch.manual_seed(80)
    model = model_ref.build_model(use_spike='True')
    model.head.use_l1 = True
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    streams = [events_ref.synth_events(n_events, *SENSOR, seed=100 + b) for b in range(cpu_batch)]
    tg = torch.zeros(cpu_batch, 50, 5)
    tg[:, 0] = torch.tensor([0, 0.3 * 320, 0.4 * 256, 0.25 * 320, 0.3 * 256])
    tg[:, 1] = torch.tensor([1, 0.7 * 320, 0.6 * 256, 0.2 * 320, 0.35 * 256])

    def step():
        frames = np.zeros((cpu_batch, 1, 4, 2) + CANVAS, np.float32)
        for b, (t, x, y, p) in enumerate(streams):
            frames[b, 0, :, :, :SENSOR[0], :SENSOR[1]] = events_ref.micro_sum(t, x, y, p, 4, *SENSOR)
        out = model(torch.from_numpy(frames), tg)
        opt.zero_grad()
        out['total_loss'].backward()
        opt.step()
        sj_ref.reset_net(model)

    t0 = time.time()
    step()                                   # warm-up (allocator, thread pools)
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while True:
        step()
        n += 1
        el = time.time() - t0
        if el > 12.0 or n >= 5 or warm > 20.0:   # bounded sample: ~10-30 s of CPU work
            break
    # SURVEY 8(d) side figures: K1's numpy restatement single-threaded (how a DataLoader worker runs it) and the eval
    # forward of one sample (BASELINE configs[0]), both bounded to a few seconds
    t1 = time.time()
    k1_n = 0
    while time.time() - t1 < 1.5:
        events_ref.micro_sum(*streams[k1_n % cpu_batch], 4, *SENSOR)
        k1_n += 1
    k1_rate = k1_n * n_events / (time.time() - t1)
    model.eval()
    one = torch.from_numpy(np.zeros((1, 1, 4, 2) + CANVAS, np.float32))
    one[0, 0, :, :, :SENSOR[0], :SENSOR[1]] = torch.from_numpy(events_ref.micro_sum(*streams[0], 4, *SENSOR).astype(np.float32))
    with torch.no_grad():
        model(one)
        sj_ref.reset_net(model)
        t2, ev_n = time.time(), 0
        while time.time() - t2 < 2.0:
            model(one)
            sj_ref.reset_net(model)
            ev_n += 1
    ev_rate = ev_n / (time.time() - t2)
    return {'value': round(cpu_batch * n / el, 3), 'unit': 'event-frames/s', 'cores': threads, 'kind': 'port', 'cpu_model': _cpu_model(),
            'host_logical_cpus': os.cpu_count(),
            'sample': f'oracle (torch-CPU fp32) SYOLOX-S T=3 256x320 fwd+bwd+Adam, batch {cpu_batch}, {n} iterations, '
                      f'{n_events} events/sample binned with numpy',
            'k1_numpy_events_per_s_1thread': round(k1_rate), 'eval_forward_1_sample_frames_per_s': round(ev_rate, 2)}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def _visible_gpus():
    """GPUs this process may use, counted WITHOUT the HIP runtime (torch.cuda.device_count() may run hipInit in the parent, which must
    not hold the GPU when it starts its children): the visibility variables if set, else the KFD topology (nodes with SIMDs are
    GPUs).  None when neither is readable -- then the children report a missing device themselves."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(',') if t.strip() != ''])
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as fh:
                for ln in fh:
                    if ln.startswith('simd_count') and int(ln.split()[1]) > 0:
                        n += 1
        return n
    except (OSError, ValueError, IndexError):
        return None


def launch_ranks(args):
    """Start one child process per GPU and wait for them.  Runs BEFORE anything in this process touches the GPU (devices are counted
    from the environment / sysfs, not through the HIP runtime); the children are fresh interpreters, never an exec of a process that
    holds the GPU.  Rank 0's JSON line reaches stdout through the inherited descriptor.  Returns the exit code (non-zero when any
    rank failed)."""
    import subprocess
    n = args.gpus
    if not args.selftest_cpu:
        have = _visible_gpus()
        if have is not None and have < n:
            print(f'bench.py: --gpus {n} but only {have} GPU(s) visible', file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r), EAS_BENCH_SPAWNED='1')))
    rc = 0
    try:
        while procs:
            for pr in list(procs):
                code = pr.poll()
                if code is None:
                    continue
                procs.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in procs:              # one rank failed: the others would wait in a collective forever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            pr.kill()
    return rc


def selftest_cpu(args, world, rank):
    """The N-rank protocol of this file without a GPU: gloo rendezvous from the launcher's environment, flat gradient exchange on a
    stand-in module, barrier-bracketed timed region, max over ranks, one JSON line from rank 0."""
    from eas_snn_amd.parallel import FlatGradAllReduce
    if os.environ.get('EAS_BENCH_SELFTEST_FAIL_RANK') == str(rank):     # test hook: this rank dies before the rendezvous
        sys.exit(7)
    if world > 1:
        dist.init_process_group('gloo')
    torch.manual_seed(rank)
    net = torch.nn.Linear(16, 4)
    sync = FlatGradAllReduce(net) if world > 1 else None
    x = torch.full((8, 16), float(rank + 1))

    def step():
        net.zero_grad(set_to_none=True)
        net(x).sum().backward()
        if sync is not None:
            sync.sync()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    g = net.weight.grad.clone()
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        lo, hi = g.clone(), g.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), 'ranks hold different averaged gradients'
    if rank == 0:
        # d(sum)/dW = 8 * x per output row; averaged over ranks r = 1..world
        want = 8.0 * sum(range(1, world + 1)) / world
        assert torch.allclose(g, torch.full_like(g, want)), (g[0, 0].item(), want)
        print(json.dumps({'selftest': True, 'n_gpus': world, 'rccl_ranks': dist.get_world_size() if world > 1 else 1, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': round(float(el) / args.steps * 1e3, 4), 'backend': 'gloo',
                          'gradient_exchange': 'flat' if world > 1 else None,
                          'spawned_by_bench': os.environ.get('EAS_BENCH_SPAWNED') == '1'}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print(f'bench.py: launched with {world} rank(s) for --gpus {args.gpus}', file=sys.stderr)
        sys.exit(3)
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_ddp = os.environ.get('EAS_BENCH_FORCE_DDP') == '1'     # development: exercise the DDP/RCCL path with one rank
    if world > 1 or force_ddp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if force_ddp and world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)      # RCCL over xGMI

    # everything runs on one non-default stream: autograd's AccumulateGrad nodes remember the stream they were created on,
    # and a later graph capture breaks if that was the default stream
    torch.cuda.set_stream(torch.cuda.Stream())

    import eas_snn_amd
    from eas_snn_amd import data, ops
    from spikingjelly.activation_based import functional
    from yolox.exp import get_exp
    eas_snn_amd.hip_library()
    ops.set_state_writeback(False)           # every step ends with reset_net (as yolox/core/trainer.py:115-117)

    exp = get_exp(None, 'e-yolox-s')
    exp.merge(OPTS)
    torch.manual_seed(80)
    model = exp.get_model().to(dev)
    model.head.use_l1 = True                 # no_aug from epoch 0 (trainer.py:157, 231-238)
    model.head.fused_assign = os.environ.get('EAS_FUSED_ASSIGN', '1') == '1'    # development switch: 0 = tensor-op SimOTA
    model.head.fused_loss = os.environ.get('EAS_FUSED_LOSS', '1') == '1'        # development switch: 0 = tensor-op loss terms
    opt = exp.get_optimizer(args.batch * world)
    net = model
    flat_sync = None
    # gradient exchange at N > 1: one flat RCCL all-reduce after backward (eas_snn_amd/parallel.py; ~7 ms less host work per step
    # than DistributedDataParallel, which hides a 0.6 ms all-reduce but pushes the host to 0.83 of the step) -- EAS_BENCH_DP=ddp
    # selects DistributedDataParallel (bucketed, overlapped with backward) instead
    dp_mode = os.environ.get('EAS_BENCH_DP', 'flat')
    if world > 1 or force_ddp:
        if dp_mode == 'ddp':
            # gradients live inside the all-reduce buckets (no per-parameter copy kernels)
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                            gradient_as_bucket_view=True)
        else:
            from eas_snn_amd.parallel import FlatGradAllReduce
            flat_sync = FlatGradAllReduce(model)

    ev = data.events_to_device(data.synth_event_batch(args.batch, args.events, *SENSOR, seed=rank), dev)
    targets = data.synth_targets(args.batch, CANVAS, dev)

    # the slab reductions of all weight gradients in one launch at the end of the backward pass; DistributedDataParallel copies
    # gradients into its buckets inside the pass, so not there
    defer = ops.deferred_wgrad_reductions(not (net is not model) and os.environ.get('EAS_DEFER_WGRAD_REDUCE', '1') == '1')

    def fwd_bwd():
        frames = data.events_to_frames(ev, exp.Tm, SENSOR, CANVAS)
        out = net(frames, targets)
        opt.zero_grad(set_to_none=True)
        with defer:
            out['total_loss'].backward()
        if flat_sync is not None:
            flat_sync.pack()
        return out['total_loss']

    def update():
        if flat_sync is not None:
            flat_sync.attach()
        opt.step()
        functional.reset_net(model)

    def step():
        loss_ = fwd_bwd()
        if flat_sync is not None:
            flat_sync.reduce()               # the one collective of the step: RCCL all-reduce of the flat gradient buffer
        update()
        return loss_

    # Launch mode.  The step has no host synchronisation, so it can be captured into HIP graphs and replayed.  One GPU: ONE
    # graph for the whole step; the warm-up times a few eager steps and a few replays and the timed region uses whichever was
    # faster (eager wins by ~3 % while the host stays ahead of the GPU, loses by 2x on a busy host).  N > 1 (flat exchange):
    # TWO graphs -- forward + backward + gradient packing | Adam + reset -- with the RCCL all-reduce launched eagerly between
    # them, so a rank costs the host three calls per step instead of ~1100 kernel launches (eight ranks share one host).
    # DistributedDataParallel (EAS_BENCH_DP=ddp) stays eager: its reducer hooks cannot be captured.
    # EAS_BENCH_GRAPH=0/1 forces eager/graph.
    mode = os.environ.get('EAS_BENCH_GRAPH', 'auto')
    for _ in range(max(args.warmup - 2, 1)):
        loss = step()
    torch.cuda.synchronize()
    t_a = time.perf_counter()
    step()
    loss = step()
    t_enq = time.perf_counter() - t_a
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_a
    graph = None
    probe = {}
    launch = 'eager launches'
    multi = world > 1 or force_ddp
    run = step
    if mode in ('1', 'auto') and not (multi and dp_mode == 'ddp'):
        for gr in opt.param_groups:
            gr['capturable'] = True
        for st_ in opt.state.values():         # Adam's step counters live on the host in eager mode; a captured step needs them on the device
            if torch.is_tensor(st_.get('step')):
                st_['step'] = st_['step'].to(dev)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        if not multi:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss = step()
            graph.replay()                       # warm-up replay
            run = graph.replay
            launch = 'hip-graph replay of the whole step'
            if mode == 'auto':
                def clock(fn, n=4):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t) / n * 1e3
                probe = {'eager_ms': round(clock(step), 3), 'graph_ms': round(clock(graph.replay), 3)}
                if probe['eager_ms'] < probe['graph_ms']:
                    run, launch = step, 'eager launches'         # eager is faster on this host right now
        else:
            pool = torch.cuda.graph_pool_handle()
            g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_a, pool=pool):
                loss = fwd_bwd()
            flat_sync.reduce()
            with torch.cuda.graph(g_b, pool=pool):
                update()
            graph = (g_a, g_b)

            def run():
                g_a.replay()
                flat_sync.reduce()
                g_b.replay()
            run()                                # warm-up replay
            launch = 'two hip-graph replays per step (fwd+bwd+pack | adam+reset) with the eager RCCL all-reduce between them'
    timer = ops.KernelTimer() if rank == 0 else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = run()
        loss = r if r is not None else loss
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # Per-kernel durations behind `roofline`: HIP events around every C-ABI call, on the stream the kernels are launched on, in
    # eager steps of this same process right after the timed region (same kernels, same data).  Not inside it: events cannot be
    # recorded into a captured graph, and ~1400 event records per eager step would slow the timed region itself by ~10 %.
    # With more than one rank every rank runs these steps (the DDP all-reduce needs all of them); rank 0 records.
    timed_steps = 3
    ops.set_timer(timer)
    for _ in range(timed_steps):
        step()
    torch.cuda.synchronize()
    ops.set_timer(None)
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el)
    assert torch.isfinite(loss), 'training step produced a non-finite loss'

    if rank == 0:
        frames_total = args.batch * world * args.steps
        summ = timer.summary()
        fam = {k: dict(calls=v['calls'], ms_per_step=round(v['ms'] / timed_steps, 4),
                       GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None,
                       **({'TFLOPs': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1)} if v['flops'] > 0 else {})) for k, v in summ.items()}
        dom = max(summ, key=lambda k: summ[k]['ms'])
        d = summ[dom]
        sec = d['ms'] * 1e-3
        common = {'kernel': dom, 'avg_launch_ms': round(d['ms'] / d['calls'], 4), 'traffic': pmc_traffic(dom, 2 if dom in ('eas_bn_lif_bwd', 'eas_bn_silu_bwd') else 1),
                  'traffic_source': 'committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes over this command (profiles/pmc_traffic_latest.json, '
                                    'scripts/gpu_profile.sh), not re-measured in this run',
                  'algorithmic_bytes_per_call': round(d['bytes'] / d['calls']), 'hip_kernel_ms_per_step': fam,
                  'hip_share_of_step': round(sum(v['ms'] for v in summ.values()) / timed_steps / (elapsed / args.steps * 1e3), 4),
                  'kernel_timing': 'HIP events on the launch stream, 3 eager steps of this process right after the timed region'}
        if d['flops'] > 0:
            # dense convolutions: bounded by the matrix cores.  achieved = algorithmic flops (2 x MAC of the fp32 convolution) per
            # second against the dense f32 MFMA peak (the arithmetic the path reproduces); every fp32 product is formed from
            # 3 (spike inputs) or 6 (general inputs) exact bf16 term products on v_mfma_f32_32x32x16_bf16, so the bf16 flops
            # actually issued and their share of the dense bf16 peak are reported next to it.
            achieved = d['flops'] / sec / 1e12
            # ceiling of the exact-bf16-term scheme itself: the bf16 pipe divided by the term products per fp32 product
            # (3 for spike inputs -> 833 TF, 6 for real-valued inputs -> 417 TF), weighted by this step's mix of the two
            scheme_ceiling = BF16_MFMA_PEAK_TF * d['flops'] / d['issue_flops']
            roofline = {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': F32_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                        'frac': round(achieved / F32_MFMA_PEAK_TF, 4),
                        'scheme_ceiling_tflops': round(scheme_ceiling, 1), 'frac_of_scheme_ceiling': round(achieved / scheme_ceiling, 4),
                        'mfma_bf16_issued_tflops': round(d['issue_flops'] / sec / 1e12, 1), 'mfma_bf16_peak_tflops': BF16_MFMA_PEAK_TF,
                        'mfma_bf16_util': round(d['issue_flops'] / sec / 1e12 / BF16_MFMA_PEAK_TF, 4),
                        'algorithmic_GBps': round(d['bytes'] / sec / 1e9, 1)}
        else:
            achieved = d['bytes'] / sec / 1e9
            roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(achieved / HBM_PEAK_GBS, 4)}
        roofline.update(common)
        # the dominant HBM-bound family as well (the elementwise/BN/LIF kernels of the step)
        hbm_fams = {k: v for k, v in summ.items() if v['flops'] == 0}
        if hbm_fams:
            hk = max(hbm_fams, key=lambda k: hbm_fams[k]['ms'])
            hv = hbm_fams[hk]
            roofline['hbm_dominant'] = {'kernel': hk, 'achieved': round(hv['bytes'] / (hv['ms'] * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                                        'unit': 'GB/s', 'frac': round(hv['bytes'] / (hv['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        'traffic': pmc_traffic(hk, 2 if hk in ('eas_bn_lif_bwd', 'eas_bn_silu_bwd') else 1),
                                        'algorithmic_bytes_per_call': round(hv['bytes'] / hv['calls'])}
        line = {'metric': 'event-frames/sec (T=3) SYOLOX-S Gen1 304x240', 'value': round(frames_total / elapsed, 2),
                'unit': 'event-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': 'SYOLOX-S Gen1 304x240 (canvas 256x320), T=3, Tm=4, arsnn sampler, batch 64/GPU, '
                                       'raw events -> histogram -> fwd + bwd + Adam + reset_net',
                           'global_batch': args.batch * world, 'events_per_sample': args.events, 'parallelism': f'dp{world}', 'gradient_exchange': (dp_mode if (world > 1 or force_ddp) else None),
                           'launch': launch, 'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                           'launch_probe_ms_per_step': probe, 'host_enqueue_share_of_step': round(t_enq / t_tot, 3)},
                'roofline': roofline}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.cpu_batch, args.events)
        print(json.dumps(line), flush=True)
    if world > 1 or force_ddp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
