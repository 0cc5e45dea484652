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

# kernel arguments in device memory: the HIP runtime's default on this image; an inherited HIP_FORCE_DEV_KERNARG=0 costs the 452 launches of a
# step +0.55 ms (HISTORY.md D, round 6).  Has to be in the environment before the runtime initialises.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
F32_MFMA_PEAK_TF = 157.3    # dense f32-input MFMA peak (MI355X_MICROARCH.md, Matrix cores)
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak
# HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same
# command, scripts/gpu_profile.sh; gfx950 corrections applied by scripts/pmc_summary.py).  C-ABI entry -> device kernels.
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic_latest.json')      # config 2; other configurations: pmc_traffic_latest_config<N>.json
PMC_KERNELS = {'eas_bn_lif_bwd': ['bn_lif_bwd_kernel', 'bn_lif_bwd_small_kernel'], 'eas_bn_lif_fwd': ['bn_lif_fwd_kernel', 'bn_lif_fwd_sp_kernel'], 'eas_bn_stats': ['bn_stats_partial'],
               'eas_bn_silu_bwd': ['bn_silu_bwd_kernel', 'bn_silu_bwd_small_kernel', 'bn_silu_bwd_group_kernel'],
               'eas_bn_silu_fwd': ['bn_silu_fwd_kernel', 'bn_silu_fwd_group_kernel'],
               'eas_event_histogram': ['event_hist_kernel', 'event_hist_banded_kernel', 'event_hist_banded16_kernel'],
               'eas_arsnn_step_fwd': ['arsnn_step_fwd_kernel', 'arsnn_fused_step_fwd_kernel'],
               'eas_smallconv_fwd': ['smallconv_kernel'], 'eas_smallconv_bwd_weight': ['smallconv_wgrad_kernel', 'smallconv_wgrad_mfma_kernel'],
               'eas_conv_fwd': ['conv_fwd_mfma_kernel', 'conv1x1_mfma_kernel', 'conv1x1_mfma_sharedA_kernel', 'conv_dgrad_s2_kernel',
                                'conv3x3_group_kernel', 'conv1x1_group_kernel', 'conv_fwd_mfma_kernel[planes]', 'conv1x1_mfma_kernel[planes]', 'conv1x1_mfma_sharedA_kernel[planes]'],
               'eas_conv_wgrad': ['conv_wgrad_mfma_kernel', 'conv1x1_wgrad_lds_kernel', 'conv_wgrad_group_kernel', 'conv1x1_wgrad_group_kernel',
                                  'conv_wgrad_mfma_kernel[planes]',
                                  'conv1x1_wgrad_lds_kernel[planes]']}


# the sources a family's kernels are built from: a committed PMC figure is only quoted while they are what it was measured on
_CONV_SRC = ['conv_mfma.hip', 'conv_mfma_body.h', 'conv1x1_mfma.hip', 'conv_group.hip', 'conv_s2d.hip', 'conv_lif_epi.h', 'conv_small_dgrad.hip', 'eas_common.h']
FAMILY_SOURCES = {'eas_conv_fwd': _CONV_SRC, 'eas_conv_wgrad': ['conv_wgrad_mfma.hip', 'conv1x1_mfma.hip', 'eas_common.h'],
                  'eas_bn_lif_bwd': ['bn_lif.hip', 'eas_common.h'], 'eas_bn_lif_fwd': ['bn_lif.hip', 'eas_common.h'], 'eas_bn_stats': ['bn_lif.hip', 'eas_common.h'],
                  'eas_bn_silu_bwd': ['bn_act.hip', 'eas_common.h'], 'eas_bn_silu_fwd': ['bn_act.hip', 'eas_common.h'],
                  'eas_event_histogram': ['events.hip', 'eas_common.h'], 'eas_arsnn_step_fwd': ['arsnn.hip', 'smallconv_core.h', 'eas_common.h'],
                  'eas_smallconv_fwd': ['smallconv.hip', 'smallconv_core.h', 'eas_common.h'],
                  'eas_smallconv_bwd_weight': ['smallconv.hip', 'smallconv_wgrad_mfma.hip', 'smallconv_core.h', 'eas_common.h']}


def csrc_hashes():
    import hashlib
    d = os.path.join(ROOT, 'eas_snn_amd', 'csrc')
    return {f: hashlib.sha256(open(os.path.join(d, f), 'rb').read()).hexdigest()[:16] for f in sorted(os.listdir(d)) if f.endswith(('.hip', '.h'))}


def pmc_traffic(entry, launches_per_call, config=2):
    """-> (bytes, reason).  Average HBM bytes one call of ``entry`` moves, from the committed PMC summary of this configuration:
    launch-weighted mean over the device kernels that implement it, times the launches one call makes.  The summary records the hashes of
    the kernel sources it was measured on (scripts/pmc_summary.py); when a source of this family has changed since, the figure is stale
    and is NOT quoted: (None, why)."""
    path = PMC_FILE if config == 2 else PMC_FILE.replace('.json', f'_config{config}.json')
    try:
        with open(path) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        return None, f'no PMC summary for this configuration ({os.path.basename(path)})'
    recorded = (pmc.get('_meta') or {}).get('csrc_sha16')
    if not recorded:
        return None, f'{os.path.basename(path)} does not record the kernel sources it was measured on'
    now = csrc_hashes()
    changed = [f for f in FAMILY_SOURCES.get(entry, sorted(now)) if recorded.get(f) != now.get(f)]
    if changed:
        return None, f'stale: {", ".join(changed)} changed after the PMC passes of {os.path.basename(path)}; re-run scripts/gpu_profile.sh'
    try:
        ks = [pmc[k] for k in PMC_KERNELS[entry] if k in pmc]
        n = sum(k['launches'] for k in ks)
        return round(sum(k['hbm_bytes_per_launch'] * k['launches'] for k in ks) / n * launches_per_call), None
    except (KeyError, ValueError, ZeroDivisionError):
        return None, f'{os.path.basename(path)} holds no kernel of {entry}'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3, 4, 5),
                    help='BASELINE.json configuration (eas_snn_amd/workloads.py): 2 = SYOLOX-S T=3 Gen1 (the headline metric, default); 3 = SYOLOX-M '
                         'T=5 RPD Gen1; 4 = SYOLOX-M 1Mpx stacked histogram T=3; 5 = SYOLOX-M N-Caltech101 T=7')
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU (default: the configuration\'s: 64 for config 2, 32 for 3-5)')
    ap.add_argument('--events', type=int, default=200_000, help='events per sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=None)
    ap.add_argument('--h2d', action='store_true',
                    help='stream a fresh pinned host batch of raw input to the device every step on a side stream (the reference\'s '
                         'DataPrefetcher shape, yolox/data/data_prefetcher.py:31-44): PCIe-inclusive rate, reported as such')
    ap.add_argument('--canvas640-child', action='store_true', help=argparse.SUPPRESS)     # internal: the 640x640 side figure's own process
    ap.add_argument('--selftest-cpu', action='store_true',
                    help='no GPU: run the launcher, rendezvous (gloo), barrier / max-over-ranks timing and the trainer\'s step object with the '
                         'bucketed gradient exchange on a stand-in CPU module (tests/test_cpu_host.py)')
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


def _usable_cpus():
    """CPUs this process can keep busy: the smaller of its affinity mask and its cgroup (v2) CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(w, cpu_batch, n_events):
    """The oracle (a port: torch-CPU restatement validated against the reference) on a bounded sample of the same workload: same
    model / configuration, batch ``cpu_batch``, numpy input reduction (event binning / stacked-histogram sum) + fwd + bwd + Adam +
    reset per iteration.  Threads = the CPUs this process may really use (affinity mask and cgroup quota): on the GPU box 256 logical CPUs
    are visible but the cgroup grants 16, and a 2..128 sweep peaks exactly there (13.9 frames/s at 16 threads, 8.5 at 32, 1.7 at 128:
    profiles/r03_cpu_baseline_threads.txt, scripts/cpu_baseline_sweep.py); EAS_CPU_THREADS overrides it."""
    from oracle import events_ref, model_ref, sj_ref
    threads = int(os.environ.get('EAS_CPU_THREADS', _usable_cpus()))
    torch.set_num_threads(threads)
    torch.manual_seed(80)
    model = model_ref.build_model(**w['oracle'])
    model.head.use_l1 = True
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sensor, canvas, Tm = tuple(w['sensor']), tuple(w['canvas']), w['Tm']
    if w['input'] == 'events':
        streams = [events_ref.synth_events(n_events, *sensor, seed=100 + b) for b in range(cpu_batch)]
    else:
        rng = np.random.default_rng(7)
        hists = [rng.poisson(0.03, (Tm, 20) + sensor).clip(0, 255).astype(np.uint8) for _ in range(cpu_batch)]
    tg = torch.zeros(cpu_batch, 50, 5)
    tg[:, 0] = torch.tensor([0, 0.3 * canvas[1], 0.4 * canvas[0], 0.25 * canvas[1], 0.3 * canvas[0]])
    tg[:, 1] = torch.tensor([1, 0.7 * canvas[1], 0.6 * canvas[0], 0.2 * canvas[1], 0.35 * canvas[0]])

    def make_frames(b):
        if w['input'] == 'events':
            return events_ref.micro_sum(*streams[b], Tm, *sensor)
        return events_ref.stacked_hist_event_sum(hists[b], Tm, *sensor)[0]

    def step():
        frames = np.zeros((cpu_batch, 1, Tm, 2) + canvas, np.float32)
        for b in range(cpu_batch):
            frames[b, 0, :, :, :sensor[0], :sensor[1]] = make_frames(b)
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
    res = {'value': round(cpu_batch * n / el, 3), 'unit': 'event-frames/s', 'cores': threads, 'kind': 'port', 'cpu_model': _cpu_model(),
           'host_logical_cpus': os.cpu_count(),
           'sample': f"oracle (torch-CPU fp32) {w['name']}: fwd+bwd+Adam+reset, batch {cpu_batch}, {n} iterations, input reduced with numpy"}
    if w['input'] == 'events':
        # SURVEY 8(d) side figures: K1's numpy restatement single-threaded (how a DataLoader worker runs it) and the eval forward of one
        # sample (BASELINE configs[0] for config 2), both bounded to a few seconds
        t1 = time.time()
        k1_n = 0
        while time.time() - t1 < 1.5:
            events_ref.micro_sum(*streams[k1_n % cpu_batch], Tm, *sensor)
            k1_n += 1
        res['k1_numpy_events_per_s_1thread'] = round(k1_n * n_events / (time.time() - t1))
    model.eval()
    one = torch.from_numpy(np.zeros((1, 1, Tm, 2) + canvas, np.float32))
    one[0, 0, :, :, :sensor[0], :sensor[1]] = torch.from_numpy(np.asarray(make_frames(0), np.float32))
    with torch.no_grad():
        model(one)
        sj_ref.reset_net(model)
        t2, ev_n = time.time(), 0
        while time.time() - t2 < 2.0:
            model(one)
            sj_ref.reset_net(model)
            ev_n += 1
    res['eval_forward_1_sample_frames_per_s'] = round(ev_n / (time.time() - t2), 2)
    return res


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


class _StandIn(torch.nn.Module):
    """CPU stand-in with the module names the trainer's overlapped exchange cuts at (sampler | backbone.backbone || neck / head)"""

    def __init__(self):
        super().__init__()
        self.embedding = torch.nn.Linear(16, 16)
        self.backbone = torch.nn.Module()
        self.backbone.backbone = torch.nn.Linear(16, 16)
        self.head = torch.nn.Linear(16, 4)

    def forward(self, x, targets=None):
        return {'total_loss': self.head(torch.tanh(self.backbone.backbone(self.embedding(x)))).sum()}


def selftest_cpu(args, world, rank):
    """The N-rank protocol of this file without a GPU: gloo rendezvous from the launcher's environment, the trainer's step object
    (yolox/core/trainer.py::TrainStep: forward, split backward, bucketed gradient exchange, optimizer step) on a stand-in module,
    barrier-bracketed timed region, max over ranks, one JSON line from rank 0."""
    from eas_snn_amd import workloads
    from eas_snn_amd.parallel import BucketedGradAllReduce
    from yolox.core.trainer import DEFAULT_CUT, DEFAULT_LOWER, TrainStep
    if os.environ.get('EAS_BENCH_SELFTEST_FAIL_RANK') == str(rank):     # test hook: this rank dies before the rendezvous
        sys.exit(7)
    w = workloads.get(args.config)
    # The order of the GPU path at N > 1 (main() below, Trainer.train_one_iter): step object with an UNBOUND exchange -> record
    # (capture(restore=True): warm-up launches, state put back; on the CPU nothing is recorded, the protocol is the same) -> process group
    # -> bind (rank 0's parameters into place) -> steps.
    torch.manual_seed(rank)
    net = _StandIn()
    twin = _StandIn()
    twin.load_state_dict(net.state_dict())              # this rank's initial values, for the group-first order below
    exchange = BucketedGradAllReduce(net, split=DEFAULT_LOWER, world=world) if world > 1 else None
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    x = torch.full((8, 16), float(rank + 1)) + torch.arange(16.0) * 0.01
    step = TrainStep(net, opt, lambda: (x, None), exchange=exchange, reset=False, defer_wgrad=False, cut=DEFAULT_CUT)
    assert exchange is None or not exchange.bound
    step.capture(warm=2, restore=True)
    if world > 1:
        assert not dist.is_initialized()
        dist.init_process_group('gloo')
        exchange.bind()
        assert exchange.bound and exchange.world == world
    losses = [float(step().detach()) for _ in range(3)]
    same_as_group_first = None
    if world > 1:
        # the order of rounds 1-5 (group first, the exchange's constructor broadcasts, steps from the first one on) on the twin: same losses
        opt2 = torch.optim.Adam(twin.parameters(), lr=1e-2)
        ex2 = BucketedGradAllReduce(twin, split=DEFAULT_LOWER)
        step2 = TrainStep(twin, opt2, lambda: (x, None), exchange=ex2, reset=False, defer_wgrad=False, cut=DEFAULT_CUT)
        losses2 = [float(step2().detach()) for _ in range(3)]
        assert losses == losses2, f'record-then-rendezvous differs from rendezvous-first: {losses} vs {losses2}'
        assert all(torch.equal(p, q) for p, q in zip(net.parameters(), twin.parameters()))
        same_as_group_first = True
        ref = _StandIn()
    for g in opt.param_groups:
        g['lr'] = 0.0                                   # the timed steps below leave the parameters alone (the gradient check needs that)
    for _ in range(args.warmup):
        step()
    if world > 1:
        ref.load_state_dict(net.state_dict())           # a plain-backward twin of the current values
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        # the split backward + bucketed exchange must equal: plain backward on every rank, gradients averaged over the ranks
        ref(x)['total_loss'].backward()
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            want = q.grad.clone()
            dist.all_reduce(want, op=dist.ReduceOp.SUM)
            want /= world
            assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), f'{n}: bucketed exchange differs from the averaged plain backward'
            lo, hi = p.grad.clone(), p.grad.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), 'ranks hold different averaged gradients'
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'rccl_ranks': dist.get_world_size() if world > 1 else 1, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': round(float(el) / args.steps * 1e3, 4), 'backend': 'gloo',
                          'gradient_exchange': (f'{exchange.nbuckets} buckets' if exchange is not None else None),
                          'order': 'record (restore) -> init_process_group -> bind -> steps', 'same_losses_as_group_first_order': same_as_group_first,
                          'first_losses': losses, 'config': {'workload': w['name'], 'config': w['config']},
                          'spawned_by_bench': os.environ.get('EAS_BENCH_SPAWNED') == '1'}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def count_launches(step, ops_lib):
    """Device launches of ONE eager training step: the library's own kernels from its launch counter (exact), the ATen kernels and the
    runtime's copy / fill kernels from a torch.profiler pass over one more step (None when the profiler is not available)."""
    import torch
    torch.cuda.synchronize()
    c0 = ops_lib.eas_launch_counter()
    step.eager()
    torch.cuda.synchronize()
    own = int(ops_lib.eas_launch_counter() - c0)
    aten = runtime = None
    try:
        from torch.autograd import DeviceType
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step.eager()
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if e.device_type == DeviceType.CUDA]
        if names:
            is_rt = lambda n: n.startswith(('Memcpy', 'Memset')) or '__amd_rocclr' in n or 'hipMemcpy' in n or 'hipMemset' in n
            runtime = sum(1 for n in names if is_rt(n))
            aten = sum(1 for n in names if not is_rt(n) and ('at::' in n or 'c10::' in n))
    except Exception:
        pass
    out = {'eas_kernels': own, 'aten_kernels': aten, 'runtime_copies_fills': runtime}
    out['total'] = own + (aten or 0) + (runtime or 0) if aten is not None else None
    return out


def family_table(summ, per):
    return {k: dict(calls=v['calls'], ms=round(v['ms'] / per, 4),
                    GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None,
                    **({'TFLOPs': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1)} if v['flops'] > 0 else {}),
                    roofline_frac=round(v['roof_ms'] / v['ms'], 3) if v['ms'] > 0 else None) for k, v in summ.items()}


def dominant_roofline(summ):
    """the roofline object of the family with the most time in a KernelTimer summary"""
    dom = max(summ, key=lambda k: summ[k]['ms'])
    d = summ[dom]
    sec = d['ms'] * 1e-3
    if d['flops'] > 0:
        ceiling = BF16_MFMA_PEAK_TF * d['flops'] / d['issue_flops']
        achieved = d['flops'] / sec / 1e12
        r = {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': round(ceiling, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / ceiling, 4),
             'peak_is': 'bf16 term ceiling: dense bf16 MFMA peak 2500 TFLOP/s / bf16 term products per fp32 product of this family\'s mix',
             'bf16_issue_frac': round(d['issue_flops'] / sec / 1e12 / BF16_MFMA_PEAK_TF, 4)}
    else:
        achieved = d['bytes'] / sec / 1e9
        r = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4)}
    r.update({'kernel': dom, 'avg_launch_ms': round(d['ms'] / d['calls'], 4), 'frac_of_per_launch_roofline': round(d['roof_ms'] / d['ms'], 4),
              'algorithmic_bytes_per_call': round(d['bytes'] / d['calls'])})
    return r


def ema_off_side_figure(trainer, step, inputs_fn, n=10):
    """Side figure (never `value`): the same step WITHOUT the weight average, i.e. what rounds 1-5 measured (exp.ema = False).  The headline
    step keeps the reference's default (exp.ema = True, event_yolox_base.py:116: ModelEMA.update after every optimizer.step(),
    trainer.py:120-121) -- the average is made by the optimizer's own launch (FusedAdam.attach_ema, eas_adam_step_ex), so the step stays one
    HIP-graph replay.  Both graphs are timed back to back in this process, alternating, so that the difference is the average and not the box."""
    ema = step.ema
    if ema is None or getattr(ema, '_fused_in', None) is not trainer.optimizer:
        return {'error': 'the headline step of this run carries no weight average made by the optimizer launch'}
    updates = trainer.optimizer.ema_updates_on_device()
    trainer.optimizer.detach_ema()           # the headline graph keeps replaying with ITS table (average included); new recordings get none
    try:
        step_off = trainer.step_fn(inputs_fn, ema=None)
        form = step_off.capture(warm=2)

        def clock(fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3
        off, on = [], []
        for _ in range(3):
            on.append(clock(step.replay))
            off.append(clock(step_off.replay))
        assert torch.isfinite(step_off.loss), 'the step without the weight average produced a non-finite loss'
        return {'ema_on_ms_per_step': round(min(on), 3), 'ema_off_ms_per_step': round(min(off), 3), 'difference_ms': round(min(on) - min(off), 3),
                'launch': form, 'updates_on_device_before': updates,
                'note': 'ModelEMA (decay 0.9998, ramp 2000) over every floating-point state-dict entry, made inside the one Adam launch of the '
                        'headline step; best of 3 x %d replays each of the headline graph and of the same step recorded without it, alternating' % n}
    finally:
        trainer.optimizer.attach_ema(ema, trainer.bare_model)


def parity_against_oracle(w, model, raw, dev):
    """Part of the CPU-baseline leg (the only place this file touches oracle/, and only as the checker): the HIP path against the CPU
    restatement on two samples of THIS run's input with THIS run's weights (the state after the timed steps) -- event binning bit-exact or
    not, and the share of eval-mode detection logits within 1e-4 (relative + absolute) end to end at the bench canvas.  The spiking backbone
    amplifies a rounding-level spike flip (chaotic cascade), so the end-to-end share is reported as measured, next to
    the per-layer teacher-forced figures of the GPU test-suite (profiles/parity_teacher_forced_latest.json when its source hashes match)."""
    from oracle import events_ref, model_ref, sj_ref
    from spikingjelly.activation_based import functional
    from eas_snn_amd import data
    out = {'samples': 2, 'canvas': list(w['canvas'])}
    try:
        B, Tm, sensor, canvas = 2, w['Tm'], tuple(w['sensor']), tuple(w['canvas'])
        if w['input'] == 'events':
            offs = raw['offsets'][:B + 1]
            n = int(offs[B])
            ev = {k: raw[k][:n] for k in ('t', 'x', 'y', 'p')}
            frames = data.events_to_frames(dict(ev, offsets=offs), Tm, sensor, canvas)
            counts = events_ref.micro_sum_batch(*(ev[k].cpu().numpy() for k in ('t', 'x', 'y', 'p')), offs.cpu().numpy(), Tm, *sensor)
            got = frames[:, 0, :, :, :sensor[0], :sensor[1]].cpu().numpy()
            pad_clear = bool(frames.sum().item() == float(got.sum()))
            out['binning'] = 'bit_exact' if (np.array_equal(got, counts.astype(np.float32)) and pad_clear) else 'MISMATCH'
            out['binning_events_checked'] = n
        else:
            from eas_snn_amd import ops
            frames = ops.stacked_hist_event_sum(raw[:B], *canvas)
            want = np.stack([events_ref.stacked_hist_event_sum(raw[b].cpu().numpy(), Tm, *sensor)[0] for b in range(B)])
            got = frames[:, 0, :, :, :sensor[0], :sensor[1]].cpu().numpy()
            out['binning'] = 'bit_exact' if np.array_equal(got, want.astype(np.float32)) else 'MISMATCH'
        ref = model_ref.build_model(**w['oracle'])
        ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
        was_training = model.training
        model.eval()
        ref.eval()
        # where the two forwards part: the sampler's output and the four backbone stages (spike tensors or firing rates) -- the share of
        # elements that differ.  A rounding-level flip of a neuron that sits on its threshold is amplified stage by stage;
        # the per-layer bar with the oracle's input at every layer is the teacher-forced figures below.
        stages, seen = ['embedding', 'backbone.backbone.dark2', 'backbone.backbone.dark3', 'backbone.backbone.dark4', 'backbone.backbone.dark5'], {}

        def tap(store, name):
            def hook(mod, inp, outp):
                o = outp[0] if isinstance(outp, (tuple, list)) else outp
                if torch.is_tensor(o):
                    store[name] = o.detach().float().cpu().numpy()
            return hook
        handles = []
        for tag, net in (('hip', model), ('ref', ref)):
            seen[tag] = {}
            for name in stages:
                try:
                    handles.append(net.get_submodule(name).register_forward_hook(tap(seen[tag], name)))
                except AttributeError:
                    pass
        try:
            with torch.no_grad():
                lh = model(frames).cpu().numpy()
                lr = ref(frames.cpu()).numpy()
        finally:
            for h in handles:
                h.remove()
        functional.reset_net(model)
        sj_ref.reset_net(ref)
        model.train(was_training)
        out['differing_share_by_stage'] = {
            name.split('.')[-1]: round(float((~np.isclose(seen['hip'][name], seen['ref'][name], rtol=1e-4, atol=1e-5)).mean()), 8)
            for name in stages if name in seen['hip'] and name in seen['ref'] and seen['hip'][name].shape == seen['ref'][name].shape}
        close = np.isclose(lh, lr, rtol=1e-4, atol=1e-4)
        rel = np.abs(lh - lr) / (np.abs(lr) + 1e-4)
        out.update({'logits_within_1e-4_share_end_to_end': round(float(close.mean()), 6), 'logits_median_rel_err': float(np.median(rel)),
                    'logits_compared': int(lh.size), 'tolerance': 'rtol 1e-4 + atol 1e-4 (north_star: 1e-4 relative)'})
    except Exception as exc:                                     # a side figure must not take the line with it
        out['error'] = f'{type(exc).__name__}: {exc}'[:300]
    # per-layer figures of the teacher-forced GPU tests (tests/parity_report.py writes them; quoted only for the kernel sources of this build)
    try:
        with open(os.path.join(ROOT, 'profiles', 'parity_teacher_forced_latest.json')) as fh:
            tf = json.load(fh)
        recorded, now = (tf.get('_meta') or {}).get('csrc_sha16') or {}, csrc_hashes()
        stale = [f for f in now if recorded.get(f) != now.get(f)]
        if stale:
            out['teacher_forced'] = {'unavailable': f'stale: {", ".join(stale[:4])} changed after tests/parity_report.py ran'}
        else:
            out['teacher_forced'] = {k: tf[k] for k in ('model', 'canvas', 'eval', 'train') if k in tf}
            steps = tf.get('train_step_vs_reference_fixture') or {}
            if steps:           # one training step against the reference's own step (fixtures from the unmodified reference classes), condensed
                out['train_step_vs_reference_fixture'] = {
                    'fixtures': sorted(steps), 'gradient_elements': sum(f['gradient_elements'] for f in steps.values()),
                    'gradient_elements_within_1e-3_share_min': min(f['gradient_elements_within_tol_share'] for f in steps.values()),
                    'loss_rel_err_max': max(max(f['loss_rel_err'].values()) for f in steps.values()),
                    'gradnorm_rel_err_median_max': max(f['gradnorm_rel_err_median'] for f in steps.values()),
                    'gradnorm_rel_err_max': max(f['gradnorm_rel_err_max'] for f in steps.values())}
    except (OSError, ValueError):
        out['teacher_forced'] = {'unavailable': 'profiles/parity_teacher_forced_latest.json not present'}
    return out


def canvas_640_figure(dev, batch, events):
    """Side figure (never `value`): the README recipe's canvas.  The reference's published commands never set input_size, so they train at
    the default (640, 640) (yolox/exp/event_yolox_base.py:66,137) through the letterbox resize of the dataset
    (yolox/data/datasets/gen1.py:433-455): workload '2b' = config 2's model behind eas_counts_letterbox.  A few HIP-graph replays of the
    trainer's step + the kernel families of three eager steps."""
    import gc
    from eas_snn_amd import ops, workloads
    w = workloads.get('2b')
    trainer, model, step = workloads.build_trainer(w, batch, dev, events=events)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    launch = 'eager launches'
    if trainer.net is trainer.bare_model and os.environ.get('EAS_BENCH_GRAPH', 'auto') != '0':
        launch = step.capture(warm=2)
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    assert torch.isfinite(loss), 'the 640x640 step produced a non-finite loss'
    ops.check_tags('the 640x640 steps')
    timer = ops.KernelTimer()
    side_batch, ops.WGRAD_SIDE_BATCH = ops.WGRAD_SIDE_BATCH, 0
    ops.set_timer(timer)
    for _ in range(3):
        step.eager()
    torch.cuda.synchronize()
    ops.set_timer(None)
    ops.WGRAD_SIDE_BATCH = side_batch
    summ = timer.summary()
    r = dominant_roofline(summ)
    out = {'workload': w['name'], 'batch': batch, 'ms_per_step': round(ms, 3), 'event_frames_per_s': round(batch / ms * 1e3, 1), 'launch': launch,
           'dominant_family': r['kernel'], 'frac': r['frac'], 'frac_of_per_launch_roofline': r['frac_of_per_launch_roofline'],
           'fused_detection_loss': bool(getattr(model.head, '_eas_fused_loss_used', False)),
           'hip_kernel_ms_per_step': {k: round(v['ms'] / 3, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['ms'])[:6]}}
    del trainer, model, step
    gc.collect()
    torch.cuda.empty_cache()
    return out


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
    if args.canvas640_child:
        import eas_snn_amd
        from eas_snn_amd import ops
        eas_snn_amd.hip_library()
        torch.cuda.set_stream(torch.cuda.Stream())
        ops.set_state_writeback(False)
        print(json.dumps(canvas_640_figure(torch.device('cuda', local_rank), args.batch or 64, args.events)), flush=True)
        return
    dev = torch.device('cuda', local_rank)
    force_ddp = os.environ.get('EAS_BENCH_FORCE_DDP') == '1'     # development: exercise the exchange / RCCL path with one rank
    if world > 1 or force_ddp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if force_ddp and world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        # The RCCL process group is created AFTER the step's HIP graphs are recorded (trainer.join_ranks below): ProcessGroupNCCL's watchdog
        # thread polls events of collectives in flight, and no capture may be open while it does.  Until then rank and world size come from
        # the rendezvous parameters (yolox.utils.dist.defer_process_group, what yolox.core.launch does for its workers too).
        from datetime import timedelta
        import eas_snn_amd  # noqa: F401  (puts the compat namespace -- yolox, spikingjelly -- on the path)
        from yolox.utils import dist as comm
        comm.defer_process_group('nccl', 'env://', world, rank, timedelta(minutes=30), local_size=world, local_rank=local_rank, device_id=dev)

    # everything runs on one non-default stream: autograd's AccumulateGrad nodes remember the stream they were created on,
    # and a later graph capture breaks if that was the default stream
    torch.cuda.set_stream(torch.cuda.Stream())

    import types
    import eas_snn_amd
    from eas_snn_amd import ops, workloads
    eas_snn_amd.hip_library()
    ops.set_state_writeback(False)           # every step ends with reset_net (as yolox/core/trainer.py:115-117)

    w = workloads.get(args.config)
    batch = args.batch or w['batch']
    # THE STEP IS THE TRAINER'S: model, optimizer, gradient exchange and the iteration itself (eager or HIP-graph replays) come from
    # yolox.core.Trainer (compat/yolox/core/trainer.py, what tools/train_event.py runs); this file only feeds it and times it.
    multi = world > 1 or force_ddp
    trainer, model, step = workloads.build_trainer(w, batch, dev, events=args.events, world=world, rank=rank, force_exchange=force_ddp)
    inputs_fn = step.inputs_fn
    h2d = _H2DFeeder(step.raw_inputs) if args.h2d else None

    def one():
        if h2d is not None:
            h2d.next()
        return step()

    # Launch mode.  The step has no host synchronisation, so it can be captured into HIP graphs and replayed (TrainStep.capture): one
    # graph on one GPU -- the warm-up times a few eager steps and a few replays and the timed region uses whichever was faster --
    # and graph replays with the eager RCCL all-reduces between them at N > 1.  DistributedDataParallel (EAS_DP=ddp) stays eager.
    # EAS_BENCH_GRAPH=0/1 forces eager/graph.
    mode = os.environ.get('EAS_BENCH_GRAPH', 'auto')
    for _ in range(max(args.warmup - 2, 1)):
        loss = one()
    torch.cuda.synchronize()
    t_a = time.perf_counter()
    one()
    loss = one()
    t_enq = time.perf_counter() - t_a
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_a
    probe = {}
    if mode in ('1', 'auto') and trainer.net is trainer.bare_model and step.ema_capturable():
        # multi: recorded before the process group exists, the warm-up launches' effect on model and optimizer state put back afterwards
        # (every rank then takes rank 0's parameters in join_ranks, as DistributedDataParallel's constructor would hand them out)
        step.capture(warm=3, restore=multi)
        if not multi and mode == 'auto':
            def clock(fn, n=4):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / n * 1e3
            probe = {'eager_ms': round(clock(step.eager), 3), 'graph_ms': round(clock(step.replay), 3)}
            if probe['eager_ms'] < probe['graph_ms']:
                step.uncapture()                     # eager is faster on this host right now
    launch = step.launch
    if multi:
        trainer.join_ranks()                     # init_process_group (RCCL), rank 0's parameters into place
        assert dist.is_initialized() and dist.get_world_size() == world and (trainer.exchange is None or trainer.exchange.bound)
        # The first collectives build the communicator's rings: not in the timed region.  Four steps, because the recording above left no
        # trace (restore) where the one-GPU path keeps its three warm-up launches and the closing replay: both paths reach the timed region
        # after the same number of training steps (tests/test_gpu_model.py compares their losses).
        for _ in range(4):
            one()
    timer = ops.KernelTimer() if rank == 0 else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.check_tags('the timed steps')        # a tensor tagged "spikes" that was not: the measured arithmetic would not be the fp32 one
    # Per-kernel durations behind `roofline`: HIP events around every C-ABI call, on the stream the kernels are launched on, in
    # eager steps of this same process right after the timed region (same kernels, same data).  Not inside it: events cannot be
    # recorded into a captured graph, and ~1400 event records per eager step would slow the timed region itself by ~10 %.
    # With more than one rank every rank runs these steps (the all-reduces need all of them); rank 0 records.
    timed_steps = 3
    # (serial launches for these steps: the weight-gradient slab kernels of the timed region run on a side stream under the backward chain,
    # where an event pair around one of them -- or around a chain kernel they share the chip with -- measures the overlap, not the kernel)
    side_batch, ops.WGRAD_SIDE_BATCH = ops.WGRAD_SIDE_BATCH, 0
    ops.set_timer(timer)
    for _ in range(timed_steps):
        step.eager()
    torch.cuda.synchronize()
    ops.set_timer(None)
    ops.WGRAD_SIDE_BATCH = side_batch
    # The first kernel of a step (raw input -> frames: K1 / the stacked-histogram reduction) starts on an idle queue in those eager steps,
    # so its event pair also spans the host's submission latency (128 us recorded for a 65 us kernel).  Re-timed back to back: eleven calls,
    # the first dropped, scaled to the three steps the other entries cover.
    if rank == 0:
        t_in = ops.KernelTimer()
        ops.set_timer(t_in)
        for _ in range(11):
            step.inputs_fn()
        torch.cuda.synchronize()
        ops.set_timer(None)
        for name, items in t_in.rec.items():
            if name in timer.rec and len(items) == 11 and len(timer.rec[name]) == timed_steps:
                timer.rec[name] = items[4:4 + timed_steps]           # calls from the middle of the run: the queue is full
                timer.input_retimed = name
    # inference side figure: eval-mode forward + reset_net of the same batch (no loss, no backward), replayed as a HIP graph like the
    # evaluator does (yolox/evaluators/event_evaluator.py), for the model as trained and after yolox.utils.fuse_model (tools/eval_event.py --fuse)
    eval_fps = eval_roofline = None
    if rank == 0 and not multi and os.environ.get('EAS_BENCH_NO_EVAL') != '1':      # (development: profiles of the training step alone)
        import copy
        from spikingjelly.activation_based import functional
        from yolox.utils import fuse_model

        def eval_rate(net):
            net.eval()

            def fwd():
                net(inputs_fn()[0])
                functional.reset_net(net)
            # (ops.frozen_weights: what EventEvaluator.evaluate opens around its loop -- weights packed once per evaluation, not per batch)
            with torch.no_grad(), ops.frozen_weights(net):
                for _ in range(2):
                    fwd()
                torch.cuda.synchronize()
                g_ev = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_ev):
                    fwd()
                g_ev.replay()
                torch.cuda.synchronize()
                t_e = time.perf_counter()
                for _ in range(10):
                    g_ev.replay()
                torch.cuda.synchronize()
                return round(10 * batch / (time.perf_counter() - t_e), 1)
        eval_fps = {'value': eval_rate(model), 'fuse_model': eval_rate(fuse_model(copy.deepcopy(model)))}
        # roofline of the forward pass itself (north_star's first clause; the reference's timers: yolox/evaluators/event_evaluator.py:190-212):
        # the kernel families of three eager eval forwards (events -> frames -> sampler -> backbone -> neck -> head, reset_net)
        t_ev = ops.KernelTimer()
        model.eval()
        ops.set_timer(t_ev)
        with torch.no_grad():
            for _ in range(3):
                model(inputs_fn()[0])
                functional.reset_net(model)
        torch.cuda.synchronize()
        ops.set_timer(None)
        ev_summ = t_ev.summary()
        eval_roofline = dominant_roofline(ev_summ)
        ev_ms = sum(v['ms'] for v in ev_summ.values()) / 3
        ev_roof = sum(v['roof_ms'] for v in ev_summ.values()) / 3
        # (no PMC pass over the eval forward exists: its traffic is not claimed)
        eval_roofline.update({'traffic': None, 'hip_kernel_ms_per_batch': family_table(ev_summ, 3),
                              'hip_kernels_ms_per_batch': round(ev_ms, 3), 'replayed_ms_per_batch': round(batch / eval_fps['value'] * 1e3, 3),
                              'all_families_frac_of_per_launch_roofline': round(ev_roof / ev_ms, 4),
                              'kernel_timing': 'HIP events on the launch stream, 3 eager eval forwards of this process'})
        model.train()
    # launches of one step (every rank runs the steps -- the exchange needs all of them --, rank 0 reports)
    launches = count_launches(step, eas_snn_amd.hip_library()) if (rank == 0 and not multi) else None
    # (side figures that record further graphs or start a child process: never under a profiler -- rocprofv3 follows the child, whose kernel
    # statistics would land next to, or over, this process's)
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROFILER_', 'ROCPROF_')) for k in os.environ)
    ema_figure = None
    if rank == 0 and not multi and step.graphs is not None and os.environ.get('EAS_BENCH_NO_EMA') != '1' and not profiled:
        try:
            ema_figure = ema_off_side_figure(trainer, step, inputs_fn)
        except Exception as exc:
            ema_figure = {'error': f'{type(exc).__name__}: {exc}'[:300]}
    canvas640 = None
    if rank == 0 and not multi and w['config'] == 2 and args.batch is None and os.environ.get('EAS_BENCH_NO_640') != '1' and not profiled:
        # in a child process of its own (started like the ranks of launch_ranks, never an exec): a side figure must not be able to take the
        # headline line with it -- a failure is reported in its place
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--canvas640-child', '--batch', os.environ.get('EAS_BENCH_640_BATCH', '64'),
                                '--events', str(args.events)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            last = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            canvas640 = json.loads(last[-1]) if (r.returncode == 0 and last) else {'error': f'rc {r.returncode}: ' + r.stderr.strip()[-300:]}
        except Exception as exc:
            canvas640 = {'error': f'{type(exc).__name__}: {exc}'[:300]}
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el)
    assert torch.isfinite(loss), 'training step produced a non-finite loss'

    if rank == 0:
        frames_total = batch * world * args.steps
        summ = timer.summary()
        fam = {k: dict(calls=v['calls'], ms_per_step=round(v['ms'] / timed_steps, 4),
                       GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['ms'] > 0 else None,
                       **({'TFLOPs': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1)} if v['flops'] > 0 else {}),
                       roofline_frac=round(v['roof_ms'] / v['ms'], 3) if v['ms'] > 0 else None) for k, v in summ.items()}
        dom = max(summ, key=lambda k: summ[k]['ms'])
        d = summ[dom]
        sec = d['ms'] * 1e-3
        traffic, traffic_why = pmc_traffic(dom, 2 if dom in ('eas_bn_lif_bwd', 'eas_bn_silu_bwd') else 1, w['config'])
        common = {'kernel': dom, 'avg_launch_ms': round(d['ms'] / d['calls'], 4),
                  'traffic': traffic,
                  'traffic_source': ('committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes over this command (profiles/pmc_traffic_latest.json, '
                                     'scripts/gpu_profile.sh; configs 2 and 3) on kernel sources whose hashes match this build\'s; not re-measured in this run'
                                     if traffic is not None else traffic_why),
                  'algorithmic_bytes_per_call': round(d['bytes'] / d['calls']), 'hip_kernel_ms_per_step': fam,
                  'hip_share_of_step': round(sum(v['ms'] for v in summ.values()) / timed_steps / (elapsed / args.steps * 1e3), 4),
                  'kernel_timing': 'HIP events on the launch stream, 3 eager steps of this process right after the timed region (all launches on one stream there)'
                                   + (f'; {timer.input_retimed} (the first kernel of a step, which starts on an idle queue there) re-timed in ten back-to-back calls' if getattr(timer, 'input_retimed', None) else '')}
        if d['flops'] > 0:
            # dense convolutions: bounded by the matrix cores.  achieved = algorithmic flops (2 x MAC of the fp32 convolution) per
            # second against the dense f32 MFMA peak (the arithmetic the path reproduces); every fp32 product is formed from
            # 3 (spike inputs) or 6 (general inputs) exact bf16 term products on v_mfma_f32_32x32x16_bf16, so the bf16 flops
            # actually issued and their share of the dense bf16 peak are reported next to it.
            achieved = d['flops'] / sec / 1e12
            # ceiling of the exact-bf16-term scheme itself: the bf16 pipe divided by the term products per fp32 product
            # (3 for spike inputs -> 833 TF, 6 for real-valued inputs -> 417 TF), weighted by this step's mix of the two
            scheme_ceiling = BF16_MFMA_PEAK_TF * d['flops'] / d['issue_flops']
            # peak = what bounds these kernels: the dense bf16 MFMA rate divided by the bf16 term products one exact fp32 product costs
            # (rounds 1-2 quoted the 157.3 TFLOP/s f32-input MFMA peak here; the M-width configurations run ABOVE that rate, which
            # is the point of the term scheme and makes it useless as a ceiling -- kept as vs_f32_mfma_peak)
            roofline = {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': round(scheme_ceiling, 1), 'unit': 'TFLOP/s',
                        'frac': round(achieved / scheme_ceiling, 4),
                        'peak_is': 'bf16 term ceiling (not a hardware figure): the dense bf16 MFMA peak divided by the bf16 term products one exact fp32 product costs',
                        'bf16_issue_frac': round(d['issue_flops'] / sec / 1e12 / BF16_MFMA_PEAK_TF, 4),
                        'peak_note': 'dense bf16 MFMA peak 2500 TFLOP/s / bf16 term products per fp32 product (3 spike inputs, 6 real inputs), weighted by this step',
                        'vs_f32_mfma_peak': round(achieved / F32_MFMA_PEAK_TF, 4), 'f32_mfma_peak_tflops': F32_MFMA_PEAK_TF,
                        'scheme_ceiling_tflops': round(scheme_ceiling, 1), 'frac_of_scheme_ceiling': round(achieved / scheme_ceiling, 4),
                        'mfma_bf16_issued_tflops': round(d['issue_flops'] / sec / 1e12, 1), 'mfma_bf16_peak_tflops': BF16_MFMA_PEAK_TF,
                        'mfma_bf16_util': round(d['issue_flops'] / sec / 1e12 / BF16_MFMA_PEAK_TF, 4),
                        'mfma_busy_measured': mfma_busy_measured(),
                        # per launch max(algorithmic bytes / 8 TB/s, bf16 term flops / 2.5 PFLOP/s), summed over this family's launches,
                        # over the measured time: the small-channel layers are HBM-bound, so the term ceiling alone overstates the headroom
                        'frac_of_per_launch_roofline': round(d['roof_ms'] / d['ms'], 4),
                        'algorithmic_GBps': round(d['bytes'] / sec / 1e9, 1)}
        else:
            achieved = d['bytes'] / sec / 1e9
            roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(achieved / HBM_PEAK_GBS, 4)}
        roofline.update(common)
        # the dominant HBM-bound family as well (the elementwise/BN/LIF kernels of the step)
        hbm_fams = {k: v for k, v in summ.items() if v['flops'] == 0 and v['bytes'] > 0}
        if hbm_fams:
            hk = max(hbm_fams, key=lambda k: hbm_fams[k]['ms'])
            hv = hbm_fams[hk]
            h_traffic, h_why = pmc_traffic(hk, 2 if hk in ('eas_bn_lif_bwd', 'eas_bn_silu_bwd') else 1, w['config'])
            roofline['hbm_dominant'] = {'kernel': hk, 'achieved': round(hv['bytes'] / (hv['ms'] * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                                        'unit': 'GB/s', 'frac': round(hv['bytes'] / (hv['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        'traffic': h_traffic, **({'traffic_unavailable': h_why} if h_traffic is None else {}),
                                        'algorithmic_bytes_per_call': round(hv['bytes'] / hv['calls'])}
        metric = ('event-frames/sec (T=3) SYOLOX-S Gen1 304x240' if w['config'] == 2 else f'event-frames/sec, BASELINE config {w["config"]}')
        line = {'metric': metric, 'value': round(frames_total / elapsed, 2),
                'unit': 'event-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': f"{w['name']}, batch {batch}/GPU, raw input -> frames -> fwd + bwd + Adam + ModelEMA update + reset_net (yolox.core.Trainer step with the reference's defaults: exp.ema = True, trainer.py:120-122; the average is made inside the one Adam launch -- the same step without it: side figure weight_average.ema_off_ms_per_step)" if step.ema is not None else f"{w['name']}, batch {batch}/GPU, raw input -> frames -> fwd + bwd + Adam + reset_net (EMA off)",
                           'baseline_config': w['config'], 'global_batch': batch * world,
                           'events_per_sample': args.events if w['input'] == 'events' else None, 'parallelism': f'dp{world}',
                           'gradient_exchange': ((f'{trainer.exchange.nbuckets} flat bucket(s)' if trainer.exchange is not None else trainer.dp) if multi else None),
                           'weight_gradient_side_stream': (f'slab kernels in batches of {ops.WGRAD_SIDE_BATCH} on one side stream, joined before the batched reduction'
                                                           if ops.WGRAD_SIDE_BATCH > 0 else None),
                           'launch': launch, 'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                           'launch_probe_ms_per_step': probe, 'host_enqueue_share_of_step': round(t_enq / t_tot, 3),
                           'h2d_per_step': (h2d.describe() if h2d is not None else None)},
                'roofline': roofline}
        line['final_loss'] = float(loss.detach())
        line['launches_per_step'] = launches
        if eval_roofline is not None:
            line['roofline_eval'] = eval_roofline
        if canvas640 is not None:
            line['canvas_640'] = canvas640
        if eval_fps is not None:
            line['eval_forward_frames_per_s'] = dict(eval_fps, batch=batch, note='model.eval() forward + reset_net on the same batch, one GPU, HIP-graph replay (as EventEvaluator); fuse_model: after yolox.utils.fuse_model (eval_event.py --fuse)')
        if ema_figure is not None:
            line['weight_average'] = ema_figure
        if world == 1 and not args.no_cpu_baseline:
            line['parity'] = parity_against_oracle(w, model, step.raw_inputs, dev)
            line['cpu_baseline'] = cpu_baseline(w, args.cpu_batch or (8 if w['config'] == 2 else 2), args.events)
        print(json.dumps(line), flush=True)
    if world > 1 or force_ddp:
        dist.destroy_process_group()


class _H2DFeeder:
    """--h2d: every step a copy of the raw input batch crosses PCIe from pinned host memory on a SIDE stream into a staging buffer while
    the previous step computes (the reference's DataPrefetcher shape, yolox/data/data_prefetcher.py:18,31-44); at the start of a step
    the main stream waits for that copy and moves the staged batch into the buffers the (captured) step reads -- a device-to-device
    copy, ~0.06 ms for 115 MB -- then the next transfer starts."""

    def __init__(self, raw):
        self.dev_tensors = list(raw.values()) if isinstance(raw, dict) else [raw]
        self.host = [t.cpu().pin_memory() for t in self.dev_tensors]
        self.stage = [torch.empty_like(t) for t in self.dev_tensors]
        self.stream = torch.cuda.Stream()
        self.bytes = sum(t.numel() * t.element_size() for t in self.host)
        self._issue()

    def _issue(self):
        with torch.cuda.stream(self.stream):
            for s_, h in zip(self.stage, self.host):
                s_.copy_(h, non_blocking=True)

    def next(self):
        main = torch.cuda.current_stream()
        main.wait_stream(self.stream)                 # the transfer issued during the previous step has landed in the staging buffers
        for d, s_ in zip(self.dev_tensors, self.stage):
            d.copy_(s_)                               # device to device, on the main stream, in front of the step
        self.stream.wait_stream(main)                 # the staging buffers are free again once that copy has run
        self._issue()                                 # the next batch crosses PCIe while this step computes

    def describe(self):
        return {'bytes': self.bytes, 'GB_per_s_needed_at_this_step_rate': None, 'source': 'pinned host memory -> staging (side stream) -> step buffers (D2D)'}


def mfma_busy_measured():
    """time-weighted SQ_VALU_MFMA_BUSY_CYCLES share over the dense-convolution kernels of one config-2 step, from the committed counter
    pass (profiles/conv_sq_counters_latest.json, scripts/pmc_conv_step.sh): rocprofv3 --pmc cannot run inside this process"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'conv_sq_counters_latest.json')) as fh:
            c = json.load(fh)
        recorded, now = (c.get('_meta') or {}).get('csrc_sha16') or {}, csrc_hashes()
        if any(recorded.get(f) != now.get(f) for f in set(_CONV_SRC + FAMILY_SOURCES['eas_conv_wgrad'])):
            return None                      # measured on other kernel sources than this build's: not quoted (scripts/pmc_conv_step.sh re-measures)
        rows = [v for k, v in c.items() if not k.startswith('_') and 'smallconv' not in k and v.get('mfma_busy_frac') is not None and v.get('total_us_under_profiler')]
        return round(sum(v['mfma_busy_frac'] * v['total_us_under_profiler'] for v in rows) / sum(v['total_us_under_profiler'] for v in rows), 4)
    except (OSError, KeyError, ValueError, ZeroDivisionError, TypeError):
        return None


if __name__ == '__main__':
    main()
