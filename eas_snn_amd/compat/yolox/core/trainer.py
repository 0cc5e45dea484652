"""Training loop (reference: yolox/core/trainer.py:36-419), reduced to what the hot path needs: model -> device, optimizer,
data-parallel gradient exchange over RCCL, EMA, per-iteration forward / backward / step / reset_net, LR schedule, 'latest'
checkpoint.  Evaluation, TensorBoard/W&B logging and dataset prefetching are out of scope (the synthetic loader yields GPU tensors).

What is different from the reference's loop, and why (MI355X):

* ``TrainStep`` is the iteration itself -- inputs -> forward -> loss -> zero_grad -> backward -> gradient exchange -> optimizer step
  -> reset_net (trainer.py:95-117 of the reference) -- as ONE object that can launch it eagerly or, since the iteration has no host
  synchronisation, as HIP-graph replays: one graph on one GPU; with N > 1 ranks (forward + backward of the head and the neck) |
  (backward of the backbone) | (optimizer + reset) with the RCCL all-reduces launched eagerly between them, the first one on a side
  stream while the backbone's backward replays.  ``bench.py`` measures exactly this object (``Trainer.step_fn``): the measured step
  IS the drop-in step.
* gradient exchange: ``eas_snn_amd.parallel.BucketedGradAllReduce`` (flat buckets, a handful of collectives) instead of
  ``DistributedDataParallel`` (trainer.py:174-176) -- same averaged gradients, per-rank BatchNorm statistics like
  ``broadcast_buffers=False``; ``EAS_DP=ddp`` selects DistributedDataParallel (eager launches only: its hooks cannot be captured).
* the learning rate lives in a device scalar per parameter group, so the schedule keeps working under graph replay.
* the weight average (``ModelEMA.update`` after every step, trainer.py:120-121) is part of ``TrainStep`` and rides in the optimizer's
  one launch (``FusedAdam.attach_ema``): ``ema = True`` -- the default of event_yolox_base.py:116 -- keeps graph replay.
* order of capture and rendezvous: the graphs are recorded at the FIRST iteration, on that iteration's batch, with model / optimizer /
  average state snapshotted and put back afterwards (``capture(restore=True)``: recording needs a few warm-up launches), and only then
  is the process group created (``yolox.utils.ensure_process_group``; ``yolox.core.launch`` defers it) and rank 0's parameters
  broadcast into place.  No HIP-graph capture is ever open while ProcessGroupNCCL's watchdog thread exists -- the thread that aborted
  one run in fourteen with hipErrorCapturedEvent when captures followed the rendezvous.
"""
import contextlib
import os
import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from eas_snn_amd import _lib, ops
from eas_snn_amd.parallel import BucketedGradAllReduce
from yolox.utils import (ModelEMA, adjust_status, all_reduce_norm, ensure_process_group, get_local_rank, get_model_info, get_rank,
                         get_world_size, is_parallel, load_ckpt, save_checkpoint, setup_logger, synchronize, wait_process_group_idle)


def optimizer_capturable(optimizer):
    """True when ``optimizer.step()`` can be recorded into a HIP graph with the learning rate held in a device scalar: torch's Adam /
    AdamW with ``capturable`` (or ``fused``) read lr and step counters on the device.  SGD (the reference's other branch,
    event_yolox_base.py:361-377) and user-supplied optimizers pass ``lr`` as a host number (``alpha=-lr`` -> ``.item()`` on a device
    tensor = a host synchronisation inside the capture), so their iterations are launched eagerly with a float lr."""
    return isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)) and all('capturable' in g for g in optimizer.param_groups)


def _reset_net(model):
    from spikingjelly.activation_based import functional
    functional.reset_net(model)


class _NoCapture(Exception):
    """raised inside TrainStep.capture when recording is not safe; the step stays eager"""


class TrainStep:
    """One training iteration.  ``inputs_fn() -> (inps, targets)`` produces the batch on the device (inside the captured region:
    e.g. the event histogram of raw events held in static buffers, or ``exp.preprocess`` of static input tensors).

    ``eager()`` launches the iteration kernel by kernel; ``capture()`` records it into HIP graphs after which ``__call__`` replays
    them; ``__call__`` before ``capture()`` is ``eager()``.  ``loss`` / ``outputs`` hold the (device) results of the last iteration.

    cut: names of sub-modules whose outputs split the backward pass (two autograd calls: first everything above the cut, then the
    rest) so that the gradient buckets of the upper part are exchanged while the lower part's backward still runs.  Only used
    with an exchange (N > 1)."""

    def __init__(self, model, optimizer, inputs_fn, exchange=None, net=None, reset=True, defer_wgrad=True, cut=(), ema=None):
        self.model, self.optimizer, self.inputs_fn = model, optimizer, inputs_fn
        self.ema = ema                                           # yolox.utils.ModelEMA or None: updated right after the optimizer step
        self.net = model if net is None else net                 # DistributedDataParallel wrapper, if any
        self.exchange = exchange                                 # BucketedGradAllReduce or None
        self.reset = reset
        self.defer = bool(defer_wgrad) and self.net is model     # DDP copies gradients into its buckets inside the pass: not there
        self.cut = tuple(cut) if exchange is not None and exchange.nbuckets > 1 else ()
        self.graphs = None
        self.launch = 'eager launches'
        self.loss = None
        self.outputs = None
        self._cut_pairs = []
        self._side = None

    # ---- the phases of an iteration
    def _split_at(self, obj):
        """the outputs of a cut module, every tensor that carries a graph replaced by a detached leaf (same storage, same eas tags): the
        part of the model above the cut builds its own autograd graph on the leaves, so its backward is a plain ``loss.backward()`` that
        ends at them; the graph below the cut is run afterwards from the gradients the leaves received"""
        if torch.is_tensor(obj):
            if not (obj.requires_grad and obj.grad_fn is not None):
                return obj
            leaf = obj.detach().requires_grad_(True)
            for k, v in obj.__dict__.items():
                if k.startswith('_eas_'):
                    setattr(leaf, k, v)
            self._cut_pairs.append((obj, leaf))
            return leaf
        if isinstance(obj, dict):
            return type(obj)((k, self._split_at(v)) for k, v in obj.items())
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._split_at(v) for v in obj)
        return obj

    def _cut_hook(self, module, inputs, output):
        return self._split_at(output)
    # the cut handles spike planes (ghost tensors keep their ``_eas_*`` attributes on the detached leaves): ``ops.packed_weights`` must not
    # switch the model to fp32 spikes because of this hook, or N > 1 ranks would run a different kernel set than one rank
    _cut_hook._eas_planes_safe = True

    def _forward(self):
        inps, targets = self.inputs_fn()
        hooks = []
        self._cut_pairs = []
        for name in self.cut:
            hooks.append(self.model.get_submodule(name).register_forward_hook(self._cut_hook))
        try:
            self.outputs = self.net(inps, targets)
        finally:
            for h in hooks:
                h.remove()
        self.loss = self.outputs['total_loss']
        self.optimizer.zero_grad(set_to_none=True)

    def _backward_upper(self):
        """backward of everything above the cut (head, neck): its parameter gradients and the gradients reaching the cut tensors"""
        with ops.deferred_wgrad_reductions(self.defer):
            self.loss.backward()
        self._cut_grads = [(orig, leaf.grad) for orig, leaf in self._cut_pairs if leaf.grad is not None]
        self._cut_pairs = []
        self.exchange.pack(0)

    def _backward_lower(self):
        ts, gs = [t for t, _ in self._cut_grads], [g for _, g in self._cut_grads]
        with ops.deferred_wgrad_reductions(self.defer):
            torch.autograd.backward(ts, gs)
        self._cut_grads = None
        for b in range(1, self.exchange.nbuckets):
            self.exchange.pack(b)

    def _backward_all(self):
        with ops.deferred_wgrad_reductions(self.defer):
            self.loss.backward()
        if self.exchange is not None:
            for b in range(self.exchange.nbuckets):
                self.exchange.pack(b)

    def _update(self):
        if self.exchange is not None:
            self.exchange.attach()
        self.optimizer.step()
        if self.ema is not None:
            self.ema.update(self.model)         # trainer.py:120-121 of the reference; a no-launch count when the optimizer's launch made it
        if self.reset:
            _reset_net(self.model)

    def _reduce_upper_async(self):
        """bucket 0's all-reduce on the side stream, behind everything the main stream has queued so far"""
        if not self.exchange.flat[0].is_cuda:           # CPU tensors (gloo tests): nothing to overlap with
            self.exchange.reduce(0)
            return
        if self._side is None:
            self._side = _lib.private_stream()
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self.exchange.reduce(0)

    def _reduce_rest(self):
        for b in range(1, self.exchange.nbuckets):
            self.exchange.reduce(b)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    # ---- launch forms
    def eager(self):
        self._forward()
        if self.cut and self._cut_pairs:
            self._backward_upper()
            self._reduce_upper_async()
            self._backward_lower()
            self._reduce_rest()
        else:
            self._backward_all()
            if self.exchange is not None:
                for b in range(self.exchange.nbuckets):
                    self.exchange.reduce(b)
        self._update()
        return self.loss

    def make_capturable(self):
        """Adam's step counters live on the host in eager mode; a captured step needs them (and the learning rate) on the device"""
        dev = next(self.model.parameters()).device
        for gr in self.optimizer.param_groups:
            if 'capturable' in gr:
                gr['capturable'] = True
        for st_ in self.optimizer.state.values():
            if torch.is_tensor(st_.get('step')):
                st_['step'] = st_['step'].to(dev)

    def ema_capturable(self):
        """a weight average can be part of a captured step only when the optimizer's launch makes it (device-side decay ramp)"""
        return self.ema is None or getattr(self.ema, '_fused_in', None) is self.optimizer

    # ---- state that the warm-up launches of a capture must not leave behind
    def _snapshot(self):
        mods = [self.model] + ([self.ema.ema] if self.ema is not None else [])
        tensors = [t for m in mods for t in m.state_dict().values()]
        snap = {'tensors': [(t, t.clone()) for t in tensors], 'opt': {}, 'ema_updates': None if self.ema is None else self.ema.updates}
        for p, st_ in self.optimizer.state.items():
            snap['opt'][p] = {k: v.clone() for k, v in st_.items() if torch.is_tensor(v)}
        return snap

    @torch.no_grad()
    def _restore(self, snap):
        """in place: every address a recorded graph holds stays valid"""
        for t, saved in snap['tensors']:
            t.copy_(saved)
        for p, st_ in self.optimizer.state.items():
            before = snap['opt'].get(p)
            for k, v in st_.items():
                if not torch.is_tensor(v):
                    continue
                if before is not None and k in before:
                    v.copy_(before[k])
                else:
                    v.zero_()                                   # created by the warm-up steps: Adam's initial moments and step count are zeros
        if self.ema is not None:
            self.ema.updates = snap['ema_updates']
            fused = getattr(self.ema, '_fused_in', None)
            if fused is not None and fused._eas_ema is not None:
                fused._eas_ema['counter'].fill_(float(self.ema.updates))

    def capture(self, warm=3, restore=False):
        """record the iteration into HIP graph(s); returns a description of the launch form.  Not with DistributedDataParallel.

        restore: model, optimizer and weight-average state are snapshotted first and put back (in place) at the end, so the ``warm``
        eager launches and the closing replay leave no trace -- the form for recording BEFORE training starts (and before the process
        group exists).  On a model without a device there is nothing to record: the warm-up / restore protocol runs and the step stays
        eager (the CPU tests walk the N-rank order with it)."""
        if self.net is not self.model:
            raise RuntimeError('DistributedDataParallel iterations cannot be captured (reducer hooks); use the bucketed exchange')
        if not self.ema_capturable():
            raise RuntimeError('a step with a weight average can only be captured when FusedAdam.attach_ema makes the average')
        on_gpu = next(self.model.parameters()).is_cuda
        snap = self._snapshot() if restore else None
        if on_gpu:
            self.make_capturable()
        for _ in range(warm):
            self.eager()
        if not on_gpu:
            if restore:
                self._restore(snap)
            return self.launch
        torch.cuda.synchronize()
        # A process group that exists already (a caller that did not defer it): ProcessGroupNCCL's watchdog thread polls the events of
        # collectives in flight and must hold none while a capture is open -- wait for exactly that condition (flight recorder), record with
        # thread_local error mode (the capture polices only the capturing thread's own API calls; kernels launched on the capturing
        # stream by the autograd thread are recorded either way).  Without the recorder the step stays eager rather than risk the abort.
        live_group = dist.is_initialized()
        mode = 'thread_local' if live_group else 'global'

        if getattr(self, '_capture_stream', None) is None:
            self._capture_stream = _lib.private_stream()      # never a pooled stream: ProcessGroupNCCL's own stream is one of those

        def graph(g, pool=None):
            torch.cuda.synchronize()
            if live_group and not wait_process_group_idle():
                raise _NoCapture()
            return torch.cuda.graph(g, pool=pool, stream=self._capture_stream, capture_error_mode=mode)
        scope = self.optimizer.capture_scope() if hasattr(self.optimizer, 'capture_scope') else contextlib.nullcontext()
        counted = None if self.ema is None else self.ema.updates
        try:
            with scope:
                self._record(graph)
        except _NoCapture:
            self.graphs, self.launch = None, 'eager launches (a live process group without a flight recorder: nothing recorded)'
            if restore:
                self._restore(snap)
            return self.launch
        if self.ema is not None:
            self.ema.updates = counted                # recording ran ema.update()'s count, but no kernel: the host mirror follows the device counter
        torch.cuda.synchronize()
        self.replay()                                 # warm-up replay
        if restore:
            self._restore(snap)
            torch.cuda.synchronize()
        return self.launch

    def _record(self, graph):
        if self.exchange is None:
            g = torch.cuda.CUDAGraph()
            with graph(g):
                self._forward()
                self._backward_all()
                self._update()
            self.graphs = (g,)
            self.launch = 'hip-graph replay of the whole step'
        elif self.cut:
            pool = torch.cuda.graph_pool_handle()
            g_a, g_b, g_c = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with graph(g_a, pool):
                self._forward()
                self._backward_upper()
            self._reduce_upper_async()
            with graph(g_b, pool):
                self._backward_lower()
            self._reduce_rest()
            with graph(g_c, pool):
                self._update()
            self.graphs = (g_a, g_b, g_c)
            self.launch = ('three hip-graph replays per step (fwd + bwd head/neck | bwd backbone | adam + reset); RCCL all-reduces eager '
                           'between them, the first one on a side stream under the backbone backward')
        else:
            pool = torch.cuda.graph_pool_handle()
            g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with graph(g_a, pool):
                self._forward()
                self._backward_all()
            for b in range(self.exchange.nbuckets):
                self.exchange.reduce(b)
            with graph(g_b, pool):
                self._update()
            self.graphs = (g_a, g_b)
            self.launch = 'two hip-graph replays per step (fwd+bwd+pack | adam+reset) with the eager RCCL all-reduce between them'

    def replay(self):
        gs = self.graphs
        if self.ema is not None:
            self.ema.updates += 1                 # the host mirror of the device counter the recorded optimizer launch advances
        if len(gs) == 1:
            gs[0].replay()
        elif len(gs) == 2:
            gs[0].replay()
            for b in range(self.exchange.nbuckets):
                self.exchange.reduce(b)
            gs[1].replay()
        else:
            gs[0].replay()
            self._reduce_upper_async()
            gs[1].replay()
            self._reduce_rest()
            gs[2].replay()
        return self.loss

    def uncapture(self):
        self.graphs, self.launch = None, 'eager launches'

    def __call__(self):
        return self.replay() if self.graphs is not None else self.eager()


# the EAS-SNN models split for the overlapped exchange: the outputs of DEFAULT_CUT (the CSPDarknet inside the PAFPN) separate "head +
# neck" (bucket 0, finished first by the backward pass) from the parameters of DEFAULT_LOWER (backbone + sampler, bucket 1)
DEFAULT_CUT = ('backbone.backbone',)
DEFAULT_LOWER = ('backbone.backbone', 'embedding')


class Trainer:
    def __init__(self, exp, args):
        self.exp, self.args = exp, args
        self.max_epoch = exp.max_epoch
        self.amp_training = getattr(args, 'fp16', False)
        if self.amp_training:
            raise NotImplementedError('the HIP hot path computes in fp32 (reference parity); --fp16 is not provided')
        self.is_distributed = get_world_size() > 1
        self.rank, self.local_rank = get_rank(), get_local_rank()
        self.device = 'cuda:{}'.format(self.local_rank) if torch.cuda.is_available() else 'cpu'
        self.use_model_ema = exp.ema
        self.input_size = exp.input_size
        self.start_epoch = 0
        self.file_name = os.path.join(exp.output_dir, getattr(args, 'experiment_name', None) or exp.exp_name)
        self.log = []
        self.best_ap = 0
        self.eval_log = []
        self.save_history_ckpt = getattr(exp, 'save_history_ckpt', False)
        self.step = None
        self.exchange = None
        if self.rank == 0:
            os.makedirs(self.file_name, exist_ok=True)
        setup_logger(self.file_name, distributed_rank=self.rank, filename='train_log.txt', mode='a')

    # ---- model, optimizer, gradient exchange (shared by train() and bench.py)
    def setup(self, dp=None, force_exchange=False):
        """model -> device, optimizer, data-parallel exchange.  dp: 'buckets' (default; BucketedGradAllReduce), 'flat' (one bucket) or
        'ddp' (DistributedDataParallel, the reference's choice)."""
        if self.device != 'cpu':
            torch.cuda.set_device(self.local_rank)
        model = self.exp.get_model()
        self.model_info = get_model_info(model, self.exp.test_size)
        model.to(self.device)
        self.optimizer = self.exp.get_optimizer(self.args.batch_size)
        model = self.resume_train(model)
        self.bare_model = model
        self.dp = dp or os.environ.get('EAS_DP', 'buckets')
        self.net = model
        if self.is_distributed or force_exchange:
            if self.dp == 'ddp':
                # per-rank BN running stats until all_reduce_norm, like the reference (trainer.py:175-176)
                self.net = DDP(model, device_ids=[self.local_rank] if self.device != 'cpu' else None, broadcast_buffers=False,
                               gradient_as_bucket_view=True)
            else:
                # (binds to the process group -- and broadcasts rank 0's parameters -- now if the group exists, else at join_ranks())
                self.exchange = BucketedGradAllReduce(model, split=DEFAULT_LOWER if self.dp == 'buckets' else (), world=get_world_size())
        self.model = self.net
        return model

    def step_fn(self, inputs_fn, reset=None, ema=None):
        """the training iteration of this trainer as a ``TrainStep`` (what ``train_one_iter`` runs and ``bench.py`` measures)"""
        if reset is None:
            reset = self.exp.use_spike not in (False, 'False')
        cut = DEFAULT_CUT if (self.exchange is not None and self.exchange.nbuckets > 1) else ()
        return TrainStep(self.bare_model, self.optimizer, inputs_fn, exchange=self.exchange, net=self.net, reset=reset,
                         defer_wgrad=os.environ.get('EAS_DEFER_WGRAD_REDUCE', '1') == '1', cut=cut, ema=ema)

    def make_ema(self, decay=0.9998, updates=0):
        """the weight average of the loop (trainer.py:169-171 of the reference); on the GPU its update becomes part of the optimizer's launch"""
        ema = ModelEMA(self.bare_model, decay)
        ema.updates = updates
        if (self.device != 'cpu' and hasattr(self.optimizer, 'attach_ema') and self.optimizer.takes_ema()
                and os.environ.get('EAS_FUSED_EMA', '1') == '1'):
            self.optimizer.attach_ema(ema, self.bare_model)
        return ema

    def join_ranks(self):
        """create the (deferred) process group and take rank 0's parameter values -- after the graphs are recorded, before the first
        training step.  A no-op on one rank or when the launcher created the group itself."""
        ensure_process_group()
        if self.exchange is not None and not self.exchange.bound and dist.is_initialized():
            self.exchange.bind()
            if self.use_model_ema and getattr(self, 'ema_model', None) is not None:
                self.ema_model.ema.load_state_dict(self.bare_model.state_dict())     # the average starts from the broadcast values (in place)

    def before_train(self):
        self.setup()
        self.train_loader = self.exp.get_data_loader(batch_size=self.args.batch_size, is_distributed=self.is_distributed,
                                                     no_aug=True, cache_img=getattr(self.args, 'cache', None))
        self.max_iter = len(self.train_loader)
        self.lr_scheduler = self.exp.get_lr_scheduler(self.exp.basic_lr_per_img * self.args.batch_size, self.max_iter)
        self.ema_model = self.make_ema(0.9998, self.max_iter * self.start_epoch) if self.use_model_ema else None
        # launch form of the iterations: HIP-graph replay on the GPU unless switched off (EAS_TRAIN_GRAPH=0) or DistributedDataParallel
        # (or a weight average the optimizer's launch does not make: its decay would be a host number frozen into the graph)
        self.use_graph = (self.device != 'cpu' and self.net is self.bare_model and os.environ.get('EAS_TRAIN_GRAPH', '1') == '1'
                          and optimizer_capturable(self.optimizer)
                          and (self.ema_model is None or getattr(self.ema_model, '_fused_in', None) is self.optimizer))
        self._static = None
        self._iters_done = 0
        # evaluation between epochs (trainer.py:178-180, 243-248 of the reference); an experiment without an evaluator trains only
        self.evaluator = None
        if getattr(self.exp, 'eval_interval', 0) and hasattr(self.exp, 'get_evaluator'):
            self.evaluator = self.exp.get_evaluator(batch_size=self.args.batch_size, is_distributed=self.is_distributed)

    def resume_train(self, model):
        ckpt_file = getattr(self.args, 'ckpt', None)
        if getattr(self.args, 'resume', False):
            ckpt_file = ckpt_file or os.path.join(self.file_name, 'latest_ckpt.pth')
            ckpt = torch.load(ckpt_file, map_location=self.device)
            model.load_state_dict(ckpt['model'])
            self.optimizer.load_state_dict(ckpt['optimizer'])
            self.best_ap = ckpt.get('best_ap', 0) or 0          # trainer.py:331 of the reference: a resumed run keeps its best checkpoint
            self.start_epoch = ckpt['start_epoch']
        elif ckpt_file is not None:
            model = load_ckpt(model, torch.load(ckpt_file, map_location=self.device)['model'])
        return model

    def train(self):
        self.before_train()
        # every iteration ends with reset_net, so the final membrane potentials never need to reach HBM (scoped: restored on exit)
        with ops.no_state_writeback() if self.exp.use_spike not in (False, 'False') else contextlib.nullcontext():
            if self.use_graph:
                # graph capture wants every node of the iteration (the AccumulateGrad nodes included) on a non-default stream from the
                # first eager iteration on: the whole loop runs on a stream of its own
                s = _lib.private_stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._train_epochs()
                torch.cuda.current_stream().wait_stream(s)
            else:
                self._train_epochs()

    def _train_epochs(self):
        for self.epoch in range(self.start_epoch, self.max_epoch):
            self.bare_model.head.use_l1 = True          # no_aug from epoch 0 (trainer.py:157, 231-238 of the reference)
            for self.iter, (inps, targets) in enumerate(self.train_loader):
                self.train_one_iter(inps, targets)
            self.after_epoch()

    def after_epoch(self):
        """trainer.py:243-248 of the reference: checkpoint, then every ``eval_interval`` epochs average the BatchNorm statistics over the
        ranks (they train with per-rank statistics) and evaluate"""
        self.save_ckpt('latest')
        if self.evaluator is not None and (self.epoch + 1) % self.exp.eval_interval == 0:
            all_reduce_norm(self.bare_model)
            self.evaluate_and_save_model()

    def evaluate_and_save_model(self):
        """trainer.py:354-386 of the reference: the EMA weights (if any) are what is evaluated and saved"""
        evalmodel = self.ema_model.ema if self.use_model_ema else self.bare_model
        with adjust_status(evalmodel, training=False):
            (ap50_95, ap50, summary), predictions = self.exp.eval(evalmodel, self.evaluator, self.is_distributed, return_outputs=True)
        update_best = ap50_95 is not None and ap50_95 > self.best_ap
        if ap50_95 is not None:
            self.best_ap = max(self.best_ap, ap50_95)
        self.eval_log.append(dict(epoch=self.epoch, ap50_95=ap50_95, ap50=ap50, summary=summary, images=len(predictions)))
        synchronize()
        self.save_ckpt('last_epoch', update_best, ap=ap50_95)
        if self.save_history_ckpt:
            self.save_ckpt(f'epoch_{self.epoch + 1}', ap=ap50_95)

    def _set_lr(self, lr):
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g['lr']):
                g['lr'].fill_(lr)                        # device scalar: read by the (captured) optimizer kernels
            else:
                g['lr'] = lr

    def train_one_iter(self, inps, targets):
        t0 = time.time()
        inps, targets = inps.to(self.device, torch.float32), targets.to(self.device, torch.float32)
        if self._static is None:
            # static input buffers: every batch is copied into them, so an iteration captured once can be replayed on new data
            self._static = (inps.clone(), targets.clone())
            self.step = self.step_fn(lambda: self.exp.preprocess(self._static[0], self._static[1], self.input_size), ema=self.ema_model)
            if self.use_graph:
                for g in self.optimizer.param_groups:    # the schedule writes a device scalar the captured Adam reads
                    g['lr'] = torch.tensor(float(g['lr']), dtype=torch.float32, device=self.device)
                # record on this first batch, leaving no trace (restore): the warm-up launches initialise allocator and optimizer state,
                # the state is put back, and only then does the process group come into being (join_ranks)
                self.step.capture(warm=2, restore=True)
            self.join_ranks()
        else:
            self._static[0].copy_(inps)
            self._static[1].copy_(targets)
        loss = self.step()                               # forward, backward, exchange, optimizer (+ weight average), reset_net
        self._iters_done += 1
        lr = self.lr_scheduler.update_lr(self.epoch * self.max_iter + self.iter + 1)
        self._set_lr(lr)
        if (self.iter + 1) % self.exp.print_interval == 0:
            self.log.append(dict(epoch=self.epoch, iter=self.iter, loss=float(loss.detach()), lr=lr, iter_time=time.time() - t0))

    def save_ckpt(self, ckpt_name, update_best_ckpt=False, ap=None):
        if self.device != 'cpu':
            ops.check_tags('training up to this checkpoint')     # a mis-tagged spike tensor anywhere since the last check: no checkpoint
        if self.rank != 0:
            return
        save_model = self.ema_model.ema if self.use_model_ema else self.bare_model
        opt_state = self.optimizer.state_dict()
        for g in opt_state['param_groups']:              # checkpoints keep plain numbers
            if torch.is_tensor(g.get('lr')):
                g['lr'] = float(g['lr'])
        # the reference's checkpoint format (trainer.py:393-400): best_ap / curr_ap ride along so that a resumed run does not overwrite
        # best_ckpt.pth with a worse model at its first evaluation
        state = {'start_epoch': self.epoch + 1, 'model': save_model.state_dict(), 'optimizer': opt_state, 'best_ap': self.best_ap, 'curr_ap': ap}
        save_checkpoint(state, update_best_ckpt, self.file_name, ckpt_name)
