"""Rank / world helpers (reference: yolox/utils/dist.py).  One process per GPU; ``get_num_devices`` asks
PyTorch-ROCm instead of shelling out to nvidia-smi (dist.py:41-48).

Deferred process group (MI355X): ``yolox.core.launch`` may hand the rendezvous parameters to ``defer_process_group`` instead of calling
``init_process_group`` itself.  Rank and world size are then known from those parameters, the RCCL communicator (and ProcessGroupNCCL's
watchdog thread, which polls events of collectives in flight) only comes into being at ``ensure_process_group()`` -- which the Trainer
calls AFTER it has recorded its HIP graphs, so no capture is ever open while a watchdog exists.  Anything that reaches for the default
group earlier (a ``DistributedDataParallel`` constructor, ``dist.barrier()``) triggers the deferred initialisation on the spot.
``wait_process_group_idle`` is the condition wait for captures that have to happen with a live group (the evaluator under the eval tool's
DistributedDataParallel wrapper): it returns once the watchdog has retired every collective, read from the flight recorder."""
import os
import pickle
import time
from contextlib import contextmanager

import torch
from torch import distributed as dist

_LOCAL_PROCESS_GROUP = None
_PENDING = None             # parameters of a deferred init_process_group (defer_process_group)
_HOOKED = False


def defer_process_group(backend, init_method, world_size, rank, timeout, local_size=None, local_rank=None, **init_kwargs):
    """remember the rendezvous; ``get_rank`` / ``get_world_size`` / ``get_local_rank`` answer from it until ``ensure_process_group``.
    init_kwargs: further ``init_process_group`` arguments (``device_id``)."""
    global _PENDING, _HOOKED
    _PENDING = dict(backend=backend, init_method=init_method, world_size=int(world_size), rank=int(rank), timeout=timeout,
                    local_size=int(local_size or world_size), local_rank=int(rank if local_rank is None else local_rank), extra=init_kwargs)
    if not _HOOKED:
        # every torch.distributed call without an explicit group (and DistributedDataParallel's constructor) asks for the default group
        # through this accessor: a deferred initialisation happens there instead of "Default process group has not been initialized".
        # (A private accessor of torch: if this torch does not have it, nothing is deferred -- the group is created right here, the
        # reference's order, and captures with a live group take the condition wait below.)
        import sys
        try:
            from torch.distributed import distributed_c10d as c10d
            inner = c10d._get_default_group
        except (ImportError, AttributeError):
            ensure_process_group()
            return

        def _get_default_group():
            if _PENDING is not None:
                ensure_process_group()
            return inner()
        for mod in list(sys.modules.values()):        # modules that imported the accessor by name (torch.nn.parallel.distributed) hold their own reference
            try:
                if getattr(mod, '_get_default_group', None) is inner:
                    mod._get_default_group = _get_default_group
            except Exception:
                pass
        _HOOKED = True


def process_group_deferred():
    return _PENDING is not None


def ensure_process_group():
    """run the deferred ``init_process_group`` (no-op when nothing is pending): default group, the per-machine local group of the
    reference's launcher (yolox/core/launch.py:134-143), one barrier"""
    global _PENDING, _LOCAL_PROCESS_GROUP
    if _PENDING is None:
        return False
    p, _PENDING = _PENDING, None
    if p['backend'] == 'nccl':
        # the flight recorder is what wait_process_group_idle reads; it has to be on when the group is created
        os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')
    dist.init_process_group(backend=p['backend'], init_method=p['init_method'], world_size=p['world_size'], rank=p['rank'], timeout=p['timeout'],
                            **p['extra'])
    assert _LOCAL_PROCESS_GROUP is None
    machines, mine = p['world_size'] // p['local_size'], p['rank'] // p['local_size']
    for i in range(machines):
        pg = dist.new_group(list(range(i * p['local_size'], (i + 1) * p['local_size'])))
        if i == mine:
            _LOCAL_PROCESS_GROUP = pg
    synchronize()
    return True


def wait_process_group_idle(limit_s=30.0):
    """Block until ProcessGroupNCCL's watchdog has retired every collective this process issued -- i.e. until it holds no event it could
    query while a HIP-graph capture is open (it aborts the process with hipErrorCapturedEvent when that happens).  A condition, not a
    delay: the flight recorder lists the collectives the watchdog has not retired yet.  True: idle (or no RCCL group at all); False: the
    recorder is not available (then the caller must not capture)."""
    if not _ready() or dist.get_backend() != 'nccl':
        return True
    try:
        # the recorder only lists collectives when it was switched on BEFORE the group was created (ensure_process_group does that); a group
        # somebody else created without it would look idle at all times
        if int(os.environ.get('TORCH_NCCL_TRACE_BUFFER_SIZE', '0')) <= 0:
            return False
        from torch._C._distributed_c10d import _dump_nccl_trace
    except (ImportError, ValueError):
        return False
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.time()
    while True:
        try:
            trace = pickle.loads(_dump_nccl_trace(True, False, True))       # collectives, no stack traces, only those not retired
        except Exception:
            return False
        entries = trace.get('entries') if isinstance(trace, dict) else None
        if entries is None:
            return False
        if not entries:
            return True
        if time.time() - t0 > limit_s:
            return False
        time.sleep(0.005)


def get_num_devices():
    visible = os.getenv('CUDA_VISIBLE_DEVICES', None) or os.getenv('HIP_VISIBLE_DEVICES', None)
    if visible is not None:
        return len([d for d in visible.split(',') if d != ''])
    return torch.cuda.device_count()


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    if _PENDING is not None:
        return _PENDING['world_size']
    return dist.get_world_size() if _ready() else 1


def get_rank() -> int:
    if _PENDING is not None:
        return _PENDING['rank']
    return dist.get_rank() if _ready() else 0


def get_local_rank() -> int:
    if _PENDING is not None:
        return _PENDING['local_rank']
    if _LOCAL_PROCESS_GROUP is None:
        return get_rank()
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP) if _ready() else 0


def get_local_size() -> int:
    if _PENDING is not None:
        return _PENDING['local_size']
    return dist.get_world_size(group=_LOCAL_PROCESS_GROUP) if _ready() else 1


def is_main_process() -> bool:
    return get_rank() == 0


_GLOO_GROUP = None


def _object_group():
    """process group for pickled python objects: the default group on gloo, a gloo twin of it on RCCL (objects live on the host;
    reference: dist.py:142-152 ``_get_global_gloo_group``)"""
    global _GLOO_GROUP
    if dist.get_backend() != 'nccl':
        return dist.group.WORLD
    if _GLOO_GROUP is None:
        _GLOO_GROUP = dist.new_group(backend='gloo')
    return _GLOO_GROUP


def gather(data, dst=0, group=None):
    """picklable ``data`` of every rank as a list on rank ``dst`` (elsewhere: []); reference: dist.py:233-273"""
    if get_world_size() == 1:
        return [data]
    ensure_process_group()
    group = _object_group() if group is None else group
    out = [None] * dist.get_world_size(group) if dist.get_rank(group) == dst else None
    dist.gather_object(data, out, dst=dst, group=group)
    return out if out is not None else []


def all_gather(data, group=None):
    """picklable ``data`` of every rank as a list on every rank; reference: dist.py:195-230"""
    if get_world_size() == 1:
        return [data]
    ensure_process_group()
    group = _object_group() if group is None else group
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, data, group=group)
    return out


def synchronize():
    ensure_process_group()
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()


@contextmanager
def wait_for_the_master(local_rank: int = None):
    if local_rank is None:
        local_rank = get_local_rank()
    ensure_process_group()
    if local_rank > 0:
        dist.barrier()
    yield
    if local_rank == 0 and _ready() and dist.get_world_size() > 1:
        dist.barrier()


def time_synchronized():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()
