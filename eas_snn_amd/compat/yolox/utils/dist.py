"""Rank / world helpers (reference: yolox/utils/dist.py).  One process per GPU; ``get_num_devices`` asks
PyTorch-ROCm instead of shelling out to nvidia-smi (dist.py:41-48)."""
import os
import time
from contextlib import contextmanager

import torch
from torch import distributed as dist

_LOCAL_PROCESS_GROUP = None


def get_num_devices():
    visible = os.getenv('CUDA_VISIBLE_DEVICES', None) or os.getenv('HIP_VISIBLE_DEVICES', None)
    if visible is not None:
        return len([d for d in visible.split(',') if d != ''])
    return torch.cuda.device_count()


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if _ready() else 1


def get_rank() -> int:
    return dist.get_rank() if _ready() else 0


def get_local_rank() -> int:
    if _LOCAL_PROCESS_GROUP is None:
        return get_rank()
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP) if _ready() else 0


def get_local_size() -> int:
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
    group = _object_group() if group is None else group
    out = [None] * dist.get_world_size(group) if dist.get_rank(group) == dst else None
    dist.gather_object(data, out, dst=dst, group=group)
    return out if out is not None else []


def all_gather(data, group=None):
    """picklable ``data`` of every rank as a list on every rank; reference: dist.py:195-230"""
    if get_world_size() == 1:
        return [data]
    group = _object_group() if group is None else group
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, data, group=group)
    return out


def synchronize():
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()


@contextmanager
def wait_for_the_master(local_rank: int = None):
    if local_rank is None:
        local_rank = get_local_rank()
    if local_rank > 0:
        dist.barrier()
    yield
    if local_rank == 0 and _ready() and dist.get_world_size() > 1:
        dist.barrier()


def time_synchronized():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()
