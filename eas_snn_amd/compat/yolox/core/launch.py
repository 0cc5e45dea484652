"""Process launch: one process per GPU, ``torch.distributed`` over RCCL/xGMI (reference: yolox/core/launch.py:39-147).

Same signature and behaviour as the reference's ``launch``.  Differences that matter on MI355X:
``backend='nccl'`` is RCCL on PyTorch-ROCm; HSA_ENABLE_IPC_MODE_LEGACY=0 is exported to the workers (dmabuf IPC);
the rendezvous address stays 127.0.0.1 (single node).

With the RCCL backend a worker does not create its process group before ``main_func`` runs: it hands the rendezvous parameters to
``yolox.utils.dist.defer_process_group`` and the group comes into being at the first use -- which for the Trainer is right after it has
recorded its HIP graphs (a capture must never be open while ProcessGroupNCCL's watchdog thread polls events).  Rank
and world size are known from the parameters all along.  ``EAS_LAZY_PG=0`` restores the reference's order (group first); ``all`` defers on
every backend (the CPU tests walk the deferred order on gloo)."""
import os
import sys
from datetime import timedelta

import torch
import torch.multiprocessing as mp

import yolox.utils.dist as comm

__all__ = ['launch']

DEFAULT_TIMEOUT = timedelta(minutes=30)


def _find_free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('', 0))
        return s.getsockname()[1]


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, backend='nccl', dist_url=None, args=(),
           timeout=DEFAULT_TIMEOUT):
    world_size = num_machines * num_gpus_per_machine
    if world_size <= 1:
        return main_func(*args)
    if dist_url == 'auto':
        assert num_machines == 1, 'dist_url=auto cannot work with distributed training.'
        dist_url = f'tcp://127.0.0.1:{_find_free_port()}'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cache = vars(args[1]).get('cache', False) if len(args) > 1 and hasattr(args[1], '__dict__') else False
    start_method = 'fork' if cache else 'spawn'
    if cache:
        assert sys.platform != 'win32'
    mp.start_processes(_distributed_worker, nprocs=num_gpus_per_machine,
                       args=(main_func, world_size, num_gpus_per_machine, machine_rank, backend, dist_url, args),
                       daemon=False, start_method=start_method)


def _distributed_worker(local_rank, main_func, world_size, num_gpus_per_machine, machine_rank, backend, dist_url, args,
                        timeout=DEFAULT_TIMEOUT):
    if backend == 'nccl':
        assert torch.cuda.is_available(), 'no GPU visible: the nccl (RCCL) backend needs one GPU per process'
        assert num_gpus_per_machine <= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    global_rank = machine_rank * num_gpus_per_machine + local_rank
    comm.defer_process_group(backend, dist_url, world_size, global_rank, timeout, local_size=num_gpus_per_machine, local_rank=local_rank)
    lazy = os.environ.get('EAS_LAZY_PG', '1')
    if not (lazy == 'all' or (lazy == '1' and backend == 'nccl')):
        comm.ensure_process_group()         # default group, the machine-local group, one barrier (launch.py:118-147 of the reference)
    main_func(*args)
