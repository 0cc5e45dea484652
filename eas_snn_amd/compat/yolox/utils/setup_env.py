"""Process environment (reference: yolox/utils/setup_env.py).  On MI355X the 'nccl' backend is RCCL over xGMI;
the InfiniBand variables of the reference are meaningless on a single node and are not set."""
import os

__all__ = ['configure_nccl', 'configure_module', 'configure_omp']


def configure_nccl():
    os.environ.setdefault('NCCL_LAUNCH_MODE', 'PARALLEL')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC, required by RCCL on this platform


def configure_omp(num_threads=1):
    os.environ.setdefault('OMP_NUM_THREADS', str(num_threads))


def configure_module(ulimit_value=8192):
    try:
        import resource
        soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        resource.setrlimit(resource.RLIMIT_NOFILE, (min(ulimit_value, hard), hard))
    except Exception:
        pass
    os.environ['OPENCV_OPENCL_RUNTIME'] = 'disabled'
