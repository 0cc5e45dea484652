"""Logging setup (reference: yolox/utils/logger.py).  loguru is optional here."""
import os
import sys


def setup_logger(save_dir, distributed_rank=0, filename='log.txt', mode='a'):
    try:
        from loguru import logger
    except ImportError:
        return
    logger.remove()
    if distributed_rank == 0:
        logger.add(sys.stderr, level='INFO', enqueue=True)
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, filename)
        if mode == 'o' and os.path.exists(path):
            os.remove(path)
        logger.add(path)
