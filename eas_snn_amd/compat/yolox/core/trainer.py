"""Training loop (reference: yolox/core/trainer.py:36-419), reduced to what the hot path needs:
model -> device, optimizer, DDP(broadcast_buffers=False) over RCCL, EMA, per-iteration
forward / backward / step / reset_net, LR schedule, 'latest' checkpoint.  Evaluation, TensorBoard/W&B
logging and dataset prefetching are out of scope (synthetic loader yields GPU tensors)."""
import contextlib
import os
import time

import torch
from torch.nn.parallel import DistributedDataParallel as DDP

from eas_snn_amd import ops
from yolox.utils import (ModelEMA, get_local_rank, get_model_info, get_rank, get_world_size, is_parallel, load_ckpt,
                         save_checkpoint, setup_logger)


class Trainer:
    def __init__(self, exp, args):
        self.exp, self.args = exp, args
        self.max_epoch = exp.max_epoch
        self.amp_training = getattr(args, 'fp16', False)
        if self.amp_training:
            raise NotImplementedError('the HIP hot path computes in fp32 (reference parity); --fp16 is not provided')
        self.is_distributed = get_world_size() > 1
        self.rank, self.local_rank = get_rank(), get_local_rank()
        self.device = 'cuda:{}'.format(self.local_rank)
        self.use_model_ema = exp.ema
        self.input_size = exp.input_size
        self.start_epoch = 0
        self.file_name = os.path.join(exp.output_dir, getattr(args, 'experiment_name', None) or exp.exp_name)
        self.log = []
        if self.rank == 0:
            os.makedirs(self.file_name, exist_ok=True)
        setup_logger(self.file_name, distributed_rank=self.rank, filename='train_log.txt', mode='a')

    def before_train(self):
        torch.cuda.set_device(self.local_rank)
        model = self.exp.get_model()
        self.model_info = get_model_info(model, self.exp.test_size)
        model.to(self.device)
        self.optimizer = self.exp.get_optimizer(self.args.batch_size)
        model = self.resume_train(model)
        self.train_loader = self.exp.get_data_loader(batch_size=self.args.batch_size, is_distributed=self.is_distributed,
                                                     no_aug=True, cache_img=getattr(self.args, 'cache', None))
        self.max_iter = len(self.train_loader)
        self.lr_scheduler = self.exp.get_lr_scheduler(self.exp.basic_lr_per_img * self.args.batch_size, self.max_iter)
        if self.is_distributed:
            # per-rank BN running stats until all_reduce_norm, like the reference (trainer.py:175-176)
            model = DDP(model, device_ids=[self.local_rank], broadcast_buffers=False)
        if self.use_model_ema:
            self.ema_model = ModelEMA(model, 0.9998)
            self.ema_model.updates = self.max_iter * self.start_epoch
        self.model = model

    def resume_train(self, model):
        ckpt_file = getattr(self.args, 'ckpt', None)
        if getattr(self.args, 'resume', False):
            ckpt_file = ckpt_file or os.path.join(self.file_name, 'latest_ckpt.pth')
            ckpt = torch.load(ckpt_file, map_location=self.device)
            model.load_state_dict(ckpt['model'])
            self.optimizer.load_state_dict(ckpt['optimizer'])
            self.start_epoch = ckpt['start_epoch']
        elif ckpt_file is not None:
            model = load_ckpt(model, torch.load(ckpt_file, map_location=self.device)['model'])
        return model

    def train(self):
        self.before_train()
        # every iteration ends with reset_net, so the final membrane potentials never need to reach HBM (scoped: restored on exit)
        with ops.no_state_writeback() if self.exp.use_spike not in (False, 'False') else contextlib.nullcontext():
            self._train_epochs()

    def _train_epochs(self):
        for self.epoch in range(self.start_epoch, self.max_epoch):
            (self.model.module if is_parallel(self.model) else self.model).head.use_l1 = True   # no_aug from epoch 0
            for self.iter, (inps, targets) in enumerate(self.train_loader):
                self.train_one_iter(inps, targets)
            self.save_ckpt('latest')

    def train_one_iter(self, inps, targets):
        from spikingjelly.activation_based import functional
        t0 = time.time()
        inps, targets = inps.to(self.device, torch.float32), targets.to(self.device, torch.float32)
        inps, targets = self.exp.preprocess(inps, targets, self.input_size)
        outputs = self.model(inps, targets)
        loss = outputs['total_loss']
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        if self.exp.use_spike not in (False, 'False'):
            functional.reset_net(self.model)
        if self.use_model_ema:
            self.ema_model.update(self.model)
        lr = self.lr_scheduler.update_lr(self.epoch * self.max_iter + self.iter + 1)
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        if (self.iter + 1) % self.exp.print_interval == 0:
            self.log.append(dict(epoch=self.epoch, iter=self.iter, loss=float(loss), lr=lr, iter_time=time.time() - t0))

    def save_ckpt(self, ckpt_name, update_best_ckpt=False):
        if self.rank != 0:
            return
        save_model = self.ema_model.ema if self.use_model_ema else self.model
        state = {'start_epoch': self.epoch + 1, 'model': save_model.state_dict(), 'optimizer': self.optimizer.state_dict()}
        save_checkpoint(state, update_best_ckpt, self.file_name, ckpt_name)
