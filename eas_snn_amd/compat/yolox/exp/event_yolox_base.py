"""Experiment description for event-camera detection (reference: yolox/exp/event_yolox_base.py:18-553).

Option names, defaults and their Python types are the reference's (``merge`` coerces overrides to the type
of the default, so they are part of the command-line contract).  ``get_model`` assembles the sampler, the
(spiking) backbone/neck and the head from eas_snn_amd's HIP-backed modules; datasets and evaluators are out
of scope (real datasets are absent, the benchmark feeds synthetic event streams)."""
import os

import torch
import torch.nn as nn

from .base_exp import BaseExp

__all__ = ['EventExp', 'check_exp_value', 'sized_exp']

# name -> default.  Grouped as in the reference: model, SNN/sampler, data, training, testing.
_DEFAULTS = dict(
    num_classes=100, depth=1.00, width=1.00, act='silu', use_spike='False', eval_proph=False, alpha=2.0, in_dim=2,
    aggregation='micro_sum',
    emb_lr=-1.0, embedding='count', embedding_depth=1, spike_attach=False, write_zero=False, abs=False, split=False,
    embedding_ksize=7, norm=None, window=-200, Tl=1, Tm=4, Ts=1, T=4, reset=0, thresh=1, readout='sum', decay=0.5,
    speed_aug=False, spike_fn='rect', data_name='n-caltech',
    data_num_workers=4, measure='count', input_size=(640, 640), multiscale_range=5, data_dir='/data2/wzm/dataset/N-Caltech',
    flip_prob=0.5,
    warmup_epochs=0, max_epoch=300, warmup_lr=0, min_lr_ratio=0.05, basic_lr_per_img=1e-3 / 64.0, scheduler='yoloxwarmcos',
    no_aug_epochs=0, ema=True, optimizer='ADAM', weight_decay=0, momentum=0.9, print_interval=10, eval_interval=10,
    save_history_ckpt=False,
    test_size=(640, 640), test_conf=0.01, nmsthre=0.65,
)


class EventExp(BaseExp):
    def __init__(self):
        super().__init__()
        for k, v in _DEFAULTS.items():
            setattr(self, k, v)
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split('.')[0]

    # ------------------------------------------------------------------ model
    def get_act_func(self):
        from spikingjelly.activation_based import surrogate
        from yolox.models.activation import Rectangle
        if self.spike_fn == 'rect':
            return Rectangle
        if self.spike_fn == 'atan':
            return surrogate.ATan(self.alpha)
        if self.spike_fn == 'sigmoid':
            return surrogate.Sigmoid(self.alpha)
        if self.spike_fn == 'patan':
            from yolox.models.activation import EfficientNoisySpikeII, InvArcTanh
            return EfficientNoisySpikeII(InvArcTanh(self.alpha), p=0)
        raise KeyError(self.spike_fn)            # the reference indexes a dict of these four names

    def get_kwargs_spikes(self):
        from yolox.models.activation import Rectangle
        from yolox.utils.util import warp_decay
        # the sampler's spike function is fixed to Rectangle in the reference (event_yolox_base.py:156)
        return {'nb_steps': self.Tm, 'vreset': self.reset, 'thresh': self.thresh, 'spike_fn': Rectangle,
                'decay': nn.Parameter(warp_decay(self.decay)), 'embedding': self.embedding, 'Ts': self.Ts,
                'spike_attach': self.spike_attach}

    def _build_embedding(self):
        from yolox.models import embedding as E
        kw = self.get_kwargs_spikes()
        if self.embedding == 'arsnn':
            return E.AdaptiveRSNNEmbedding(kernel_size=self.embedding_ksize, in_channel=2, out_channel=2, readout=self.readout,
                                           split=self.split, write_zero=self.write_zero, abs=self.abs,
                                           depth=self.embedding_depth, **kw)
        if self.embedding == 'count':
            return E.SpikeCountEmbedding(kw['nb_steps'])
        if self.embedding == 'rsnn':
            return E.SpikingEmbedding(kernel_size=self.embedding_ksize, in_channel=2, out_channel=2, readout=self.readout,
                                      relu=self.abs, depth=self.embedding_depth, **kw)
        if self.embedding == 'snn':
            return E.LIFEmbedding(kernel_size=self.embedding_ksize, in_channel=2, out_channel=2, readout=self.readout,
                                  depth=self.embedding_depth, **kw)
        raise KeyError(self.embedding)

    def get_model(self):
        from yolox.models import YOLOX, YOLOPAFPN, YOLOXHead, SpikingYOLOX, SpikingYOLOXHead, SpikingYOLOPAFPN
        from yolox.utils.utils_snn import convert_to_spiking
        if getattr(self, 'model', None) is None:
            embedding = self._build_embedding()
            if self.norm is not None:
                embedding = nn.ModuleList([embedding, nn.BatchNorm2d(2)])
            chans = [256, 512, 1024]
            if self.use_spike == 'True' or self.use_spike is True:              # spiking backbone, ANN neck + head
                backbone = SpikingYOLOPAFPN(self.depth, self.width, in_channels=chans, in_dim=self.in_dim, act=self.act,
                                            spike_fn=self.get_act_func())
                head = YOLOXHead(self.num_classes, self.width, in_channels=chans, act=self.act)
                self.model = SpikingYOLOX(backbone, head, embedding, T=self.T)
            elif isinstance(self.use_spike, str) and 'full_spike' in self.use_spike:   # neck spiking too; 'v2': head too
                backbone = convert_to_spiking(YOLOPAFPN(self.depth, self.width, in_channels=chans, in_dim=2, act=self.act),
                                              spike_fn=self.get_act_func())
                head = SpikingYOLOXHead(self.num_classes, self.width, in_channels=chans, act=self.act,
                                        spike_fn=self.get_act_func(), full_spike=('v2' in self.use_spike))
                self.model = SpikingYOLOX(backbone, head, embedding, T=self.T)
            elif self.use_spike is False or self.use_spike == 'False':          # ANN behind the sampler
                backbone = YOLOPAFPN(self.depth, self.width, in_channels=chans, in_dim=2, act=self.act)
                head = YOLOXHead(self.num_classes, self.width, in_channels=chans, act=self.act)
                self.model = YOLOX(backbone, head, embedding)
            else:
                raise ValueError(f'use_spike={self.use_spike!r}')
            from yolox.models.network_blocks import enable_spike_planes
            enable_spike_planes(self.model)       # converted blocks hand their spikes on as bf16 planes inside the model's forward
        for m in self.model.modules():                                          # init_yolo (:179-183)
            if isinstance(m, nn.BatchNorm2d):
                m.eps = 1e-3
                m.momentum = 0.03
        self.model.head.initialize_biases(1e-2)
        self.model.train()
        return self.model

    # ------------------------------------------------------------------ optimisation
    def get_optimizer(self, batch_size):
        """Parameter groups of the reference (event_yolox_base.py:352-414): BN weights | conv weights (decay) |
        biases | neuron parameters | sampler parameters (own lr when emb_lr >= 0)."""
        from yolox.utils.utils_snn import is_spiking_neuron
        if isinstance(self.optimizer, torch.optim.Optimizer):
            return self.optimizer
        lr = self.warmup_lr if self.warmup_epochs > 0 else self.basic_lr_per_img * batch_size
        bn_w, conv_w, biases, neuron_p = [], [], [], []
        adam = self.optimizer == 'ADAM'
        for k, v in self.model.named_modules():
            if adam and 'embedding' in k:       # own group below; the SGD branch of the reference (:361-377) keeps the
                continue                        # sampler's convolutions in the weight / bias groups and has no neuron group
            if hasattr(v, 'bias') and isinstance(v.bias, nn.Parameter):
                biases.append(v.bias)
            if isinstance(v, nn.BatchNorm2d) or 'bn' in k:
                bn_w.append(v.weight)
            elif hasattr(v, 'weight') and isinstance(v.weight, nn.Parameter):
                conv_w.append(v.weight)
            if is_spiking_neuron(v):
                neuron_p.extend(p for _, p in v.named_parameters())
        emb_p = [p for _, p in self.model.embedding.named_parameters() if p.requires_grad]
        if adam:
            # same update rule as the reference's torch.optim.Adam; on the GPU the single-kernel-per-group implementation
            # (40 small foreach kernels per step become 5)
            on_gpu = all(p.is_cuda for p in bn_w + conv_w + biases + neuron_p + emb_p)
            if on_gpu:
                from eas_snn_amd.optim import FusedAdam          # torch.optim.Adam subclass: the step of all groups as one launch (EAS_FUSED_ADAM=0: torch's)
                opt = FusedAdam(bn_w, lr=lr, amsgrad=False)
            else:
                opt = torch.optim.Adam(bn_w, lr=lr, amsgrad=False)
        else:
            opt = torch.optim.SGD(bn_w, lr=lr, momentum=self.momentum, nesterov=True)
        opt.add_param_group({'params': conv_w, 'weight_decay': self.weight_decay})
        opt.add_param_group({'params': biases})
        if adam:
            opt.add_param_group({'params': neuron_p})
            opt.add_param_group({'params': emb_p, 'lr': lr if self.emb_lr < 0 else self.emb_lr})
        self.optimizer = opt
        return opt

    def get_lr_scheduler(self, lr, iters_per_epoch):
        from yolox.utils import LRScheduler
        return LRScheduler(self.scheduler, lr, iters_per_epoch, self.max_epoch, warmup_epochs=self.warmup_epochs,
                           warmup_lr_start=self.warmup_lr, no_aug_epochs=self.no_aug_epochs, min_lr_ratio=self.min_lr_ratio)

    def preprocess(self, inputs, targets, tsize):
        assert tuple(tsize) == tuple(self.input_size), 'Only support scale_x or scale_y in Dataset'
        return inputs, targets

    # ------------------------------------------------------------------ data (synthetic only; real datasets are out of scope)
    def get_slice_args(self):
        return {'aggregation': self.aggregation, 'overlap': 0, 'num_slice': self.Tl, 'micro_slice': self.Tm,
                'measure': self.measure, 'window': (self.window * 1000, 0)}

    def get_dataset(self, cache=False, cache_type='ram'):
        from eas_snn_amd.data import SyntheticEventDataset
        return SyntheticEventDataset(self)

    def get_data_loader(self, batch_size, is_distributed, no_aug=False, cache_img=None):
        from eas_snn_amd.data import SyntheticEventLoader
        from yolox.utils import get_world_size      # (answers from the launcher's parameters while the process group is still deferred)
        if is_distributed:
            batch_size = batch_size // get_world_size()
        return SyntheticEventLoader(self, batch_size)

    def get_eval_dataset(self, **kwargs):
        from eas_snn_amd.data import SyntheticEvalDataset
        return SyntheticEvalDataset(self, length=int(getattr(self, 'eval_samples', 256)), n_events=int(getattr(self, 'eval_events', 200_000)))

    def get_eval_loader(self, batch_size, is_distributed, **kwargs):
        """reference: event_yolox_base.py:483-507 -- twice the training batch, split over the ranks, samples rank, rank + world, ...
        in order (DistributedSampler(shuffle=False)); the streams are synthetic and binned on the GPU"""
        from eas_snn_amd.data import SyntheticEvalLoader
        from yolox.utils import get_rank, get_world_size
        valdataset = self.get_eval_dataset(**kwargs)
        batch_size *= 2
        rank, world = 0, 1
        if is_distributed:
            rank, world = get_rank(), get_world_size()
            batch_size = batch_size // world
        n = len(valdataset)
        per_rank = (n + world - 1) // world                 # DistributedSampler pads with the first samples so every rank gets the same count
        idx = [(rank + k * world) % n for k in range(per_rank)]
        # synthetic streams on the Gen1 sensor (or on the canvas itself when that is smaller); ``eval_sensor_hw`` / ``eval_events`` override
        sensor = getattr(self, 'eval_sensor_hw', None) or (min(240, self.test_size[0]), min(304, self.test_size[1]))
        return SyntheticEvalLoader(self, batch_size, idx, n_events=valdataset.n_events, sensor_hw=tuple(sensor), dataset=valdataset)

    def get_evaluator(self, batch_size, is_distributed, testdev=False, legacy=False):
        """reference: event_yolox_base.py:509-534.  ``eval_proph`` (the Prophesee metric toolbox on real Gen1 / 1 Mpx recordings) has no
        counterpart here: its inference loop is the same one (psee_evaluator.py:180-215), its metric code needs the datasets."""
        from yolox.evaluators import EventEvaluator
        if 'gen' in str(self.data_name) and self.eval_proph not in (False, 'False', None, 0):
            # the README's eval line (readme.md:157-160, --eval_proh) selects PSEEEvaluator there: say so instead of substituting silently
            import warnings
            warnings.warn('eval_proph is set: the reference would build PSEEEvaluator (Prophesee metric toolbox, event_yolox_base.py:512-523). '
                          'Its metric code is outside this package (it needs the toolbox and the real Gen1 / 1 Mpx label files); this run uses '
                          'EventEvaluator -- the same inference loop, timers and detection records, COCO-style AP instead of the toolbox\'s.',
                          RuntimeWarning, stacklevel=2)
        return EventEvaluator(dataloader=self.get_eval_loader(batch_size, is_distributed, testdev=testdev, legacy=legacy),
                              img_size=self.test_size, confthre=self.test_conf, nmsthre=self.nmsthre, num_classes=self.num_classes,
                              testdev=testdev, snn_reset=self.use_spike)

    def get_trainer(self, args):
        from yolox.core import Trainer
        return Trainer(self, args)

    def eval(self, model, evaluator, is_distributed, half=False, return_outputs=False):
        return evaluator.evaluate(model, is_distributed, half, return_outputs=return_outputs)


def check_exp_value(exp):
    h, w = exp.input_size
    assert h % 32 == 0 and w % 32 == 0, 'input size must be multiples of 32'


def sized_exp(exp_file, depth, width, max_epoch):
    """Factory for the exps/default/e_yolox_{s,m,l}.py experiment files."""
    name = os.path.split(os.path.realpath(exp_file))[1].split('.')[0]

    class Exp(EventExp):
        def __init__(self):
            super().__init__()
            self.depth, self.width, self.max_epoch, self.exp_name = depth, width, max_epoch, name

    return Exp
