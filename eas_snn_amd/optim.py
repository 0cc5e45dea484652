"""torch.optim.Adam whose step is ONE launch over all parameter groups (eas_adam_step_ex, csrc/adam.hip) instead of torch's eleven multi-tensor
kernels + six step-counter launches per step of SYOLOX-S.  Same class hierarchy, same state (``step`` / ``exp_avg`` / ``exp_avg_sq`` per
parameter: checkpoints of either implementation load into the other), same update rule (torch's fused kernel, with its float / double
promotions).  Reference: the optimizer of yolox/exp/event_yolox_base.py:352-414 (``torch.optim.Adam``, five parameter groups).

The weight average of the training loop rides in the same launch: ``attach_ema(ModelEMA, model)`` gives every table entry the address of
the tensor's twin in the averaged model, BatchNorm running statistics join as entries without a gradient, and the decay ramp is computed on
the device from a device counter -- ``ModelEMA.update`` (yolox/utils/ema.py:44-60, called after every ``optimizer.step()`` in
yolox/core/trainer.py:120-121 of the reference) then costs no launch and survives HIP-graph replay."""
import contextlib
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, stream

ENABLED = os.environ.get('EAS_FUSED_ADAM', '1') != '0'      # 0: torch's fused implementation
_WORDS = 12                                                 # int64 words per table entry (96 bytes, csrc/adam.hip AdamTensor)
_MAX_GROUPS = 16                                            # EAS_ADAM_MAX_GROUPS


class FusedAdam(torch.optim.Adam):
    """``torch.optim.Adam(..., fused=True)`` with the step on the own kernel when every parameter is a dense fp32 CUDA tensor (else torch's).

    Graph capture is opt-in: a capture records the launches against a pointer table that can only be filled after the capture has ended, so
    only a caller that opens ``capture_scope()`` around its captures (``TrainStep.capture`` does) gets the own kernel recorded; any other
    capture of ``step()`` records torch's implementation."""

    def __init__(self, params, **kw):
        kw.setdefault('fused', True)
        super().__init__(params, **kw)
        self._eas_tables = {}
        self._eas_armed = None          # inside capture_scope(): {'spares': [tables allocated before the capture began]}
        self._eas_ema = None
        self._eas_ema_serial = 0

    # ---- weight average (ModelEMA) inside the step
    def takes_ema(self):
        """this optimizer's step runs on the own kernel (so a weight average can ride in its launch): switched on, and every group eligible"""
        return ENABLED and len(self.param_groups) <= _MAX_GROUPS and all(self._eligible(g) for g in self.param_groups)

    def attach_ema(self, ema, model):
        """From now on ``step()`` also takes the update ``ema.update(model)`` would make (same arithmetic, see csrc/adam.hip); ``ema.update``
        itself only counts.  ``ema``: yolox.utils.ModelEMA; ``model``: the module whose state dict it averages."""
        live, avg = model.state_dict(), ema.ema.state_dict()
        twins, order = {}, []
        for name, a in avg.items():
            if not a.is_floating_point():
                continue
            src = live[name]
            if not (a.is_cuda and a.dtype == torch.float32 and src.dtype == torch.float32 and a.is_contiguous() and src.is_contiguous()
                    and a.shape == src.shape):
                raise RuntimeError(f'attach_ema: {name} is not a dense fp32 device tensor in both models')
            twins[src.data_ptr()] = a
            order.append((src, a))
        dev = order[0][0].device
        self._eas_ema = {'obj': ema, 'twins': twins, 'order': order, 'decay': float(ema.nominal_decay), 'ramp': float(ema.ramp),
                         'counter': torch.tensor([float(ema.updates)], dtype=torch.float64, device=dev)}
        self._eas_ema_serial = getattr(self, '_eas_ema_serial', 0) + 1       # part of the table key: tables of other attachments (a step
        ema._fused_in = self                                                 # recorded into a graph keeps replaying with its own) stay alive

    def detach_ema(self):
        if self._eas_ema is not None:
            self._eas_ema['obj']._fused_in = None
            self._eas_ema = None

    def ema_updates_on_device(self):
        """the device counter's value (a host synchronisation: tests / checkpoints)"""
        return None if self._eas_ema is None else int(self._eas_ema['counter'].item())

    def _ema_unfused(self):
        """the average after a step that fell back to torch's implementation: the three tensor operators of the reference, multi-tensor"""
        e = self._eas_ema
        n = e['obj'].updates + 1                   # ema.update(), called by the training step right after this, makes it the current count
        keep = e['obj'].decay(n)
        avg, cur = [a for _, a in e['order']], [s_ for s_, _ in e['order']]
        torch._foreach_mul_(avg, keep)
        torch._foreach_add_(avg, torch._foreach_mul(cur, 1.0 - keep))
        e['counter'].add_(1.0)

    def _torch_step(self, closure=None):
        if self._eas_ema is not None and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('FusedAdam: a step with an attached weight average cannot be captured on torch\'s implementation (the decay would '
                               'be a host number frozen into the graph)')
        out = super().step(closure) if closure is not None else super().step()
        if self._eas_ema is not None:
            self._ema_unfused()
        return out

    # ---- capture
    @contextlib.contextmanager
    def capture_scope(self):
        """Open around the HIP-graph capture(s) of a step that contains ``step()``.  Inside a capture nothing can be copied from the host
        (pinned-memory bookkeeping records events on the capturing stream; a pageable copy would be replayed from a dead staging buffer), so
        a captured ``step()`` records its launches against a table that is filled when the scope closes -- a capture only records, it does
        not run.  The table must be ordinary memory: memory from the graph's private pool did not keep what was written into it after the
        capture (illegal addresses at the first replay), hence the spares allocated here, before any capture is open."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('FusedAdam.capture_scope must be opened before the capture begins')
        n_max = sum(len(g['params']) for g in self.param_groups) + (len(self._eas_ema['order']) if self._eas_ema is not None else 0)
        dev = self.param_groups[0]['params'][0].device
        self._eas_armed = {'spares': [torch.empty(n_max * _WORDS, dtype=torch.int64, device=dev) for _ in range(2)]}
        try:
            yield self
        finally:
            self._eas_armed = None
            if not torch.cuda.is_current_stream_capturing():
                self.sync_tables()

    def sync_tables(self):
        """fill the tables made inside a graph capture; runs when ``capture_scope`` closes, before the first replay"""
        for slot in self._eas_tables.values():
            if slot['pending'] is not None:
                slot['table'][:slot['pending'].size].copy_(torch.from_numpy(slot['pending']))
                slot['pending'] = None

    def _eligible(self, group):
        return (ENABLED and not group['amsgrad'] and not group['maximize'] and not group.get('differentiable', False)
                and not group.get('decoupled_weight_decay', False) and not isinstance(group['betas'][0], torch.Tensor)
                and not isinstance(group['betas'][1], torch.Tensor))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(self, 'grad_scale', None) is not None or getattr(self, 'found_inf', None) is not None:
            return self._torch_step(closure)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and self._eas_armed is None:
            return self._torch_step()       # a capture nobody announced: its table would never be filled (see capture_scope)
        if len(self.param_groups) > _MAX_GROUPS:
            return self._torch_step()
        entries = []
        betas = eps = None
        hyper = _lib.EasAdamHyper()
        for gi, group in enumerate(self.param_groups):
            if not self._eligible(group):
                return self._torch_step()
            if betas is None:
                betas, eps = tuple(float(b) for b in group['betas']), float(group['eps'])
            elif betas != tuple(float(b) for b in group['betas']) or eps != float(group['eps']):
                return self._torch_step()                                    # one (beta1, beta2, eps) per launch
            ps, gs, ms, vs, mx, steps = [], [], [], [], [], []
            if self._init_group(group, ps, gs, ms, vs, mx, steps):            # complex parameters
                return self._torch_step()
            lr = group['lr']
            if torch.is_tensor(lr):
                if not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
                    return self._torch_step()
            else:
                hyper.group_lr[gi] = float(lr)
            for p, g, m, v, st in zip(ps, gs, ms, vs, steps):
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse and p.is_contiguous()
                        and g.is_contiguous() and m.is_contiguous() and v.is_contiguous() and st.is_cuda and st.dtype == torch.float32):
                    return self._torch_step()
                entries.append((p, g, m, v, st, lr, float(group['weight_decay']), gi))
        ema = self._eas_ema
        if not entries and ema is None:
            return None
        # the learning rate is not part of the key: a python float travels as a launch argument (hyper.group_lr), a device scalar by address
        key = (self._eas_ema_serial if ema is not None else 0,) + tuple(
            (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr(), lr.data_ptr() if torch.is_tensor(lr) else -1 - gi,
             wd, p.numel()) for p, g, m, v, st, lr, wd, gi in entries)
        L = _lib.lib()
        slot = self._eas_tables.get(key)
        if slot is None:
            slot = self._build_table(L, entries, capturing)
            if len(self._eas_tables) >= 16:                      # eager steps with ever new gradient addresses: keep the captured ones
                for k in [k for k, v in self._eas_tables.items() if not v.get('captured')][:8]:
                    del self._eas_tables[k]
            self._eas_tables[key] = slot
        if capturing:
            slot['captured'] = True                                  # recorded into a graph now: never evicted
        hyper.beta1, hyper.beta2, hyper.eps = betas[0], betas[1], eps
        counter = None
        hyper.ema_ramp = 1.0
        if ema is not None:
            counter = ema['counter'].data_ptr()
            hyper.ema_updates, hyper.ema_decay, hyper.ema_ramp = counter, ema['decay'], ema['ramp']
        check(L.eas_adam_step_ex(slot['table'].data_ptr(), slot['n'], slot['blocks'], C.byref(hyper), stream()), 'eas_adam_step_ex')
        check(L.eas_adam_advance_steps_ex(slot['table'].data_ptr(), slot['n'], counter, stream()), 'eas_adam_advance_steps_ex')
        return None

    def _build_table(self, L, entries, capturing):
        esz, chunk = L.eas_adam_table_entry_bytes(), L.eas_adam_chunk()
        assert esz == _WORDS * 8
        ema = self._eas_ema
        rows = []                                                # (p, g, m, v, step, lr_ptr, ema, wd, numel, group)
        stepped = set()
        for p, g, m, v, st, lr, wd, gi in entries:
            twin = ema['twins'].get(p.data_ptr()) if ema is not None else None
            rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr(), lr.data_ptr() if torch.is_tensor(lr) else 0,
                         twin.data_ptr() if twin is not None else 0, wd, p.numel(), -1 if torch.is_tensor(lr) else gi))
            stepped.add(p.data_ptr())
        if ema is not None:                                      # what the average follows without an Adam step: BatchNorm running statistics,
            for src, a in ema['order']:                          # frozen parameters, parameters that received no gradient in this step
                if src.data_ptr() not in stepped:
                    rows.append((src.data_ptr(), 0, 0, 0, 0, 0, a.data_ptr(), 0.0, src.numel(), -1))
        n = len(rows)
        host = np.zeros(n * _WORDS, dtype=np.int64)
        view_f, view_i32 = host.view(np.float64), host.view(np.int32)
        blocks = 0
        for i, (p, g, m, v, st, lrp, tw, wd, numel, gi) in enumerate(rows):
            o = i * _WORDS
            host[o + 0], host[o + 1], host[o + 2], host[o + 3], host[o + 4], host[o + 5], host[o + 6] = p, g, m, v, st, lrp, tw
            view_f[o + 7] = 0.0
            view_f[o + 8] = wd
            host[o + 9] = numel
            host[o + 10] = blocks
            view_i32[(o + 11) * 2] = gi
            blocks += (numel + chunk - 1) // chunk
        dev = entries[0][0].device if entries else ema['order'][0][0].device
        if capturing:
            spares = self._eas_armed['spares']
            if not spares or spares[0].numel() < n * _WORDS:
                raise RuntimeError('FusedAdam: more distinct steps captured inside one capture_scope than tables were set aside')
            table = spares.pop(0)
            return {'table': table, 'n': n, 'blocks': blocks, 'pending': host, 'captured': True, 'keep': ema}     # (keep: the counter a graph reads)
        table = torch.empty(n * _WORDS, dtype=torch.int64, device=dev)
        # (a pageable copy: the staging buffer is the runtime's, the copy is complete when the call returns)
        table.copy_(torch.from_numpy(host))
        return {'table': table, 'n': n, 'blocks': blocks, 'pending': None, 'captured': False, 'keep': ema}
