"""torch.optim.Adam whose step is ONE launch over all parameter groups (eas_adam_step, csrc/adam.hip) instead of torch's eleven multi-tensor
kernels + six step-counter launches per step of SYOLOX-S.  Same class hierarchy, same state (``step`` / ``exp_avg`` / ``exp_avg_sq`` per
parameter: checkpoints of either implementation load into the other), same update rule (torch's fused kernel, with its float / double
promotions).  Reference: the optimizer of yolox/exp/event_yolox_base.py:352-414 (``torch.optim.Adam``, five parameter groups)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, stream

ENABLED = os.environ.get('EAS_FUSED_ADAM', '1') != '0'      # 0: torch's fused implementation


class FusedAdam(torch.optim.Adam):
    """``torch.optim.Adam(..., fused=True)`` with the step on the own kernel when every parameter is a dense fp32 CUDA tensor (else torch's)."""

    def __init__(self, params, **kw):
        kw.setdefault('fused', True)
        super().__init__(params, **kw)
        self._eas_tables = {}
        self._eas_spare = None

    def _eligible(self, group):
        return (ENABLED and not group['amsgrad'] and not group['maximize'] and not group.get('differentiable', False)
                and not group.get('decoupled_weight_decay', False) and not isinstance(group['betas'][0], torch.Tensor)
                and not isinstance(group['betas'][1], torch.Tensor))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(self, 'grad_scale', None) is not None or getattr(self, 'found_inf', None) is not None:
            return super().step(closure)
        entries = []
        betas = eps = None
        for group in self.param_groups:
            if not self._eligible(group):
                return super().step()
            if betas is None:
                betas, eps = tuple(float(b) for b in group['betas']), float(group['eps'])
            elif betas != tuple(float(b) for b in group['betas']) or eps != float(group['eps']):
                return super().step()                                        # one (beta1, beta2, eps) per launch
            ps, gs, ms, vs, mx, steps = [], [], [], [], [], []
            if self._init_group(group, ps, gs, ms, vs, mx, steps):            # complex parameters
                return super().step()
            lr = group['lr']
            for p, g, m, v, st in zip(ps, gs, ms, vs, steps):
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse and p.is_contiguous()
                        and g.is_contiguous() and m.is_contiguous() and v.is_contiguous() and st.is_cuda and st.dtype == torch.float32):
                    return super().step()
                entries.append((p, g, m, v, st, lr, float(group['weight_decay'])))
        if not entries:
            return None
        key = tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr(),
                     lr.data_ptr() if torch.is_tensor(lr) else float(lr), wd, p.numel()) for p, g, m, v, st, lr, wd in entries)
        L = _lib.lib()
        n = len(entries)
        slot = self._eas_tables.get(key)
        if slot is not None and torch.cuda.is_current_stream_capturing():
            slot['captured'] = True                                  # recorded into a graph now: never evicted
        if slot is None and torch.cuda.is_current_stream_capturing() and (self._eas_spare is None or self._eas_spare.numel() != n * 10):
            return super().step()          # a capture before any eager step of this parameter set: no ordinary-memory table to record (see below)
        if slot is None:
            esz, chunk = L.eas_adam_table_entry_bytes(), L.eas_adam_chunk()
            assert esz == 80
            host = np.zeros(n * 10, dtype=np.int64)
            view_f = host.view(np.float64)
            blocks = 0
            for i, (p, g, m, v, st, lr, wd) in enumerate(entries):
                if torch.is_tensor(lr) and not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
                    return super().step()
                o = i * 10
                host[o + 0], host[o + 1], host[o + 2], host[o + 3], host[o + 4] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr()
                host[o + 5] = lr.data_ptr() if torch.is_tensor(lr) else 0
                view_f[o + 6] = 0.0 if torch.is_tensor(lr) else float(lr)
                view_f[o + 7] = wd
                host[o + 8] = p.numel()
                host[o + 9] = blocks
                blocks += (p.numel() + chunk - 1) // chunk
            # A table per set of addresses, never overwritten: a captured step keeps replaying with ITS table while eager steps (other
            # gradient addresses) come and go.  Inside a capture nothing can be copied from the host (pinned-memory bookkeeping records events
            # on the capturing stream; a pageable copy would be replayed from a dead staging buffer): the table is filled right after the
            # capture (``sync_tables``, called by TrainStep.capture) -- a capture only records the launches, it does not run them.  And the
            # table of a captured step must be ordinary memory, allocated by an eager step before (the spare): memory taken from the graph's
            # private pool during the capture did not keep what was written into it afterwards (illegal addresses at the first replay).
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing and self._eas_spare is not None and self._eas_spare.numel() == n * 10:
                table, self._eas_spare = self._eas_spare, None       # allocated by an eager step: ordinary memory, not the graph's private pool
            else:
                table = torch.empty(n * 10, dtype=torch.int64, device=entries[0][0].device)
            if not capturing and self._eas_spare is None:
                self._eas_spare = torch.empty(n * 10, dtype=torch.int64, device=entries[0][0].device)
            slot = {'table': table, 'blocks': blocks, 'pending': None}
            if torch.cuda.is_current_stream_capturing():
                slot['pending'] = host
            else:
                table.copy_(torch.from_numpy(host))
            if len(self._eas_tables) >= 16:                      # eager steps with ever new gradient addresses: keep the captured ones
                for k in [k for k, v in self._eas_tables.items() if not v.get('captured')][:8]:
                    del self._eas_tables[k]
            slot['captured'] = torch.cuda.is_current_stream_capturing()
            self._eas_tables[key] = slot
        check(L.eas_adam_step(slot['table'].data_ptr(), n, slot['blocks'], betas[0], betas[1], eps, stream()), 'eas_adam_step')
        check(L.eas_adam_advance_steps(slot['table'].data_ptr(), n, stream()), 'eas_adam_advance_steps')
        return None

    def sync_tables(self):
        """fill the tables made inside a graph capture (see ``step``); call after the capture has ended and before the first replay"""
        for slot in self._eas_tables.values():
            if slot['pending'] is not None:
                slot['table'].copy_(torch.from_numpy(slot['pending']))
                slot['pending'] = None
