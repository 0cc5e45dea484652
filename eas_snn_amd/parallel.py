"""Data-parallel gradient exchange for one process per GPU over RCCL/xGMI.

The reference wraps the model in ``DistributedDataParallel`` (yolox/core/trainer.py:174-176); the compat trainer does the
same.  ``FlatGradAllReduce`` is the lean equivalent for a step that has no unused parameters: after ``backward`` every
gradient is packed into ONE contiguous buffer (a handful of launches), averaged over the ranks with ONE all-reduce
(35.8 MB for SYOLOX-S: ~0.6 ms on eight xGMI-connected GPUs) and unpacked into the ``.grad`` tensors.  It trades DDP's
overlap of that all-reduce with the backward pass for ~7 ms less host work per step (DDP's per-parameter hooks and bucket
bookkeeping; measured on MI355X: host enqueue share 0.83 of a 32 ms step with DDP, 0.60 without) -- with eight ranks
launching ~1100 kernels per step from one host, staying ahead of the GPU matters more than hiding half a millisecond.
Initial parameter values are broadcast from rank 0 exactly as DDP's constructor does.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """``pack()`` -> ``reduce()`` -> ``attach()`` between backward and the optimizer step (``sync()`` does all three).

    The flat buffer is allocated once, so the three phases can live in different HIP graphs: ``pack`` at the end of a captured
    forward+backward, ``reduce`` (the RCCL call) launched eagerly between two graph replays, ``attach`` at the start of a captured
    optimizer step.  ``attach`` launches nothing: it makes every ``p.grad`` a view of the reduced buffer, which the optimizer
    reads in place (no unpack copy)."""

    def __init__(self, module, process_group=None, broadcast_parameters=True):
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat = None
        self.views = None
        if broadcast_parameters and dist.is_initialized():
            with torch.no_grad():
                flat = torch.cat([p.detach().reshape(-1) for p in self.params])
                dist.broadcast(flat, src=0, group=process_group)
                torch._foreach_copy_([p.detach() for p in self.params], [c.view_as(p) for c, p in zip(flat.split(self.sizes), self.params)])

    def pack(self):
        """all gradients -> the flat buffer (one concatenation kernel family)"""
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError('FlatGradAllReduce: a parameter has no gradient (unused parameters need DistributedDataParallel)')
        if self.flat is None:
            self.flat = torch.empty(sum(self.sizes), dtype=grads[0].dtype, device=grads[0].device)
            self.views = [c.view_as(p) for c, p in zip(self.flat.split(self.sizes), self.params)]
        stray = [i for i, (g, v) in enumerate(zip(grads, self.views)) if g.data_ptr() != v.data_ptr()]
        if not stray:
            return                                   # every gradient was accumulated into its attached view: already packed
        if len(stray) == len(grads):
            torch.cat([g.reshape(-1) for g in grads], out=self.flat)
        else:                                        # a mix (partial zero_grad / partial backward): copy the ones that live elsewhere
            torch._foreach_copy_([self.views[i] for i in stray], [grads[i] for i in stray])

    def reduce(self):
        """ONE all-reduce of the flat buffer, then the 1/world scale (SUM + scale: every backend has it, gloo has no AVG)"""
        if dist.is_initialized():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                self.flat.mul_(1.0 / self.world)

    def attach(self):
        """``p.grad`` := the parameter's slice of the reduced buffer (host-side pointer assignment, no launch)"""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def sync(self):
        """average the gradients over the ranks (call between backward and the optimizer step)"""
        if self.world == 1 and not dist.is_initialized():
            return
        self.pack()
        self.reduce()
        self.attach()
