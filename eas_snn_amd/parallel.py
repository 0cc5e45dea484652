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
    def __init__(self, module, process_group=None, broadcast_parameters=True):
        self.group = process_group
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if broadcast_parameters and dist.is_initialized():
            with torch.no_grad():
                flat = torch.cat([p.detach().reshape(-1) for p in self.params])
                dist.broadcast(flat, src=0, group=process_group)
                torch._foreach_copy_([p.detach() for p in self.params], [c.view_as(p) for c, p in zip(flat.split([p.numel() for p in self.params]), self.params)])

    def sync(self):
        """average the gradients over the ranks (call between backward and the optimizer step)"""
        if self.world == 1 and not dist.is_initialized():
            return
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError('FlatGradAllReduce: a parameter has no gradient (unused parameters need DistributedDataParallel)')
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)      # SUM + scale: every backend has it (gloo has no AVG)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
