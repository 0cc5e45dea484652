import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29544'); os.environ.setdefault('RANK','0'); os.environ.setdefault('WORLD_SIZE','1')
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
torch.cuda.set_stream(torch.cuda.Stream())
sizes = [ (i % 7 + 1) * 4096 for i in range(230)]
grads = [torch.randn(n, device=dev) for n in sizes]
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n): fn()
    h = (time.perf_counter() - a) / n * 1e3
    torch.cuda.synchronize()
    return h
flat = torch.cat([g.reshape(-1) for g in grads])
print('cat host ms', t(lambda: torch.cat([g.reshape(-1) for g in grads])))
print('all_reduce host ms', t(lambda: dist.all_reduce(flat)))
print('mul host ms', t(lambda: flat.mul_(0.5)))
print('split+copy host ms', t(lambda: torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split(sizes), grads)])))
h = dist.all_reduce(flat, async_op=True)
print('all_reduce async host ms', t(lambda: dist.all_reduce(flat, async_op=True)))
dist.destroy_process_group()
