import torch, time
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream())
for numel in (64, 1 << 20):
    xs = [torch.zeros(numel, device=dev) for _ in range(8)]
    def body(n):
        for i in range(n):
            xs[i % 8].add_(1.0)
    for n in (200, 1000, 2000):
        body(n); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(n)
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        print(f'numel {numel}: graph of {n} kernels: {dt * 1e3:.3f} ms = {dt / n * 1e6:.2f} us per node', flush=True)
