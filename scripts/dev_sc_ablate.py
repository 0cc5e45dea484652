import os, sys
sys.path.insert(0, '/root/repo')
import torch
from eas_snn_amd import ops
dev = torch.device('cuda:0')
os.environ['EAS_SC_FORM'] = 'mfma'
x = torch.randn(64, 4, 256, 320, device=dev); w = torch.randn(4, 4, 5, 5, device=dev) * 0.2; b = torch.randn(4, device=dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print('dbg', os.environ.get('EAS_SC_DBG', '0'), f'{timeit(lambda: ops.smallconv_fwd(x, w, b, relu=True)):.1f} us')
