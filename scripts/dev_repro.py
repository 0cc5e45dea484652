"""development: run-to-run differences of one training step of a golden model (which parameters, how large)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import eas_snn_amd  # noqa: puts compat/ on the path
import conftest  # noqa
import test_gpu_model as T
from spikingjelly.activation_based import functional
name = sys.argv[1] if len(sys.argv) > 1 else 'model_m_fullv2_t5_64x96_train'
dev = torch.device('cuda:0')
g, model = T._build(name, dev)
model.train(); model.head.use_l1 = True
x = torch.from_numpy(g['x']).to(dev); tg = torch.from_numpy(g['targets']).to(dev)
state = {k: v.clone() for k, v in model.state_dict().items()}
runs = []
for i in range(4):
    model.load_state_dict(state); model.zero_grad(set_to_none=True)
    out = model(x, tg); out['total_loss'].backward(); functional.reset_net(model)
    runs.append((out['total_loss'].item(), {n: p.grad.clone() for n, p in model.named_parameters()}))
names = [n for n, _ in model.named_parameters()]
for i in range(1, 4):
    bad = [n for n in names if not torch.equal(runs[0][1][n], runs[i][1][n])]
    print('run', i, 'loss', runs[0][0] == runs[i][0], 'differing params', len(bad), 'of', len(names))
    print('  identical:', [n for n in names if n not in bad])
