"""Repeat tests/test_gpu_train.py::test_conv_gradients for one case many times in one process and report every relative
error (diagnosis of an intermittent failure): python scripts/flake_hunt.py [case] [algo] [repeats]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import test_gpu_train as T  # noqa: E402
from edvr_amd import functional as F_, ops  # noqa: E402

case = T.CONV_CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
algo = sys.argv[2] if len(sys.argv) > 2 else 'winograd'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ops.CONV_ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[algo]
gpu = torch.device('cuda:0')
n, c1, c2, h, w, co, ks, stride, actn, nres, out_mode = case
assert c2 == 0 and nres == 0 and out_mode == 0 and actn == 'lrelu'
worst = {}
for rep in range(reps):
    g = torch.Generator().manual_seed(11)
    m = torch.nn.Conv2d(c1, co, ks, stride, ks // 2)
    x1 = torch.randn(n, c1, h, w, generator=g)
    m64 = torch.nn.Conv2d(c1, co, ks, stride, ks // 2).double()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    leaf = x1.double().requires_grad_()
    y = F.leaky_relu(m64(leaf), 0.1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    m = m.to(gpu)
    xd = x1.to(gpu).requires_grad_()
    junk = torch.full((int(1e6) + rep * 4097,), float('nan'), device=gpu)  # poison freed memory: uninitialised reads show up as NaN
    del junk
    out = F_.conv(m, xd, act=2)
    out.backward(dy.to(gpu))
    errs = {'out': T._rel(out.detach(), y.detach()), 'dx': T._rel(xd.grad, leaf.grad), 'dw': T._rel(m.weight.grad, m64.weight.grad),
            'db': T._rel(m.bias.grad, m64.bias.grad)}
    for k, v in errs.items():
        if not (v < (2e-5 if k == 'out' else T.GRAD_RTOL)):
            print(f'rep {rep}: {k} = {v}', flush=True)
        worst[k] = max(worst.get(k, 0.0), v) if v == v else float('nan')
print('worst', worst)
