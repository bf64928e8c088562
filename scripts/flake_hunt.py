"""Root-cause hunt for the intermittent tests/test_gpu_train.py::test_conv_gradients[winograd-case0] failure of round 1.

Round-1 facts: one failure in ~25 executions, only the Winograd variant, same binary green before and after.  The kernels are
bit-reproducible (scripts/repeat_hunt.py: 16 case x algorithm combinations x 400 repetitions with NaN-poisoned outputs and
workspaces, 0 mismatches), so the variation had to come from the INPUTS: the test drew its conv weights from torch's global RNG
(nn.Conv2d's default init), whose state depends on which tests ran before.

Hypothesis checked here: the gradient of act(z) is discontinuous at z = 0 (ReLU / LeakyReLU).  The HIP path takes the derivative
from the sign of its fp32 output, the oracle from its fp64 one; when some |z| is below the forward rounding error the two can pick
different sides, and ONE such element changes dz by 0.9 * dy there - far above the 5e-4 gradient tolerance.  For every seed this
script runs the test's computation, and on a gradient mismatch re-runs the oracle with the activation derivative taken from the
HIP forward's signs: if that oracle agrees, the mismatch is the tie-break, not a kernel error.

    python scripts/flake_hunt.py [case] [algo] [n_seeds]
"""
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
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ops.CONV_ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[algo]
gpu = torch.device('cuda:0')
n, c1, c2, h, w, co, ks, stride, actn, nres, out_mode = case
assert c2 == 0 and nres == 0 and out_mode == 0 and actn == 'lrelu'


def oracle(m64, x1, dy, slope_mask=None):
    leaf = x1.double().requires_grad_()
    z = m64(leaf)
    if slope_mask is None:
        y = F.leaky_relu(z, 0.1)
    else:
        y = torch.where(slope_mask, z, 0.1 * z)  # derivative side dictated by the caller
    m64.zero_grad()
    y.backward(dy.double())
    return z.detach(), y.detach(), leaf.grad, m64.weight.grad.clone(), m64.bias.grad.clone()


fails = explained = 0
worst_fwd = 0.0
for seed in range(seeds):
    torch.manual_seed(seed)  # what the round-1 test did NOT do: the weights came from whatever state the global RNG was in
    g = torch.Generator().manual_seed(11)
    m = torch.nn.Conv2d(c1, co, ks, stride, ks // 2)
    x1 = torch.randn(n, c1, h, w, generator=g)
    m64 = torch.nn.Conv2d(c1, co, ks, stride, ks // 2).double()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    dy = torch.randn(n, co, h, w, generator=g)
    z64, y64, dx64, dw64, db64 = oracle(m64, x1, dy)
    m = m.to(gpu)
    xd = x1.to(gpu).requires_grad_()
    out = F_.conv(m, xd, act=2)
    out.backward(dy.to(gpu))
    e_fwd = T._rel(out.detach(), y64)
    worst_fwd = max(worst_fwd, e_fwd)
    errs = {'dx': T._rel(xd.grad, dx64), 'dw': T._rel(m.weight.grad, dw64), 'db': T._rel(m.bias.grad, db64)}
    if e_fwd < 2e-5 and all(v < T.GRAD_RTOL for v in errs.values()):
        continue
    fails += 1
    side_gpu = out.detach().cpu() > 0
    side_64 = z64 > 0
    flips = side_gpu != side_64
    zmax = z64.abs().max().item()
    _, _, dx2, dw2, db2 = oracle(m64, x1, dy, slope_mask=side_gpu)
    errs2 = {'dx': T._rel(xd.grad, dx2), 'dw': T._rel(m.weight.grad, dw2), 'db': T._rel(m.bias.grad, db2)}
    ok2 = all(v < T.GRAD_RTOL for v in errs2.values())
    explained += bool(ok2 and flips.any())
    print(f'seed {seed}: forward {e_fwd:.2e}; grad errors {({k: f"{v:.2e}" for k, v in errs.items()})}; '
          f'{int(flips.sum())} element(s) on different sides of 0, |z64| there = {[f"{v:.2e}" for v in z64[flips].abs().tolist()]} '
          f'(max|z| {zmax:.2f}); with the HIP forward\'s sides in the oracle: {({k: f"{v:.2e}" for k, v in errs2.items()})} -> '
          f'{"EXPLAINED by the tie-break" if ok2 and flips.any() else "NOT explained"}', flush=True)
print(f'case {case} {algo}: {seeds} seeds, {fails} gradient mismatches, {explained} explained by a derivative tie-break at |z| ~ 0; '
      f'worst forward error {worst_fwd:.2e}')
sys.exit(0 if fails == explained else 1)
