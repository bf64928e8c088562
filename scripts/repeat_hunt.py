"""Bitwise-repeatability hunt for the conv forward / data-gradient / weight-gradient launches (diagnosis of the intermittent
tests/test_gpu_train.py::test_conv_gradients failure of round 1).

Every kernel on this path is deterministic (no atomics: split-K partials are reduced in a fixed order), so two executions on the
same inputs must agree BIT FOR BIT.  For every case x algorithm this runs forward + backward `reps` times and compares each
tensor with the first repetition.  `torch.empty` is patched inside edvr_amd.ops so that every output / workspace / packed-weight
buffer starts as NaN: an element a kernel forgets to write shows up as NaN instead of as whatever the allocator handed out.

    python scripts/repeat_hunt.py [reps] [case ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import test_gpu_train as T  # noqa: E402
from edvr_amd import functional as F_, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cases = [int(a) for a in sys.argv[2:]] or list(range(len(T.CONV_CASES)))
gpu = torch.device('cuda:0')

_real_empty = torch.empty


def _poisoned_empty(*a, **k):
    t = _real_empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float('nan'))
    elif t.dtype == torch.uint8 and t.is_cuda:
        t.fill_(0xFF)  # workspace bytes: 0xFFFFFFFF is a NaN pattern
    return t


def _poisoned_empty_like(t, **k):
    return _poisoned_empty(t.shape, dtype=k.get('dtype', t.dtype), device=k.get('device', t.device))


class _TorchProxy:
    def __getattr__(self, name):
        return {'empty': _poisoned_empty, 'empty_like': _poisoned_empty_like}.get(name) or getattr(torch, name)


ops.torch = _TorchProxy()
ops._WS.clear()

bad = 0
for ci in cases:
    n, c1, c2, h, w, co, ks, stride, actn, nres, out_mode = T.CONV_CASES[ci]
    for algo in ('direct', 'winograd'):
        ops.CONV_ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[algo]
        g = torch.Generator().manual_seed(11)
        m = torch.nn.Conv2d(c1 + c2, co, ks, stride, ks // 2).to(gpu)
        x1 = torch.randn(n, c1, h, w, generator=g).to(gpu)
        x2 = torch.randn(n, c2, h, w, generator=g).to(gpu) if c2 else None
        act, act_from = {'none': (0, 0), 'relu': (1, 0), 'lrelu': (2, 0), 'sigmoid_from': (3, 2 * co // 3)}[actn]
        ho, wo = (h + 2 * (ks // 2) - ks) // stride + 1, (w + 2 * (ks // 2) - ks) // stride + 1
        res = [torch.randn(n, co, ho, wo, generator=g).to(gpu) for _ in range(nres)]
        dy = None
        first = None
        nbad = 0
        for rep in range(reps):
            ops._WS.clear()  # a fresh (poisoned) workspace each repetition
            ops._PACKED.clear()
            leaves = [t.detach().clone().requires_grad_() for t in [x1] + ([x2] if c2 else []) + res]
            rs = leaves[(2 if c2 else 1):]
            m.zero_grad(set_to_none=True)
            out = F_.conv(m, leaves[0], x2=leaves[1] if c2 else None, act=act, act_from=act_from, res1=rs[0] if nres > 0 else None,
                          res2=rs[1] if nres > 1 else None, out_mode=out_mode)
            if dy is None:
                dy = torch.randn(out.shape, generator=g).to(gpu)
            out.backward(dy)
            got = {'out': out.detach(), 'dw': m.weight.grad, 'db': m.bias.grad}
            for i, t in enumerate(leaves):
                got[f'd_in{i}'] = t.grad
            got = {k: v.clone() for k, v in got.items()}
            if first is None:
                first = got
                for k, v in got.items():
                    if not torch.isfinite(v).all():
                        print(f'case {ci} {algo}: {k} has non-finite values in repetition 0', flush=True)
                        nbad += 1
                continue
            for k, v in got.items():
                if not torch.equal(v, first[k]):
                    diff = (v - first[k]).abs()
                    nz = int((v != first[k]).sum().item())
                    print(f'case {ci} {algo} rep {rep}: {k} differs in {nz} elements, max |d| {diff.max().item():.3e} '
                          f'(nan: {int(torch.isnan(v).sum().item())}) first idx {torch.nonzero(v != first[k])[0].tolist()}', flush=True)
                    nbad += 1
        print(f'case {ci} {T.CONV_CASES[ci]} {algo}: {reps} repetitions, {nbad} mismatches', flush=True)
        bad += nbad
ops.CONV_ALGO = ops.CONV_AUTO
print('TOTAL mismatches', bad)
sys.exit(1 if bad else 0)
