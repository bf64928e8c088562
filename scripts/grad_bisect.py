"""Find where the gradients of two runs of the same network (conv algorithm A vs B) start to differ, walking the autograd graph
backwards: every F_.conv / dcn / glue output keeps its gradient; the report lists, in BACKWARD order, the relative difference of
each output's gradient between the two runs.   python scripts/grad_bisect.py L_T7 direct winograd"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from util_edvr import build
from edvr_amd import functional as F_, ops
from edvr_amd.autograd import charbonnier_loss

name, algo_a, algo_b = sys.argv[1], sys.argv[2], sys.argv[3]
ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD, 'auto': ops.CONV_AUTO}
dev = torch.device('cuda')
net, x, kwargs = build(name)
net = net.train().to(dev)
names = {id(m): n for n, m in net.named_modules()}
kept = []


def wrap(fn, label):
    def inner(*a, **k):
        out = fn(*a, **k)
        if torch.is_tensor(out) and out.requires_grad:
            out.retain_grad()
            tag = label
            if a and isinstance(a[0], torch.nn.Module):
                tag += ':' + names.get(id(a[0]), '?')
            kept.append((tag, out))
        return out
    return inner


for fn in ('conv', 'dcn_from_packed', 'upsample2x', 'pool_maxavg', 'tsa_temporal', 'tsa_combine', 'upsample4x_add'):
    setattr(F_, fn, wrap(getattr(F_, fn), fn))
import edvr_amd.edvr_arch as EA, edvr_amd.arch_util as AU  # they call F_.<fn> through the module attribute: already patched


def run(algo):
    kept.clear()
    ops.CONV_ALGO = ALGO[algo]
    net.zero_grad(set_to_none=True)
    out = net(x.to(dev))
    gt = torch.rand(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    charbonnier_loss(out, gt).backward()
    return [(t, o.detach().clone(), None if o.grad is None else o.grad.clone()) for t, o in kept]


A, B = run(algo_a), run(algo_b)
from collections import defaultdict


def keyed(rows):  # (tag, k-th occurrence) -> row; the fused residual-block node of one algorithm hides its convs from the other
    seen, out = defaultdict(int), {}
    for t, o, g in rows:
        out[(t, seen[t])] = (o, g)
        seen[t] += 1
    return out


KA, KB = keyed(A), keyed(B)
rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
print(f'{len(A)} / {len(B)} recorded outputs; walking backwards (last forward op first): d(out) {algo_a} vs {algo_b}')
seen = defaultdict(int)
order = []
for t, o, g in A:
    order.append((t, seen[t]))
    seen[t] += 1
for key in reversed(order):
    if key not in KB:
        continue
    (oa, ga), (ob, gb) = KA[key], KB[key]
    if ga is None or gb is None:
        continue
    e, f = rel(ga, gb), rel(oa, ob)
    print(f'{e:.2e}  (forward diff {f:.1e})  {key[0]}#{key[1]} {tuple(oa.shape)}' + ('   <<<<' if e > 1e-4 else ''))
