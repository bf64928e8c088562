"""Per-tensor gradient error of the HIP training path vs the fp64 oracle, next to the fp32-oracle noise floor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from util_edvr import build, oracle_kwargs
from edvr_amd.autograd import charbonnier_loss
from oracle import dcn_oracle as O, edvr_oracle as EO

name = sys.argv[1] if len(sys.argv) > 1 else 'M_T5'
if len(sys.argv) > 2:
    from edvr_amd import ops
    ops.CONV_ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD, 'auto': ops.CONV_AUTO}[sys.argv[2]]
net, x, kwargs = build(name)
net.train()
state = {k: v.detach().clone() for k, v in net.state_dict().items()}
dev = torch.device('cuda')
from util_edvr import DecisionRecorder
net = net.to(dev)
with DecisionRecorder(net) as rec:
    out_hip = net(x.to(dev))
rec.bind(net)
follow = '--nofollow' not in sys.argv
def oracle_grads(dt):
    sd = {k: v.to(dt).requires_grad_() for k, v in state.items()}
    out = EO.edvr_forward(sd, x.to(dt), dcn=O.dcnv2_c, **(rec.oracle_kwargs() if follow else {}), **oracle_kwargs(kwargs))
    gt = torch.rand(out.shape, generator=torch.Generator().manual_seed(1))
    EO.charbonnier_sum(out, gt.to(dt)).backward()
    return gt, {k: v.grad for k, v in sd.items()}
gt, g64 = oracle_grads(torch.float64)
_, g32 = oracle_grads(torch.float32)
charbonnier_loss(out_hip, gt.to(dev)).backward()
rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
rows = []
for k, p in net.named_parameters():
    if g64[k].abs().max() == 0: continue
    rows.append((rel(p.grad, g64[k]), rel(g32[k], g64[k]), k))
order = [k for k, _ in net.named_parameters()]
print('--- in network order (every 6th tensor)')
for o, f, k in sorted(rows, key=lambda r: order.index(r[2]))[::6]:
    print(f'{o:.2e}  floor {f:.2e}  {k}')
rows.sort(reverse=True)
print('--- worst')
for o, f, k in rows[:12]:
    print(f'{o:.2e}  floor {f:.2e}  ratio {o / max(f, 1e-12):5.1f}  {k}')
import statistics
print('median ours', statistics.median(r[0] for r in rows), 'median floor', statistics.median(r[1] for r in rows))
