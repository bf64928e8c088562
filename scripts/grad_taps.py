"""d(loss)/d(aligned | fused | trunk) of the HIP run against the fp64 oracle (pool inputs following the HIP run):
python scripts/grad_taps.py L_T7 direct"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from util_edvr import build, oracle_kwargs
from edvr_amd import functional as F_, ops
from edvr_amd.autograd import charbonnier_loss
from oracle import dcn_oracle as O, edvr_oracle as EO
name = sys.argv[1]
ops.CONV_ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD, 'auto': ops.CONV_AUTO}[sys.argv[2]]
dev = torch.device('cuda')
net, x, kwargs = build(name)
net.train()
state = {k: v.detach().clone() for k, v in net.state_dict().items()}
pool_in = []
real_pool = F_.pool_maxavg
F_.pool_maxavg = lambda t: (pool_in.append(t.detach().cpu()), real_pool(t))[1]
net = net.to(dev)
net.taps = {}
out = net(x.to(dev))
for v in net.taps.values():
    v.retain_grad()
sd = {k: v.double().requires_grad_() for k, v in state.items()}
taps = {}
o64 = EO.edvr_forward(sd, x.double(), dcn=O.dcnv2_c, pool_inputs=pool_in, taps=taps, **oracle_kwargs(kwargs))
for v in taps.values():
    v.retain_grad()
gt = torch.rand(o64.shape, generator=torch.Generator().manual_seed(1))
EO.charbonnier_sum(o64, gt.double()).backward()
charbonnier_loss(out, gt.to(dev)).backward()
rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
for k in ('trunk', 'fused', 'aligned'):
    print(f'{k}: forward {rel(net.taps[k].detach(), taps[k].detach()):.2e}   d/d{k} {rel(net.taps[k].grad, taps[k].grad):.2e}')
    d = (net.taps[k].grad.double().cpu() - taps[k].grad).abs()
    i = torch.nonzero(d == d.max())[0].tolist()
    print('   worst element', i, 'ours', net.taps[k].grad.cpu()[tuple(i)].item(), 'oracle', taps[k].grad[tuple(i)].item(),
          ' elements with |diff| > 1e-3 max|ref|:', int((d > 1e-3 * taps[k].grad.abs().max()).sum()))
