"""Which conv inputs of an EDVR-L forward / training step arrive without a magnitude bound (ops.input_bound runs the reduction kernel
for them)?  python scripts/amax_passes.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from edvr_amd import ops
from edvr_amd.edvr_arch import EDVR
torch.manual_seed(0)
net = EDVR(num_feat=128, num_reconstruct_block=40, num_frame=5).cuda()
x = torch.rand(2, 5, 3, 64, 64, device='cuda')
for mode in ('infer', 'train'):
    for rep in range(2):
        ops.AMAX_LOG = []
        if mode == 'infer':
            with torch.no_grad():
                net(x)
        else:
            net.train()
            net(x).sum().backward()
        torch.cuda.synchronize()
    c = collections.Counter((sh, tuple(st)) for sh, st in ops.AMAX_LOG)
    print(f'== {mode}: {len(ops.AMAX_LOG)} reduction passes')
    for (sh, st), k in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f'  {k:3d} x {sh}  <- {" > ".join(st)}')
ops.AMAX_LOG = None
