"""A handful of launches of ONE conv layer on the split-operand F(4x4) kernel (for rocprofv3 --pmc passes): n c h w co [algo]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
n, c, h, w, co = (int(v) for v in sys.argv[1:6])
algo = {'f4s': ops.CONV_WINOGRAD_F4S, 'f4': ops.CONV_WINOGRAD_F4}[sys.argv[6] if len(sys.argv) > 6 else 'f4s']
dev = torch.device('cuda')
x = torch.randn(n, c, h, w, device=dev); wt = torch.randn(co, c, 3, 3, device=dev) * 0.05; b = torch.randn(co, device=dev)
wpk, wf4, wf4s = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4=True), ops.pack_conv_weight(wt, f4s=True)
bound = ops.amax(x)
y = torch.empty(n, co, h, w, device=dev)
for _ in range(6):
    ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4=wf4, wpk_f4s=wf4s, x_amax=bound, algo=algo, out=y)
torch.cuda.synchronize()
