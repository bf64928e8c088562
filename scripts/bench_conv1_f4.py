"""One conv shape on the F(4x4,3x3) kernel for PMC profiling (n=20, 128->128, 180x320, bias + LeakyReLU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
x = torch.randn(20, 128, 180, 320, device=dev); w = torch.randn(128, 128, 3, 3, device=dev) * 0.05; b = torch.randn(128, device=dev)
wpk, wf4 = ops.pack_conv_weight(w), ops.pack_conv_weight(w, f4=True)
for _ in range(8): ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4)
torch.cuda.synchronize()
