"""One DCN forward shape (B=20, C=128, 180x320, sigma 1) for PMC profiling."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
B, C, H, W = 20, 128, 180, 320
x = torch.randn(B, C, H, W, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05; b = torch.randn(C, device=dev)
off = torch.randn(B, 144, H, W, device=dev); m = torch.rand(B, 72, H, W, device=dev)
for _ in range(6): y = ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8)
torch.cuda.synchronize()
