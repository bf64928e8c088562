"""conv_last shape alone (n=4, 64 -> 3, 720x1280) for PMC profiling of conv3x3_smallco_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
x = torch.randn(4, 64, 720, 1280, device=dev); w = torch.randn(3, 64, 3, 3, device=dev) * 0.05; b = torch.randn(3, device=dev)
wpk = ops.pack_conv_weight(w)
for _ in range(8): ops.conv2d(x, wpk, b, 3, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv2d(x, wpk, b, 3, 3)
e1.record()
torch.cuda.synchronize()
print(f'conv_last 4x64x720x1280 -> 3: {e0.elapsed_time(e1) / 20:.3f} ms per launch')
