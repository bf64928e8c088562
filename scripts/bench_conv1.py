"""One conv shape for PMC profiling: python bench_conv1.py [fwd|wgrad]  (n=20, 128->128, 180x320 forward; n=160, 128->128, 64x64 wgrad)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
if len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
    x = torch.randn(160, 128, 64, 64, device=dev); dz = torch.randn(160, 128, 64, 64, device=dev)
    for _ in range(8): ops.conv2d_wgrad(x, None, None, dz, 128, 3, 1)
else:
    x = torch.randn(20, 128, 180, 320, device=dev); w = torch.randn(128, 128, 3, 3, device=dev) * 0.05; b = torch.randn(128, device=dev)
    wpk = ops.pack_conv_weight(w)
    for _ in range(8): ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU)
torch.cuda.synchronize()
