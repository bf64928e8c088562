import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for ci in (128, 16, 32, 64, 128, 256, 512, 128):
    x = torch.randn(n, ci, 180, 320, device=dev); wt = torch.randn(128, ci, 3, 3, device=dev) * 0.05
    wpk = ops.pack_conv_weight(wt)
    for _ in range(6): ops.conv2d(x, wpk, None, 128, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d(x, wpk, None, 128, 3)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'n={n} ci={ci:4d}: {ms:7.3f} ms  {2.0*n*57600*128*ci*9/ms/1e9:7.2f} TF/s', flush=True)
