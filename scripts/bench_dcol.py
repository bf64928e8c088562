"""dcol = W^T dY of the DCNv2 backward as a 1x1 convolution (128 -> 1152 channels): streaming kernel vs direct kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
for (n, ci, h, w, co) in [(160, 128, 64, 64, 1152), (160, 128, 32, 32, 1152), (20, 128, 180, 320, 1152)]:
    x = torch.randn(n, ci, h, w, device=dev)
    wpk = ops.pack_conv_weight(torch.randn(co, ci, 1, 1, device=dev) * 0.05)
    out = torch.empty(n, co, h, w, device=dev)
    for name, algo in (('stream', ops.CONV_AUTO), ('direct', ops.CONV_DIRECT)):
        for _ in range(2):
            ops.conv2d(x, wpk, None, co, 1, out=out, algo=algo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv2d(x, wpk, None, co, 1, out=out, algo=algo)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f'dcol n={n} {ci}->{co} {h}x{w} {name}: {ms:7.3f} ms  {2.0 * n * h * w * ci * co / ms / 1e9:7.2f} TF/s  write {4.0 * n * co * h * w / ms / 1e9:6.2f} TB/s', flush=True)
