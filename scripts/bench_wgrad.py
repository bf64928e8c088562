"""Micro-benchmark of the wgrad kernel on training shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
SHAPES = [(160, 128, 64, 64, 128, 3, 1), (32, 128, 64, 64, 128, 3, 1), (160, 256, 64, 64, 128, 3, 1), (160, 128, 64, 64, 216, 3, 1),
          (160, 128, 32, 32, 128, 3, 1), (32, 64, 256, 256, 64, 3, 1), (32, 640, 64, 64, 128, 1, 1)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[1].split(',')]
dev = torch.device('cuda')
for (n, ci, h, w, co, ks, st) in SHAPES:
    x = torch.randn(n, ci, h, w, device=dev)
    dz = torch.randn(n, co, h // st, w // st, device=dev)
    for _ in range(2):
        dw = ops.conv2d_wgrad(x, None, None, dz, co, ks, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        dw = ops.conv2d_wgrad(x, None, None, dz, co, ks, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n * dz.shape[2] * dz.shape[3] * co * ci * ks * ks
    print(f'wgrad n={n:3d} ci={ci:4d} {h}x{w} co={co:3d} k{ks} s{st}: {ms:8.3f} ms  {flops / ms / 1e9:7.2f} TF/s', flush=True)
