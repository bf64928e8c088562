"""Micro-benchmark of the conv kernel on the shapes that dominate EDVR (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops

SHAPES = [  # n, ci, h, w, co, ks, stride
    (20, 128, 180, 320, 128, 3, 1),
    (4, 128, 180, 320, 128, 3, 1),
    (20, 256, 180, 320, 128, 3, 1),
    (20, 128, 180, 320, 216, 3, 1),
    (20, 64, 180, 320, 64, 3, 1),
    (4, 64, 720, 1280, 64, 3, 1),
    (4, 128, 180, 320, 512, 3, 1),
    (20, 128, 90, 160, 128, 3, 1),
    (20, 128, 45, 80, 128, 3, 1),
    (20, 128, 180, 320, 128, 3, 2),
    (4, 640, 180, 320, 128, 1, 1),
    (20, 1152, 180, 320, 128, 1, 1),
]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[1].split(',')]
F4 = len(sys.argv) > 2 and sys.argv[2] == 'f4'  # 3x3 / stride-1 shapes on the F(4x4,3x3) kernel
dev = torch.device('cuda')
for (n, ci, h, w, co, ks, st) in SHAPES:
    x = torch.randn(n, ci, h, w, device=dev)
    wt = torch.randn(co, ci, ks, ks, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    wpk = ops.pack_conv_weight(wt)
    kw = dict(wpk_f4=ops.pack_conv_weight(wt, f4=True), algo=ops.CONV_WINOGRAD_F4) if (F4 and ks == 3 and st == 1) else {}
    for _ in range(2):
        y = ops.conv2d(x, wpk, b, co, ks, stride=st, act=ops.ACT_LRELU, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        y = ops.conv2d(x, wpk, b, co, ks, stride=st, act=ops.ACT_LRELU, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n * y.shape[2] * y.shape[3] * co * ci * ks * ks
    print(f'n={n:3d} ci={ci:4d} {h}x{w} co={co:3d} k{ks} s{st}: {ms:8.3f} ms  {flops / ms / 1e9:7.2f} TF/s', flush=True)
