"""The small HBM-bound launches of the step, one by one (HIP events; A/B of library builds via EDVR_AMD_LIB):
conv_last forward (64 -> 3, 720x1280), its weight gradient (training resolution 256x256), the x2 bilinear upsampling of the PCD pyramid."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
label = sys.argv[1] if len(sys.argv) > 1 else 'default'


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
x = torch.randn(10, 64, 720, 1280, device=dev)
w = torch.randn(3, 64, 3, 3, device=dev) * 0.05
b = torch.randn(3, device=dev)
wpk = ops.pack_conv_weight(w)
ms = timed(lambda: ops.conv2d(x, wpk, b, 3, 3))
out.append(f'conv_last fwd 10x64x720x1280: {ms:.3f} ms ({(x.numel() + 10 * 3 * 720 * 1280) * 4 / ms / 1e9:.2f} TB/s)')
del x
x = torch.randn(32, 64, 256, 256, device=dev)
dz = torch.randn(32, 3, 256, 256, device=dev)
ms = timed(lambda: ops.conv2d_wgrad(x, None, None, dz, 3, 3, 1))
out.append(f'conv_last wgrad 32x64x256x256: {ms:.3f} ms ({(x.numel() + dz.numel()) * 4 / ms / 1e9:.2f} TB/s)')
del x, dz
for (n, c, h, w_) in [(50, 128, 90, 160), (50, 128, 45, 80), (50, 144, 90, 160)]:
    x = torch.randn(n, c, h, w_, device=dev)
    ms = timed(lambda: ops.upsample2x(x, 2.0))
    out.append(f'upsample2x {n}x{c}x{h}x{w_}: {ms:.3f} ms ({x.numel() * 5 * 4 / ms / 1e9:.2f} TB/s)')
print(f'{label:10s} ' + ' | '.join(out), flush=True)
# training-side glue: adjoint of the x2 upsampling, TSA temporal attention backward
out = []
for (n, c, h, w_) in [(160, 128, 32, 32), (160, 144, 32, 32), (160, 128, 16, 16)]:
    dy = torch.randn(n, c, 2 * h, 2 * w_, device=dev)
    ms = timed(lambda: ops.upsample2x_backward(dy, 2.0))
    out.append(f'upsample2x_bwd {n}x{c}x{h}x{w_}: {ms:.3f} ms ({dy.numel() * 1.25 * 4 / ms / 1e9:.2f} TB/s)')
emb = torch.randn(32, 5, 128, 64, 64, device=dev) * 0.1
ref = torch.randn(32, 128, 64, 64, device=dev) * 0.1
al = torch.randn(32, 5, 128, 64, 64, device=dev)
do = torch.randn(32, 5, 128, 64, 64, device=dev)
ms = timed(lambda: ops.tsa_temporal_backward(emb, ref, al, do), reps=10)
out.append(f'tsa_temporal_bwd 32x5x128x64x64: {ms:.3f} ms ({(6 * emb.numel() + 2 * ref.numel()) * 4 / ms / 1e9:.2f} TB/s)')
print(f'{label:10s} ' + ' | '.join(out), flush=True)
