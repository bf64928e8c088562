"""Launch time of conv2d_wgrad on the training layer (ablation builds compute wrong results on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
label = sys.argv[1] if len(sys.argv) > 1 else 'default'
out = []
for (n, ci, h, w, co) in [(160, 128, 64, 64, 128), (32, 128, 64, 64, 128), (20, 128, 180, 320, 128)]:
    x = torch.randn(n, ci, h, w, device=dev); dz = torch.randn(n, co, h, w, device=dev) * 1e-3
    ops.input_bound(x); ops.input_bound(dz)
    ops.set_f4s(training=True)
    for _ in range(3): ops.conv2d_wgrad(x, None, None, dz, co, 3, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv2d_wgrad(x, None, None, dz, co, 3, 1)
    e1.record(); torch.cuda.synchronize()
    out.append(f'{n}x{ci}x{h}x{w}: {e0.elapsed_time(e1) / 10:.3f} ms')
    del x, dz
print(f'{label:24s} ' + ' | '.join(out), flush=True)
