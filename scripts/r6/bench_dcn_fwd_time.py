"""Split tap-window DCN forward on the headline L1 layer (ablation builds compute wrong results on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
out = []
for (B, C, H, W) in [(50, 128, 180, 320), (160, 128, 64, 64)]:
    dg = 8
    x = torch.randn(B, C, H, W, device=dev, generator=g); w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05; b = torch.randn(C, device=dev, generator=g)
    off = (torch.randn(1, dg * 18, 1, 1, device=dev, generator=g) * 0.5 + torch.randn(B, dg * 18, H, W, device=dev, generator=g) * 0.02).contiguous()
    m = torch.rand(B, dg * 9, H, W, device=dev, generator=g)
    bound = ops.amax(x); y = torch.empty(B, C, H, W, device=dev)
    run = lambda: ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, dg, act=ops.ACT_LRELU, halo_hint=ops.DCN_HALO_TAPWIN, out=y, xm_bound=bound)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    out.append(f'{B}x{C}x{H}x{W}: {e0.elapsed_time(e1) / 10:.3f} ms')
print(f'{sys.argv[1] if len(sys.argv) > 1 else "default":14s} ' + ' | '.join(out), flush=True)
