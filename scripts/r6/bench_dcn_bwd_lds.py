"""DCNv2 backward, LDS-window path, on the trained-like field of scripts/bench_dcn_bwd_paths.py (ablation builds give wrong results on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
B, C, H, W, dg = 160, 128, 64, 64, 8
x = torch.randn(B, C, H, W, device=dev, generator=g)
w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
m = torch.rand(B, dg * 9, H, W, device=dev, generator=g)
dy = torch.randn(B, C, H, W, device=dev, generator=g) * 1e-3
bx, bd = ops.amax(x), ops.amax(dy)
coarse = torch.randn(B, dg * 18, H // 16 + 1, W // 16 + 1, device=dev, generator=g) * 0.5
off = (torch.randn(1, dg * 18, 1, 1, device=dev, generator=g) * 4.0 + F.interpolate(coarse, scale_factor=16, mode='bilinear', align_corners=False)[:, :, :H, :W]
       + torch.randn(B, dg * 18, H, W, device=dev, generator=g) * 0.15).contiguous()
run = lambda: ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, dg, scatter_hint=ops.DCN_SCATTER_LDS, xm_bound=bx, dy_bound=bd)
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f'{sys.argv[1] if len(sys.argv) > 1 else "default":16s} {e0.elapsed_time(e1) / 5:.2f} ms per backward call (LDS-window path, trained-like field)', flush=True)
