"""DCNv2 backward (160 x 128 x 64 x 64), LDS-window path with the compare-and-swap adds: window margin 3 / 6 / 8 px (EDVR_DCN_BWD_MARGIN, read once per process)."""
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
coarse = torch.randn(B, dg * 18, H // 16 + 1, W // 16 + 1, device=dev, generator=g)
smooth = F.interpolate(coarse, scale_factor=16, mode='bilinear', align_corners=False)[:, :, :H, :W]
noise = torch.randn(B, dg * 18, H, W, device=dev, generator=g)
out = []
for sigma, sm, nz in ((2.0, 0.5, 0.15), (4.0, 0.5, 0.15), (3.0, 3.0, 0.3), (6.0, 1.0, 0.3)):
    off = (torch.randn(1, dg * 18, 1, 1, device=dev, generator=g) * sigma + smooth * sm + noise * nz).contiguous()
    run = lambda: ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, dg, scatter_hint=ops.DCN_SCATTER_LDS, xm_bound=bx, dy_bound=bd)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): run()
    e1.record(); torch.cuda.synchronize()
    out.append(f'sigma {sigma} smooth x{sm} noise {nz} (mean {off.abs().mean().item():.2f} px): {e0.elapsed_time(e1) / 4:.2f} ms')
print(f"margin {os.environ.get('EDVR_DCN_BWD_MARGIN', '3')}: " + ' | '.join(out), flush=True)
