"""Split-operand tap-window DCNv2 forward (csrc/dcn_tapwin_s.hip) beside the fp32 kernel on the layer shapes of the steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
for (B, C, H, W) in [(50, 128, 180, 320), (50, 128, 90, 160), (50, 128, 45, 80), (160, 128, 64, 64), (20, 64, 180, 320)]:
    dg = 8
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(C, device=dev, generator=g)
    bias = torch.randn(1, dg * 18, 1, 1, device=dev, generator=g) * 4.0
    coarse = torch.randn(B, dg * 18, (H + 15) // 16 + 1, (W + 15) // 16 + 1, device=dev, generator=g) * 0.5
    off = (bias + F.interpolate(coarse, scale_factor=16, mode='bilinear', align_corners=False)[:, :, :H, :W] + torch.randn(B, dg * 18, H, W, device=dev, generator=g) * 0.15).contiguous()
    m = torch.rand(B, dg * 9, H, W, device=dev, generator=g)
    bound = ops.amax(x)
    y = torch.empty(B, C, H, W, device=dev)
    res = {}
    for name, kw in (('fp32', {}), ('split', {'xm_bound': bound})):
        run = lambda: ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, dg, act=ops.ACT_LRELU, halo_hint=ops.DCN_HALO_TAPWIN, out=y, **kw)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 10, y.clone())
    fl = 2.0 * 9 * C * C * B * H * W
    d = ((res['split'][1] - res['fp32'][1]).abs().max() / res['fp32'][1].abs().max()).item()
    print(f'dcn fwd {B}x{C}x{H}x{W}: fp32 {res["fp32"][0]:.3f} ms ({fl / res["fp32"][0] / 1e9:.1f} TF/s) | split {res["split"][0]:.3f} ms '
          f'({fl / res["split"][0] / 1e9:.1f} TF/s, {res["fp32"][0] / res["split"][0]:.2f}x) | split vs fp32 {d:.2e}', flush=True)
