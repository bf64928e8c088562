"""dW = sum dY col^T of the DCN backward: the split-operand product (csrc/gemm_nt_s.hip) beside the fp32 kernel, whole backward calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
for (B, C, H, W) in [(160, 128, 64, 64), (160, 128, 32, 32), (40, 128, 64, 64)]:
    dg = 8
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
    off = torch.randn(B, dg * 18, H, W, device=dev, generator=g) * 0.3
    m = torch.rand(B, dg * 9, H, W, device=dev, generator=g)
    dy = torch.randn(B, C, H, W, device=dev, generator=g) * 1e-3
    bx, bd = ops.amax(x), ops.amax(dy)
    res = {}
    for name, kw in (('fp32', {}), ('split', {'xm_bound': bx, 'dy_bound': bd})):
        per = {}
        def hook(nm, flops, launch, *a):
            launch()
        run = lambda: ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, dg, scatter_hint=ops.DCN_SCATTER_STRIP, **kw)
        for _ in range(2):
            out = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 5, out[3].clone())
    d = ((res['split'][1] - res['fp32'][1]).abs().max() / res['fp32'][1].abs().max()).item()
    fl = 2.0 * 9 * C * C * B * H * W
    print(f'dcn bwd {B}x{C}x{H}x{W}: whole backward fp32 dW {res["fp32"][0]:.3f} ms | split dW {res["split"][0]:.3f} ms | saved {res["fp32"][0] - res["split"][0]:.3f} ms '
          f'(the product alone at 100 TF/s: {fl / 1e11:.2f} ms) | dW split vs fp32 {d:.2e}', flush=True)
