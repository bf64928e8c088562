"""Streaming 1x1 conv: the split-operand kernel (csrc/conv1x1_s.hip) beside the fp32 one on TSA's feat_fusion shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
for (n, ci, h, w, co) in [(10, 640, 180, 320, 128), (8, 896, 180, 320, 128), (4, 640, 720, 1280, 128), (32, 640, 64, 64, 128)]:
    x = torch.randn(n, ci, h, w, device=dev, generator=g)
    wt = torch.randn(co, ci, 1, 1, device=dev, generator=g) * 0.05
    b = torch.randn(co, device=dev, generator=g)
    wpk, wq = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4s=True)
    bound = ops.amax(x)
    y = torch.empty(n, co, h, w, device=dev)
    res = {}
    for name, kw in (('fp32', {}), ('split', {'wpk_f4s': wq, 'x_amax': bound})):
        run = lambda: ops.conv2d(x, wpk, b, co, 1, act=ops.ACT_LRELU, out=y, **kw)
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
    fl, nb = 2.0 * ci * co * n * h * w, 4.0 * n * h * w * (ci + co)
    d = ((res['split'][1] - res['fp32'][1]).abs().max() / res['fp32'][1].abs().max()).item()
    print(f'conv1x1 {n}x{ci}x{h}x{w} -> {co}: fp32 {res["fp32"][0]:.3f} ms ({fl / res["fp32"][0] / 1e9:.1f} TF/s, {nb / res["fp32"][0] / 1e6:.0f} GB/s) | '
          f'split {res["split"][0]:.3f} ms ({fl / res["split"][0] / 1e9:.1f} TF/s, {nb / res["split"][0] / 1e6:.0f} GB/s, {res["fp32"][0] / res["split"][0]:.2f}x) | split vs fp32 {d:.2e}', flush=True)
