"""Split F(4x4) kernel: does the residual epilogue cost time (its rows are requested ONE output phase ahead)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
for n, c, h, w in [(50, 128, 180, 320), (10, 128, 180, 320), (32, 128, 64, 64)]:
    x = torch.randn(n, c, h, w, device=dev); wt = torch.randn(c, c, 3, 3, device=dev) * 0.05; b = torch.randn(c, device=dev)
    res = torch.randn(n, c, h, w, device=dev); res2 = torch.randn(n, c, h, w, device=dev)
    wpk, wf4s = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4s=True)
    bound = ops.amax(x)
    y = torch.empty(n, c, h, w, device=dev)
    out = []
    for tag, kw in [('lrelu', dict(act=ops.ACT_LRELU)), ('res1', dict(res1=res)), ('res1+res2', dict(res1=res, res2=res2)), ('relu', dict(act=ops.ACT_RELU)), ('lrelu again', dict(act=ops.ACT_LRELU))]:
        run = lambda: ops.conv2d(x, wpk, b, c, 3, wpk_f4s=wf4s, x_amax=bound, algo=ops.CONV_WINOGRAD_F4S, out=y, **kw)
        for _ in range(4): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append(f'{tag} {e0.elapsed_time(e1) / 20:.3f}')
    print(f'{n}x{c}x{h}x{w}: ' + ' | '.join(out) + ' ms', flush=True)
