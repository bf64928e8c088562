import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
n, ci = 20, 128
x = torch.randn(n, ci, 180, 320, device=dev); wt = torch.randn(128, ci, 3, 3, device=dev) * 0.05
b = torch.randn(128, device=dev); res = torch.randn(n, 128, 180, 320, device=dev)
wpk = ops.pack_conv_weight(wt)
def run(tag, **kw):
    for _ in range(2): ops.conv2d(x, wpk, kw.pop('bias', None), 128, 3, **kw) if False else None
for tag, bias, act, r in [('none', None, 0, None), ('bias', b, 0, None), ('lrelu', None, 2, None), ('relu', None, 1, None), ('bias+lrelu', b, 2, None), ('res', None, 0, res), ('bias+res', b, 0, res)]:
    for _ in range(2): ops.conv2d(x, wpk, bias, 128, 3, act=act, res1=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv2d(x, wpk, bias, 128, 3, act=act, res1=r)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'{tag:12s}: {ms:7.3f} ms  {2.0*n*57600*128*ci*9/ms/1e9:7.2f} TF/s', flush=True)
