"""Split-operand Winograd-domain weight gradient (csrc/winograd_wgrad_s.hip) beside the fp32 kernel on the training shapes: launch
time and error of both against torch's fp64 conv2d weight gradient (CPU, a slice of the batch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
SHAPES = [(160, 128, 64, 64, 128), (32, 128, 64, 64, 128), (160, 256, 64, 64, 128), (160, 128, 64, 64, 216), (160, 128, 32, 32, 128), (32, 64, 256, 256, 64)]
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
for (n, ci, h, w, co) in SHAPES:
    x = torch.randn(n, ci, h, w, device=dev)
    dz = torch.randn(n, co, h, w, device=dev) * 1e-3
    ops.input_bound(x); ops.input_bound(dz)
    res = {}
    for split in (False, True):
        ops.set_f4s(training=split)
        for _ in range(2):
            dw = ops.conv2d_wgrad(x, None, None, dz, co, 3, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dw = ops.conv2d_wgrad(x, None, None, dz, co, 3, 1)
        e1.record()
        torch.cuda.synchronize()
        res[split] = (e0.elapsed_time(e1) / 5, dw)
    nn = min(n, 4)
    ref = torch.nn.grad.conv2d_weight(x[:nn].double().cpu(), (co, ci, 3, 3), dz[:nn].double().cpu(), padding=1)
    errs = {}
    for split in (False, True):
        ops.set_f4s(training=split)
        d4 = ops.conv2d_wgrad(x[:nn].contiguous(), None, None, dz[:nn].contiguous(), co, 3, 1)
        errs[split] = ((d4.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * n * h * w * co * ci * 9
    print(f'wgrad n={n:3d} ci={ci:4d} {h}x{w} co={co:3d}: fp32 {res[False][0]:7.3f} ms ({fl / res[False][0] / 1e9:6.1f} TF/s) | split {res[True][0]:7.3f} ms '
          f'({fl / res[True][0] / 1e9:6.1f} TF/s, {res[False][0] / res[True][0]:.2f}x) | err vs fp64 (4 images): fp32 {errs[False]:.2e} split {errs[True]:.2e} | '
          f'split vs fp32 on the batch: {((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max()).item():.2e}', flush=True)
