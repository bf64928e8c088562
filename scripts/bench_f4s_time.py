"""Launch time of the split-operand F(4x4) kernel only (ablation builds compute wrong results on purpose; this script only times):
    EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_X.so python scripts/bench_f4s_time.py [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('EDVR_AMD_LIB', 'default'))
out = []
for n, c, h, w, co in [(50, 128, 180, 320, 128), (10, 128, 180, 320, 128), (20, 64, 180, 320, 64)]:
    x = torch.randn(n, c, h, w, device=dev); wt = torch.randn(co, c, 3, 3, device=dev) * 0.05; b = torch.randn(co, device=dev)
    wpk, wf4s = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4s=True)
    bound = ops.amax(x)
    y = torch.empty(n, co, h, w, device=dev)
    run = lambda: ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, x_amax=bound, algo=ops.CONV_WINOGRAD_F4S, out=y)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    out.append(f'{n}x{c}x{h}x{w}: {e0.elapsed_time(e1) / 20:.3f} ms')
    del x, y
print(f'{label:36s} ' + ' | '.join(out), flush=True)
