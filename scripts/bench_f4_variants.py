"""Average launch time of the F(4x4) kernel on a few layer shapes, HIP events on the launch stream.  One library per process:
    EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_X.so python scripts/bench_f4_variants.py [label]
(ablation builds compute wrong results on purpose; this script only times)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('EDVR_AMD_LIB', 'default'))
SHAPES = [(20, 128, 180, 320, 128), (50, 128, 180, 320, 128), (160, 128, 64, 64, 128), (5, 128, 720, 1280, 128)]
out = []
for n, c, h, w, co in SHAPES:
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(co, c, 3, 3, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    wpk, wf4 = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4=True)
    y = torch.empty(n, co, h, w, device=dev)
    run = lambda: ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4, out=y)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * 9 * c * co * n * h * w / 4.0 / ms / 1e9
    out.append(f'{n}x{c}x{h}x{w}: {ms:.3f} ms ({tf:.1f} TF/s exec)')
    del x, y
print(f'{label:28s} ' + ' | '.join(out), flush=True)
