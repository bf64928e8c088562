"""Split-operand F(4x4) kernel (csrc/winograd_f4s.hip) beside the fp32 F(4x4) kernel on the layer shapes of the EDVR-L step: average launch
time (HIP events on the launch stream, 20 launches) and the error of both against an fp64 convolution of one image on the CPU.
    python scripts/bench_f4s.py [label]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from edvr_amd import ops
dev = torch.device('cuda')
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('EDVR_AMD_LIB', 'default'))
SHAPES = [(50, 128, 180, 320, 128), (10, 128, 180, 320, 128), (160, 128, 64, 64, 128), (20, 64, 180, 320, 64), (5, 128, 720, 1280, 128)]
if os.environ.get('F4S_SHAPES'):
    SHAPES = [tuple(int(v) for v in s.split('x')) for s in os.environ['F4S_SHAPES'].split(',')]


def timed(run, reps=20):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f'# {label}', flush=True)
for n, c, h, w, co in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, c, h, w, generator=g).to(dev) if n * c * h * w < 2 ** 28 else torch.randn(n, c, h, w, device=dev)
    wt = (torch.randn(co, c, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    wpk, wf4, wf4s = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4=True), ops.pack_conv_weight(wt, f4s=True)
    y = torch.empty(n, co, h, w, device=dev)
    bound = ops.amax(x)
    run32 = lambda: ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4, out=y)
    runs = lambda: ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, x_amax=bound, algo=ops.CONV_WINOGRAD_F4S, out=y)
    runs_amax = lambda: ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, algo=ops.CONV_WINOGRAD_F4S, out=y)
    ms32, mss, mssa = timed(run32), timed(runs), timed(runs_amax)
    # error of both on image 0 against fp64 (rows 0..47 only for the big images: the CPU conv is the slow part)
    hh = min(h, 48)
    ref = F.leaky_relu(F.conv2d(x[:1, :, :hh + 1].double().cpu(), wt.double().cpu(), b.double().cpu(), 1, 1), 0.1)[:, :, :hh - 1]
    run32()
    e32 = ((y[:1, :, :hh - 1].double().cpu() - ref).abs().max() / ref.abs().max()).item()
    runs()
    es = ((y[:1, :, :hh - 1].double().cpu() - ref).abs().max() / ref.abs().max()).item()
    alg = 2.0 * 9 * c * co * n * h * w
    gb = 4.0 * n * h * w * (c + co) / 1e9
    print(f'{n}x{c}x{h}x{w}->{co}: fp32 {ms32:.3f} ms | split {mss:.3f} ms ({ms32 / mss:.2f}x; {alg / mss / 1e9:.0f} TF/s algorithmic, '
          f'{gb / mss * 1e3 / 1e3:.2f} TB/s of algorithmic bytes) | split + amax pass {mssa:.3f} ms | max err vs fp64: fp32 {e32:.2e} split {es:.2e}', flush=True)
    del x, y
