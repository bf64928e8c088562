"""Fused DCNv2 forward on the layer shapes of the headline step, HIP events (A/B of library builds via EDVR_AMD_LIB).
    python scripts/bench_dcn_fwd_ab.py [label] [hint]     hint: 16 = per-tap windows (default), 3 / 7 = zero-centred halo, -1 = columns + GEMM
BENCH_ONLY=n: only the n-th shape (counter collection)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
label = sys.argv[1] if len(sys.argv) > 1 else 'default'
HINT = int(sys.argv[2]) if len(sys.argv) > 2 else ops.DCN_HALO_TAPWIN
only = os.environ.get('BENCH_ONLY')
out = []
SHAPES = [(50, 128, 180, 320, 0.5), (50, 128, 90, 160, 0.5), (50, 128, 45, 80, 0.5), (20, 128, 180, 320, 2.0), (160, 128, 64, 64, 0.3), (16, 64, 180, 320, 0.5)]
for si, (B, C, H, W, sig) in enumerate(SHAPES):
    if only is not None and int(only) != si:
        continue
    hint = HINT
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(C, device=dev, generator=g)
    off = torch.randn(B, 144, H, W, device=dev, generator=g) * sig
    m = torch.rand(B, 72, H, W, device=dev, generator=g)
    for _ in range(3):
        ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8, halo_hint=hint)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8, halo_hint=hint)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append(f'{B}x{C}x{H}x{W} s{sig} hint {hint}: {ms:.3f} ms ({2.0 * 9 * C * C * B * H * W / ms / 1e9:.1f} TF/s)')
print(f'{label:12s} ' + ' | '.join(out), flush=True)
