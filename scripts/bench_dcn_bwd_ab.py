"""DCNv2 backward of the training layers (EDVR-L, 32 clips x 5 frames of 64x64 crops): ms per call for a scatter hint.
A/B of the fused kernel (dcn_bwd_fused.hip) against the staged path: run once as is and once with EDVR_DCN_BWD_FUSED=0.
    python scripts/bench_dcn_bwd_ab.py [sigma]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from edvr_amd import ops  # noqa: E402

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
print('EDVR_DCN_BWD_FUSED =', os.environ.get('EDVR_DCN_BWD_FUSED', '(unset: fused)'), ' sigma =', sigma)
only = os.environ.get('BENCH_ONLY')  # e.g. 'L1': that layer only, strip hint only (for counter collection)
for name, (B, C, H, W) in {'L1 160x128x64x64': (160, 128, 64, 64), 'L2 160x128x32x32': (160, 128, 32, 32), 'L3 160x128x16x16': (160, 128, 16, 16)}.items():
    if only and not name.startswith(only):
        continue
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
    off = torch.randn(B, 144, H, W, device=dev, generator=g) * sigma
    m = torch.rand(B, 72, H, W, device=dev, generator=g)
    dy = torch.randn(B, C, H, W, device=dev, generator=g)
    for hint_name, hint in (('strip', ops.DCN_SCATTER_STRIP), ('lds', ops.DCN_SCATTER_LDS))[:1 if only else 2]:
        for _ in range(3):
            ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'{name}  hint {hint_name:6s}: {ms:7.3f} ms per backward ({4.0 * B * H * W * C * C * 9 / ms / 1e9:6.1f} TF/s on its 2 GEMMs)')
