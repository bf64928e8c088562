"""Parity of a variant build's per-tap-window DCN forward against the zero-centred halo kernel of the SAME library (smooth field, a few
outliers): python scripts/check_dcn_variant.py   (EDVR_AMD_LIB selects the library)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(3)
worst = 0.0
for (B, C, H, W, Co, dg, sig) in [(2, 128, 21, 64, 128, 8, 4.0), (2, 64, 24, 40, 64, 8, 3.0), (1, 128, 40, 100, 40, 8, 6.0), (1, 128, 8, 32, 128, 8, 0.0)]:
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    w = torch.randn(Co, C, 3, 3, device=dev, generator=g) * 0.1
    b = torch.randn(Co, device=dev, generator=g)
    low = torch.randn(B, dg * 18, (H + 7) // 8, (W + 7) // 8, device=dev, generator=g) * sig
    off = torch.nn.functional.interpolate(low, size=(H, W), mode='bilinear', align_corners=False).contiguous()
    off[:, :, H // 2, W // 3] += 9.0
    m = torch.rand(B, dg * 9, H, W, device=dev, generator=g)
    y16 = ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, dg, halo_hint=ops.DCN_HALO_TAPWIN)
    y7 = ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, dg, halo_hint=7)
    worst = max(worst, float((y16 - y7).abs().max() / y7.abs().max()))
print('variant parity: worst relative difference', worst, 'OK' if worst < 2e-5 else 'FAIL')
