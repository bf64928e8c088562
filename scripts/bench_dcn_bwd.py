"""DCNv2 backward micro-benchmark on the training shapes (run on the GPU box) for each dX accumulation strategy
(EDVR_DCN_SCATTER_*) and offset scales from fresh-model (sub-pixel) to white noise; EDVR_DCN_BLAS=0 for the GEMM A/B."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edvr_amd import ops  # noqa: E402

dev = torch.device('cuda')
HINTS = {'device': ops.DCN_SCATTER_DEVICE, 'lds': ops.DCN_SCATTER_LDS, 'strip': ops.DCN_SCATTER_STRIP}
shapes = [(160, 128, 64, 64, 0.2), (160, 128, 64, 64, 0.5), (160, 128, 64, 64, 1.0), (160, 128, 32, 32, 0.2), (160, 128, 64, 64, 4.0)]
for (B, C, H, W, sigma) in shapes:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    off = torch.randn(B, 144, H, W, device=dev) * sigma
    m = torch.rand(B, 72, H, W, device=dev)
    dy = torch.randn(B, C, H, W, device=dev)
    ref = None
    for name, hint in HINTS.items():
        for _ in range(2):
            r = ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            r = ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = r[0]
        err = ((r[0] - ref).abs().max() / ref.abs().max()).item()
        print(f'dcn bwd B={B} C={C} {H}x{W} sigma={sigma} {name:6s}: {e0.elapsed_time(e1) / 4:7.3f} ms   dx vs device: {err:.1e}', flush=True)
