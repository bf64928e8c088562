"""DCNv2 backward micro-benchmark on the training shapes (run on the GPU box); EDVR_DCN_BWD_TILE=0 / EDVR_DCN_BLAS=0 for A/B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
for (B, C, H, W, sigma) in [(160, 128, 64, 64, 1.0), (160, 128, 32, 32, 1.0), (160, 128, 64, 64, 4.0), (20, 128, 180, 320, 1.0)]:
    x = torch.randn(B, C, H, W, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    off = torch.randn(B, 144, H, W, device=dev) * sigma; m = torch.rand(B, 72, H, W, device=dev); dy = torch.randn(B, C, H, W, device=dev)
    for _ in range(2): r = ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8)
    e1.record(); torch.cuda.synchronize()
    print(f'dcn bwd B={B} C={C} {H}x{W} sigma={sigma}: {e0.elapsed_time(e1) / 5:7.3f} ms  tile={os.environ.get("EDVR_DCN_BWD_TILE", "1")} blas={os.environ.get("EDVR_DCN_BLAS", "1")}', flush=True)
