"""DCNv2 forward micro-benchmark (fused vs EDVR_DCN_FUSED=0 generic path), run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
dev = torch.device('cuda')
for (B, C, H, W, sigma) in [(20, 128, 180, 320, 0.3), (20, 128, 180, 320, 1.0), (20, 64, 180, 320, 1.0), (20, 128, 90, 160, 1.0), (20, 128, 180, 320, 4.0), (160, 128, 64, 64, 1.0)]:
    x = torch.randn(B, C, H, W, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05; b = torch.randn(C, device=dev)
    off = torch.randn(B, 144, H, W, device=dev) * sigma; m = torch.rand(B, 72, H, W, device=dev)
    for _ in range(2): y = ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'dcn fwd B={B} C={C} {H}x{W} sigma={sigma}: {ms:7.3f} ms  {2.0*B*H*W*C*C*9/ms/1e9:7.2f} TF/s  fused={os.environ.get("EDVR_DCN_FUSED","1")}', flush=True)
