"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section):
590 MB (> the 256 MiB Infinity Cache) copied with 16 B/lane accesses (aligned) and with 4 B/lane accesses (misaligned view)."""
import torch

n = 20 * 128 * 180 * 320
src = torch.randn(n + 4, device='cuda')
dst = torch.empty(n + 4, device='cuda')
for _ in range(3):
    dst[:n].copy_(src[:n])          # vectorised (float4) elementwise copy: reads 4n B, writes 4n B
    torch.cuda.synchronize()
for _ in range(3):
    dst[1:n + 1].copy_(src[3:n + 3])  # misaligned: scalar 4 B/lane copy, same byte counts
    torch.cuda.synchronize()
print('calib bytes per copy', 4 * n)
