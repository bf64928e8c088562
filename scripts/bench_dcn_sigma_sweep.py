"""DCNv2 forward / backward over a synthetic offset scale sigma in {0 (integer grid), 1, 4, 16, 64 px} - gather locality,
the fused kernel's halo classes and its global-memory slow path, the column-buffer fallback, the backward's scatter strategies, and
the `Offset abs mean ... larger than 50` warning path - with the hints the host derives from the layer's mean |offset| (functional.py)."""
import logging
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops
from edvr_amd.arch_util import warn_offset_absmean
from edvr_amd.functional import halo_hint_from_absmean, scatter_hint_from_absmean

logging.basicConfig(level=logging.WARNING, format='    [basicsr logger] %(message)s')
dev = torch.device('cuda')


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


HALO = {3: 'fused, halo 3', 7: 'fused, halo 7', -1: 'column buffer + GEMM'}
SCAT = {ops.DCN_SCATTER_STRIP: 'register-ring strip', ops.DCN_SCATTER_DEVICE: 'device atomics', ops.DCN_SCATTER_LDS: 'LDS window'}
for sigma in (0.0, 1.0, 4.0, 16.0, 64.0):
    g = torch.Generator(device=dev).manual_seed(int(sigma) + 1)
    # forward: the L1 layer of the 180x320 workloads (20 images); backward: the L1 layer of the training step (160 images of 64x64)
    for tag, (B, C, H, W) in (('fwd', (20, 128, 180, 320)), ('bwd', (160, 128, 64, 64))):
        x = torch.randn(B, C, H, W, device=dev, generator=g)
        w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
        b = torch.randn(C, device=dev, generator=g)
        off = torch.randn(B, 144, H, W, device=dev, generator=g) * sigma
        if sigma == 0.0:
            off = off.round()  # integer grid: every tap on a pixel centre
        m = torch.rand(B, 72, H, W, device=dev, generator=g)
        absmean = off.abs().mean().item()
        if tag == 'fwd':
            hint = halo_hint_from_absmean(absmean)
            ms = timed(lambda: ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8, halo_hint=hint))
            print(f'sigma {sigma:5.1f} px  mean|offset| {absmean:6.2f}  forward  {B}x{C}x{H}x{W}: {ms:8.3f} ms = {2.0 * 9 * C * C * B * H * W / ms / 1e9:6.1f} TF/s'
                  f'   [{HALO[hint]}]', flush=True)
            warn_offset_absmean(absmean)  # arch_util.py:248-253
        else:
            dy = torch.randn(B, C, H, W, device=dev, generator=g)
            hint = scatter_hint_from_absmean(absmean)
            ms = timed(lambda: ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint), reps=3)
            print(f'sigma {sigma:5.1f} px  mean|offset| {absmean:6.2f}  backward {B}x{C}x{H}x{W}: {ms:8.3f} ms = {6.0 * 9 * C * C * B * H * W / ms / 1e9:6.1f} TF/s'
                  f'   [dX by {SCAT[hint]}]', flush=True)
        del x, off, m
