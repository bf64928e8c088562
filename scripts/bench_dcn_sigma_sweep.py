"""DCNv2 forward / backward over the offset scale - gather locality, the fused kernels' LDS windows and their
global-memory slow path, the column-buffer fallback, the backward's scatter strategies, and the `Offset abs mean ... larger than
50` warning path.  Two families of offset fields:

  white   every offset element ~ N(0, sigma^2) independently (sigma = 0: the integer grid): no locality at all beyond sigma;
  piecewise  two rigidly moving regions (object boundaries): per-tap constants, a jump of 4 / 8 px across diagonal boundaries that cross
          every 8 x 32 tile - the tap-window kernels' fix-up paths;
  smooth  what a (trained) conv_offset produces: a per-channel constant ~ N(0, sigma^2) (the bias: each of the dg x 9 taps has its
          own displacement), a low-frequency motion component (N(0, 0.5^2) on a 16-px grid, bilinearly interpolated) and a white
          residual of 0.15 px (the size of the spatial part of bench.py's synthetic network).

Every forward kernel class (halo hint) and every dX strategy (scatter hint) is timed on every field; `*` marks the choice the host
derives from the layer's statistics (functional.py: mean |offset| and mean |horizontal difference|).  Flops: forward 2 * 9 C Co per
pixel, backward 4 * 9 C Co (its two GEMMs, deform_conv_cuda.cpp:623-632,659-672)."""
import logging
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from edvr_amd import ops
from edvr_amd.arch_util import warn_offset_absmean
from edvr_amd.functional import halo_hint_from_stats, scatter_hint_from_stats

logging.basicConfig(level=logging.WARNING, format='    [basicsr logger] %(message)s')
dev = torch.device('cuda')
QUICK = '--quick' in sys.argv


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


def field(kind, sigma, B, H, W, g):
    if kind == 'white':
        off = torch.randn(B, 144, H, W, device=dev, generator=g) * sigma
        return off.round() if sigma == 0.0 else off
    if kind == 'piecewise':  # two rigid regions: per-tap constants ~ N(0, 3^2), + a jump of `sigma` px on every other 24-px diagonal stripe
        base = torch.randn(1, 144, 1, 1, device=dev, generator=g) * 3.0
        step = (torch.rand(1, 144, 1, 1, device=dev, generator=g) * 2 - 1) * sigma
        yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing='ij')
        side = (((xx + yy) // 24) % 2).float().view(1, 1, H, W)
        return (base + side * step + torch.randn(B, 144, H, W, device=dev, generator=g) * 0.05).contiguous()
    bias = torch.randn(1, 144, 1, 1, device=dev, generator=g) * sigma
    coarse = torch.randn(B, 144, (H + 15) // 16 + 1, (W + 15) // 16 + 1, device=dev, generator=g) * 0.5
    low = F.interpolate(coarse, scale_factor=16, mode='bilinear', align_corners=False)[:, :, :H, :W]
    return (bias + low + torch.randn(B, 144, H, W, device=dev, generator=g) * 0.15).contiguous()


HALO = {3: 'halo 3', 7: 'halo 7', ops.DCN_HALO_TAPWIN: 'tap windows', -1: 'columns+GEMM'}
SCAT = {ops.DCN_SCATTER_STRIP: 'strip/fused', ops.DCN_SCATTER_DEVICE: 'device atomics', ops.DCN_SCATTER_LDS: 'LDS window'}
FIELDS = [('white', s) for s in (0.0, 0.5, 1.0, 4.0, 16.0, 64.0)] + [('smooth', s) for s in (0.5, 2.0, 4.0, 10.0)] + [('piecewise', s) for s in (4.0, 8.0)]
if QUICK:
    FIELDS = [('white', 0.5), ('white', 4.0), ('smooth', 0.5), ('smooth', 4.0), ('smooth', 10.0), ('piecewise', 4.0), ('piecewise', 8.0)]
for kind, sigma in FIELDS:
    g = torch.Generator(device=dev).manual_seed(int(sigma * 10) + (1000 if kind == 'smooth' else 1))
    # forward: the L1 layer of the 180x320 workloads (20 images); backward: the L1 layer of the training step (160 images of 64x64)
    for tag, (B, C, H, W) in (('fwd', (20, 128, 180, 320)), ('bwd', (160, 128, 64, 64))):
        x = torch.randn(B, C, H, W, device=dev, generator=g)
        w = torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05
        b = torch.randn(C, device=dev, generator=g)
        off = field(kind, sigma, B, H, W, g)
        m = torch.rand(B, 72, H, W, device=dev, generator=g)
        absmean = off.abs().mean().item()
        rough = (off[..., 1:] - off[..., :-1]).abs().mean().item()
        head = f'{kind:6s} sigma {sigma:5.1f}  mean|offset| {absmean:6.2f}  mean|dx offset| {rough:5.2f}'
        if tag == 'fwd':
            auto = halo_hint_from_stats(absmean, rough)
            flops = 2.0 * 9 * C * C * B * H * W
            cells = []
            for hint in (3, 7, ops.DCN_HALO_TAPWIN, -1):
                ms = timed(lambda: ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8, halo_hint=hint), reps=3 if QUICK else 5)
                cells.append(f'{"*" if hint == auto else " "}{HALO[hint]} {ms:7.3f} ms {flops / ms / 1e9:5.1f} TF/s')
            print(f'{head}  forward  {B}x{C}x{H}x{W}: ' + ' | '.join(cells), flush=True)
            warn_offset_absmean(absmean)  # arch_util.py:248-253
        else:
            dy = torch.randn(B, C, H, W, device=dev, generator=g)
            auto = scatter_hint_from_stats(absmean, rough)
            flops = 4.0 * 9 * C * C * B * H * W
            cells = []
            for hint in (ops.DCN_SCATTER_STRIP, ops.DCN_SCATTER_DEVICE, ops.DCN_SCATTER_LDS):
                if hint == ops.DCN_SCATTER_DEVICE and kind == 'white' and sigma >= 4.0 and not QUICK:
                    ms = timed(lambda: ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint), reps=1)
                else:
                    ms = timed(lambda: ops.dcnv2_backward(x, off, m, w, dy, True, 1, 1, 1, 1, 8, scatter_hint=hint), reps=3)
                cells.append(f'{"*" if hint == auto else " "}{SCAT[hint]} {ms:7.3f} ms {flops / ms / 1e9:5.1f} TF/s')
            print(f'{head}  backward {B}x{C}x{H}x{W}: ' + ' | '.join(cells), flush=True)
        del x, off, m
