"""-m gpu: seeded random shape sweep - the two independent convolution kernels (Winograd F(2x2,3x3) vs direct implicit GEMM)
and the two weight-gradient kernels (Winograd-domain vs direct) must agree on arbitrary sizes: odd / tiny / non-multiple-of-tile
images, channel counts that are not multiples of 16 / 32 / 64, concatenated inputs, residuals, PixelShuffle, sigmoid-from."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _conv_cases(n_cases, seed):
    rng = random.Random(seed)
    cases = []
    for _ in range(n_cases):
        c1 = rng.choice([3, 8, 16, 24, 40, 64, 100, 128, 216])
        c2 = rng.choice([0, 0, 0, 16, 64, 20])
        co = rng.choice([3, 16, 48, 64, 70, 128, 216, 256])
        h, w = rng.randint(2, 40), rng.randint(2, 70)
        n = rng.randint(1, 3)
        mode = rng.choice(['plain', 'lrelu', 'relu_res1', 'lrelu_res2', 'shuffle', 'sigmoid_from'])
        cases.append((n, c1, c2, h, w, co, mode))
    return cases


@pytest.mark.parametrize('case', _conv_cases(40, 20260923), ids=lambda c: 'n%d_c%d+%d_%dx%d_co%d_%s' % c)
def test_winograd_equals_direct_on_random_shapes(gpu, case):
    from edvr_amd import ops
    n, c1, c2, h, w, co, mode = case
    if mode == 'shuffle':
        co = max(4, co // 4 * 4)
    g = torch.Generator().manual_seed(n + 7 * c1 + 13 * c2 + 101 * h + 1009 * w + 31 * co + len(mode))  # stable across runs
    x1 = torch.randn(n, c1, h, w, generator=g).to(gpu)
    x2 = torch.randn(n, c2, h, w, generator=g).to(gpu) if c2 else None
    wt = (torch.randn(co, c1 + c2, 3, 3, generator=g) * 0.05).to(gpu)
    b = torch.randn(co, generator=g).to(gpu)
    kw = {}
    if mode == 'lrelu':
        kw = dict(act=ops.ACT_LRELU)
    elif mode == 'relu_res1':
        kw = dict(act=ops.ACT_RELU, res1=torch.randn(n, co, h, w, generator=g).to(gpu))
    elif mode == 'lrelu_res2':
        kw = dict(act=ops.ACT_LRELU, res1=torch.randn(n, co, h, w, generator=g).to(gpu), res2=torch.randn(n, co, h, w, generator=g).to(gpu))
    elif mode == 'shuffle':
        kw = dict(act=ops.ACT_LRELU, out_mode=ops.OUT_PIXEL_SHUFFLE2)
    elif mode == 'sigmoid_from':
        kw = dict(act=ops.ACT_SIGMOID, act_from=2 * co // 3)
    wpk = ops.pack_conv_weight(wt)
    d = ops.conv2d(x1, wpk, b, co, 3, x2=x2, algo=ops.CONV_DIRECT, **kw)
    wg = ops.conv2d(x1, wpk, b, co, 3, x2=x2, algo=ops.CONV_WINOGRAD, **kw)  # falls back to the direct kernel where not applicable
    assert d.shape == wg.shape
    assert ((wg - d).abs().max() / d.abs().max().clamp_min(1e-30)).item() < 3e-5


def _wgrad_cases(n_cases, seed):
    rng = random.Random(seed)
    cases = []
    for _ in range(n_cases):
        c1 = rng.choice([16, 48, 64, 100, 128, 192])
        c2 = rng.choice([0, 0, 64, 24])
        co = rng.choice([3, 48, 64, 96, 128, 216])
        h, w = 2 * rng.randint(1, 16), 2 * rng.randint(1, 30)
        if rng.random() < 0.2:
            h += 1  # odd size: only the direct kernel applies, AUTO / WINOGRAD must fall back
        cases.append((rng.randint(1, 5), c1, c2, h, w, co))
    return cases


@pytest.mark.parametrize('case', _wgrad_cases(24, 7), ids=lambda c: 'n%d_c%d+%d_%dx%d_co%d' % c)
def test_winograd_wgrad_equals_direct_on_random_shapes(gpu, case):
    from edvr_amd import ops
    n, c1, c2, h, w, co = case
    g = torch.Generator().manual_seed(n + 7 * c1 + 13 * c2 + 101 * h + 1009 * w + 31 * co)
    x1 = torch.randn(n, c1, h, w, generator=g).to(gpu)
    x2 = torch.randn(n, c2, h, w, generator=g).to(gpu) if c2 else None
    dz = torch.randn(n, co, h, w, generator=g).to(gpu)
    out = {}
    for name, algo in (('direct', ops.CONV_DIRECT), ('winograd', ops.CONV_WINOGRAD), ('auto', ops.CONV_AUTO)):
        prev = ops.set_wgrad_algo(algo)
        try:
            out[name] = ops.conv2d_wgrad(x1, x2, None, dz, co, 3, 1, want_db=True)
        finally:
            ops.set_wgrad_algo(prev)
    (dw_d, db_d), (dw_w, db_w) = out['direct'], out['winograd']
    assert ((dw_w - dw_d).abs().max() / dw_d.abs().max().clamp_min(1e-30)).item() < 3e-5
    assert ((db_w - db_d).abs().max() / db_d.abs().max().clamp_min(1.0)).item() < 3e-5
    dw_a, db_a = out['auto']  # AUTO may pick the VALU kernel (co <= 4), the Winograd-domain kernel or the direct one
    assert ((dw_a - dw_d).abs().max() / dw_d.abs().max().clamp_min(1e-30)).item() < 3e-5
    assert ((db_a - db_d).abs().max() / db_d.abs().max().clamp_min(1.0)).item() < 3e-5
