"""-m gpu: seeded random shape sweep - the two independent convolution kernels (Winograd F(2x2,3x3) vs direct implicit GEMM)
and the two weight-gradient kernels (Winograd-domain vs direct) must agree on arbitrary sizes: odd / tiny / non-multiple-of-tile
images, channel counts that are not multiples of 16 / 32 / 64, concatenated inputs, residuals, PixelShuffle, sigmoid-from.
The F(4x4,3x3) kernel (csrc/winograd_f4.hip, the dominant kernel of the clip workloads) gets its own sweep against torch's CPU
convolution in float64: both block shapes, every epilogue, ragged blocks, channel padding, the frame map."""
import ctypes
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


F4_MODES = ['plain', 'lrelu', 'relu_res1', 'relu_res1_scaled', 'lrelu_res2', 'shuffle', 'sigmoid_from', 'gate_relu', 'gate_lrelu', 'x2', 'x2_map']


def _f4_cases(n_cases, seed):
    rng = random.Random(seed)
    cases = []
    for i in range(n_cases):
        mode = F4_MODES[i % len(F4_MODES)]  # every epilogue gets the same number of shapes
        c1 = rng.choice([8, 16, 20, 32, 48, 64, 100, 128, 216])
        co = rng.choice([48, 64, 70, 96, 128, 200, 216])
        h, w = rng.randint(4, 50), 4 * rng.randint(8, 50)  # w in 32..200, a multiple of 4 (the kernel's eligibility rule)
        n = rng.randint(1, 6)
        c2 = 0
        if mode.startswith('x2'):
            c1, c2 = rng.choice([(16, 16), (64, 64), (100, 20), (32, 8), (128, 128)])  # c1 even: a staging wave's channel pair never straddles
            if mode == 'x2_map':
                n = rng.choice([2, 3]) * rng.randint(1, 2)  # t frames per clip below
        if mode == 'shuffle':
            co = co // 4 * 4
        cases.append((n, c1, c2, h, w, co, mode))
    return cases


@pytest.mark.parametrize('case', _f4_cases(44, 20260924), ids=lambda c: 'n%d_c%d+%d_%dx%d_co%d_%s' % c)
def test_winograd_f4_matches_fp64_on_random_shapes(gpu, case):
    import torch.nn.functional as F
    from edvr_amd import _lib, ops
    n, c1, c2, h, w, co, mode = case
    g = torch.Generator().manual_seed(3 * n + 7 * c1 + 13 * c2 + 101 * h + 1009 * w + 31 * co + len(mode))
    x1 = torch.randn(n, c1, h, w, generator=g)
    x2 = x2_map = None
    xin = x1
    if mode == 'x2':
        x2 = torch.randn(n, c2, h, w, generator=g)
        xin = torch.cat([x1, x2], 1)
    elif mode == 'x2_map':  # the reference frame of each clip, read in place through the image map (edvr_arch.py:392-401)
        t = 2 if n % 2 == 0 else 3
        ctr = t // 2
        x2 = torch.randn(n, c2, h, w, generator=g)
        x2_map = (t, t, ctr)
        xin = torch.cat([x1, x2[[(i // t) * t + ctr for i in range(n)]]], 1)
    wt = torch.randn(co, c1 + c2, 3, 3, generator=g) * 0.05
    b = torch.randn(co, generator=g)
    ref = F.conv2d(xin.double(), wt.double(), b.double(), 1, 1)
    kw = {}
    if mode in ('lrelu', 'lrelu_res2', 'shuffle', 'x2', 'x2_map'):
        ref, kw['act'] = F.leaky_relu(ref, 0.1), ops.ACT_LRELU
    elif mode in ('relu_res1', 'relu_res1_scaled'):
        ref, kw['act'] = F.relu(ref), ops.ACT_RELU
    elif mode == 'sigmoid_from':
        af = 2 * co // 3
        ref = torch.cat([ref[:, :af], torch.sigmoid(ref[:, af:])], 1)
        kw.update(act=ops.ACT_SIGMOID, act_from=af)
    if mode == 'relu_res1_scaled':
        kw['y_scale'] = 0.3
        ref = ref * 0.3
    if mode.startswith('gate'):
        slope = 0.0 if mode == 'gate_relu' else 0.1
        gt = torch.randn(ref.shape, generator=g).relu()
        ref = ref * torch.where(gt > 0, 1.0, slope).double()
        kw.update(gate=gt.to(gpu), gate_slope=slope)
    nres = {'relu_res1': 1, 'relu_res1_scaled': 1, 'lrelu_res2': 2}.get(mode, 0)
    for k in range(nres):
        r = torch.randn(ref.shape, generator=g)
        ref = ref + r.double()
        kw[f'res{k + 1}'] = r.to(gpu)
    if mode == 'shuffle':
        ref, kw['out_mode'] = F.pixel_shuffle(ref, 2), ops.OUT_PIXEL_SHUFFLE2
    wg = wt.to(gpu)
    wpk, wf4 = ops.pack_conv_weight(wg), ops.pack_conv_weight(wg, f4=True)
    x1g, x2g = x1.to(gpu), None if x2 is None else x2.to(gpu)
    d = _lib.ConvDesc()  # the request must reach the F(4x4) kernel, not a fallback
    d.c1, d.c2, d.n, d.h, d.w, d.co, d.ks, d.stride, d.algo = c1, c2, n, h, w, co, 3, 1, ops.CONV_WINOGRAD_F4
    d.x1, d.wpk_f4, d.out_mode, d.act = x1g.data_ptr(), wf4.data_ptr(), kw.get('out_mode', 0), kw.get('act', 0)
    if x2g is not None:
        d.x2 = x2g.data_ptr()
    buf = ctypes.create_string_buffer(96)
    _lib.lib().edvr_conv2d_kernel_name(ctypes.byref(d), buf, 96)
    assert buf.value == b'conv3x3_winograd_f4_kernel', buf.value
    y = ops.conv2d(x1g, wpk, b.to(gpu), co, 3, x2=x2g, x2_map=x2_map, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4, **kw)
    assert y.shape == ref.shape
    err = ((y.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    assert err < 3e-5, err  # tests/test_gpu_conv_f4.py: F(4x4) transforms hold coefficients up to 8
