"""-m gpu: the split-operand F(4x4,3x3) Winograd kernel (csrc/winograd_f4s.hip: fp32 operands as f16 (hi, lo) pairs on the f16
matrix pipe, all four cross products, fp32 accumulation) through the C ABI against torch's CPU conv in fp64 - on every case of the
fp32 F(4x4) kernel's test at THE SAME tolerance (3e-5 of the output scale), and against that kernel's own error on the same input:
the split may not be the less accurate of the two by more than rounding noise."""
import ctypes
import zlib

import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv_f4 import CASES, RTOL_F4, _rel

pytestmark = pytest.mark.gpu


def _case_tensors(case):
    n, c1, c2, h, w, co, actn, nres, out_mode, x2map, gate = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))  # (hash() of a tuple holding strings changes from process to process)
    x1 = torch.randn(n, c1, h, w, generator=g)
    n2 = n if x2map is None else (n // x2map[0]) * x2map[1]
    x2 = torch.randn(n2, c2, h, w, generator=g) if c2 else None
    wt = torch.randn(co, c1 + c2, 3, 3, generator=g) * 0.1
    b = torch.randn(co, generator=g)
    if x2 is None:
        xin = x1
    elif x2map is None:
        xin = torch.cat([x1, x2], 1)
    else:
        idx = [(i // x2map[0]) * x2map[1] + x2map[2] for i in range(n)]
        xin = torch.cat([x1, x2[idx]], 1)
    ref = F.conv2d(xin.double(), wt.double(), b.double(), 1, 1)
    act, act_from = {'none': (0, 0), 'relu': (1, 0), 'lrelu': (2, 0), 'sigmoid_from': (3, 2 * co // 3)}[actn]
    if actn == 'relu':
        ref = F.relu(ref)
    elif actn == 'lrelu':
        ref = F.leaky_relu(ref, 0.1)
    elif actn == 'sigmoid_from':
        ref = torch.cat([ref[:, :act_from], torch.sigmoid(ref[:, act_from:])], 1)
    y_scale = 0.25 if (nres == 1 and actn == 'relu' and h == 16) else 1.0
    ref = ref * y_scale
    gt = None
    if gate is not None:
        gt = torch.randn(ref.shape, generator=g).relu()
        ref = ref * torch.where(gt > 0, 1.0, gate).double()
    res = [torch.randn(ref.shape, generator=g) for _ in range(nres)]
    for r in res:
        ref = ref + r.double()
    if out_mode == 1:
        ref = F.pixel_shuffle(ref, 2)
    return x1, x2, wt, b, ref, act, act_from, y_scale, gt, res


@pytest.mark.parametrize('case', CASES)
def test_f4s_conv_matches_fp64(gpu, case):
    from edvr_amd import _lib, ops
    n, c1, c2, h, w, co, actn, nres, out_mode, x2map, gate = case
    x1, x2, wt, b, ref, act, act_from, y_scale, gt, res = _case_tensors(case)
    wg = wt.to(gpu)
    wpk, wf4, wf4s = ops.pack_conv_weight(wg), ops.pack_conv_weight(wg, f4=True), ops.pack_conv_weight(wg, f4s=True)
    x1g = x1.to(gpu)
    kw = dict(x2=None if x2 is None else x2.to(gpu), x2_map=x2map, act=act, act_from=act_from, res1=res[0].to(gpu) if nres > 0 else None,
              res2=res[1].to(gpu) if nres > 1 else None, out_mode=out_mode, gate=None if gt is None else gt.to(gpu),
              gate_slope=gate or 0.0, y_scale=y_scale, wpk_f4=wf4)
    # the request must reach the split-operand kernel (not fall back)
    d = _lib.ConvDesc()
    d.c1, d.c2, d.n, d.h, d.w, d.co, d.ks, d.stride, d.algo = c1, c2, n, h, w, co, 3, 1, ops.CONV_WINOGRAD_F4S
    bound = ops.amax(x1g)
    d.x1, d.wpk_f4, d.wpk_f4s, d.x_amax, d.out_mode, d.act = x1g.data_ptr(), wf4.data_ptr(), wf4s.data_ptr(), bound.data_ptr(), out_mode, act
    if c2:
        d.x2 = kw['x2'].data_ptr()
    buf = ctypes.create_string_buffer(96)
    _lib.lib().edvr_conv2d_kernel_name(ctypes.byref(d), buf, 96)
    assert buf.value == b'conv3x3_winograd_f4s_kernel', buf.value
    y = ops.conv2d(x1g, wpk, b.to(gpu), co, 3, wpk_f4s=wf4s, algo=ops.CONV_WINOGRAD_F4S, **kw)
    y32 = ops.conv2d(x1g, wpk, b.to(gpu), co, 3, algo=ops.CONV_WINOGRAD_F4, **kw)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    e_split, e_f32 = _rel(y, ref), _rel(y32, ref)
    assert e_split < RTOL_F4, e_split
    assert e_split < 1.5 * e_f32 + 2e-7, (e_split, e_f32)  # not the less accurate of the two (both are ~1e-6 at 128 channels)


@pytest.mark.parametrize('scale', [1e-25, 1e-12, 1e-4, 1.0, 1e4, 1e12, 1e25])
def test_f4s_is_scale_invariant(gpu, scale):
    """The power-of-two operand scales follow the tensors' magnitudes: inputs of 1e-25 .. 1e25 (the transformed input of the largest
    would overflow f16 65504 by 24 orders of magnitude unscaled) give the same RELATIVE error."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 16, 64, generator=g) * scale
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.1 / (scale ** 0.5 if scale > 1 else 1.0)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    wg = wt.to(gpu)
    y = ops.conv2d(x.to(gpu), ops.pack_conv_weight(wg), None, 64, 3, wpk_f4s=ops.pack_conv_weight(wg, f4s=True), algo=ops.CONV_WINOGRAD_F4S)
    assert torch.isfinite(y).all()
    assert _rel(y, ref) < RTOL_F4, _rel(y, ref)


def test_f4s_loose_bound_and_outliers(gpu):
    """`x_amax` is only a bound: 2^10 too large costs nothing measurable; one outlier 1000x the rest leaves the small elements their
    absolute accuracy (error measured against the output scale WITHOUT the outlier's neighbourhood)."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 64, 16, 64, generator=g)
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.1
    wg, xg = wt.to(gpu), x.to(gpu)
    wpk, wf4s = ops.pack_conv_weight(wg), ops.pack_conv_weight(wg, f4s=True)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    loose = ops.amax(xg) * 1024.0
    y = ops.conv2d(xg, wpk, None, 64, 3, wpk_f4s=wf4s, x_amax=loose, algo=ops.CONV_WINOGRAD_F4S)
    assert _rel(y, ref) < RTOL_F4, _rel(y, ref)
    x[0, 5, 3, 7] = 1000.0
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    y = ops.conv2d(x.to(gpu), wpk, None, 64, 3, wpk_f4s=wf4s, algo=ops.CONV_WINOGRAD_F4S)
    far = torch.ones(16, 64, dtype=torch.bool)
    far[0:8, 0:16] = False  # tiles whose patches contain the outlier
    err = (y.double().cpu() - ref)[0, :, far].abs().max().item() / ref[0, :, far].abs().max().item()
    assert err < RTOL_F4, err
    assert _rel(y, ref) < RTOL_F4


def test_f4s_zero_input_and_zero_weights(gpu):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(13)
    wt = (torch.randn(64, 32, 3, 3, generator=g) * 0.1).to(gpu)
    b = torch.randn(64, generator=g).to(gpu)
    x = torch.zeros(1, 32, 8, 64, device=gpu)
    y = ops.conv2d(x, ops.pack_conv_weight(wt), b, 64, 3, wpk_f4s=ops.pack_conv_weight(wt, f4s=True), algo=ops.CONV_WINOGRAD_F4S)
    assert torch.equal(y, b.view(1, -1, 1, 1).expand_as(y))
    w0 = torch.zeros_like(wt)
    x = torch.randn(1, 32, 8, 64, generator=g).to(gpu)
    y = ops.conv2d(x, ops.pack_conv_weight(w0), b, 64, 3, wpk_f4s=ops.pack_conv_weight(w0, f4s=True), algo=ops.CONV_WINOGRAD_F4S)
    assert torch.equal(y, b.view(1, -1, 1, 1).expand_as(y))


def test_f4s_is_deterministic(gpu):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 128, 36, 128, generator=g).to(gpu)
    wt = (torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(gpu)
    b = torch.randn(128, generator=g).to(gpu)
    wpk, wf4s = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4s=True)
    first = ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, algo=ops.CONV_WINOGRAD_F4S)
    ref = ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, algo=ops.CONV_DIRECT)
    assert _rel(first, ref.double().cpu()) < RTOL_F4
    for _ in range(10):
        assert torch.equal(ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, algo=ops.CONV_WINOGRAD_F4S), first)


def test_f4s_data_gradient_packing(gpu):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 24, 10, 16, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(40, 24, 3, 3, generator=g, dtype=torch.float64) * 0.1
    dy = torch.randn(1, 40, 10, 16, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, 1, 1).backward(dy)
    wg = wt.float().to(gpu)
    dx = ops.conv2d(dy.float().to(gpu), ops.pack_conv_weight(wg, transpose_flip=True), None, 24, 3,
                    wpk_f4s=ops.pack_conv_weight(wg, transpose_flip=True, f4s=True), algo=ops.CONV_WINOGRAD_F4S)
    assert _rel(dx, x.grad) < RTOL_F4


def test_amax_kernel(gpu):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(14)
    for shape in [(3, 5, 7, 9), (2, 64, 45, 80), (1, 1, 1, 3)]:
        x = torch.randn(*shape, generator=g).to(gpu)
        assert ops.amax(x).item() == x.abs().max().item()
    x = torch.randn(4, 6, 10, 12, generator=g).to(gpu)
    v = x[:, 1:4]  # channel slice: images 6 planes apart
    assert ops.amax(v).item() == v.abs().max().item()
    both = ops.amax(x[:2])
    ops.amax(x[2:] * 3, out=both)
    assert both.item() == max(x[:2].abs().max().item(), (x[2:] * 3).abs().max().item())


# ---- the streaming 1x1 kernel's split form (csrc/conv1x1_s.hip): TSA's feat_fusion shapes and ragged ones
C1X1_CASES = [
    # n, c1, c2, h, w, co, act, residuals, x2_map
    (2, 640, 0, 20, 36, 128, 'lrelu', 0, None),      # feat_fusion (edvr_arch.py:190-193): t * c -> c
    (1, 896, 0, 17, 23, 128, 'none', 0, None),       # seven frames, ragged pixel count (391 = 3 blocks + 7)
    (3, 320, 320, 9, 40, 96, 'relu', 1, None),       # two inputs, three output tiles
    (4, 384, 128, 8, 16, 200, 'lrelu', 2, (2, 1, 0)),  # x2 broadcast through the image map, one full block + three tiles
    (1, 328, 0, 5, 7, 33, 'sigmoid_from', 0, None),  # channel count not a multiple of the 64-channel slab, two ragged tiles
]


def _c1x1_tensors(case):
    n, c1, c2, h, w, co, actn, nres, x2map = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))  # (hash() of a tuple holding strings changes from process to process)
    x1 = torch.randn(n, c1, h, w, generator=g)
    n2 = n if x2map is None else (n // x2map[0]) * x2map[1]
    x2 = torch.randn(n2, c2, h, w, generator=g) if c2 else None
    wt = torch.randn(co, c1 + c2, 1, 1, generator=g) * 0.05
    b = torch.randn(co, generator=g)
    if x2 is None:
        xin = x1
    elif x2map is None:
        xin = torch.cat([x1, x2], 1)
    else:
        xin = torch.cat([x1, x2[[(i // x2map[0]) * x2map[1] + x2map[2] for i in range(n)]]], 1)
    ref = F.conv2d(xin.double(), wt.double(), b.double())
    act, act_from = {'none': (0, 0), 'relu': (1, 0), 'lrelu': (2, 0), 'sigmoid_from': (3, 2 * co // 3)}[actn]
    if actn == 'relu':
        ref = F.relu(ref)
    elif actn == 'lrelu':
        ref = F.leaky_relu(ref, 0.1)
    elif actn == 'sigmoid_from':
        ref = torch.cat([ref[:, :act_from], torch.sigmoid(ref[:, act_from:])], 1)
    res = [torch.randn(ref.shape, generator=g) for _ in range(nres)]
    for r in res:
        ref = ref + r.double()
    return x1, x2, wt, b, ref, act, act_from, res


@pytest.mark.parametrize('case', C1X1_CASES)
def test_split_conv1x1_matches_fp64(gpu, case):
    from edvr_amd import ops
    n, c1, c2, h, w, co, actn, nres, x2map = case
    x1, x2, wt, b, ref, act, act_from, res = _c1x1_tensors(case)
    wg = wt.to(gpu)
    wpk, wq = ops.pack_conv_weight(wg), ops.pack_conv_weight(wg, f4s=True)
    x1g, x2g = x1.to(gpu), (x2.to(gpu) if x2 is not None else None)
    rg = [r.to(gpu) for r in res]
    kw = dict(x2=x2g, x2_map=x2map, act=act, act_from=act_from, res1=rg[0] if nres > 0 else None, res2=rg[1] if nres > 1 else None)
    seen = []
    ops.LAUNCH_HOOK = lambda name, flops, launch, *a: (seen.append(name), launch())
    try:
        y32 = ops.conv2d(x1g, wpk, b.to(gpu), co, 1, **kw)
        ys = ops.conv2d(x1g, wpk, b.to(gpu), co, 1, wpk_f4s=wq, **kw)
    finally:
        ops.LAUNCH_HOOK = None
    assert [k for k in seen if k.startswith('conv')] == ['conv1x1_stream_kernel', 'conv1x1_split_kernel'], seen
    e32, es = _rel(y32, ref), _rel(ys, ref)
    assert es < 2e-6 and es < 1.5 * e32 + 2e-7, (es, e32)
    bound = ops.get_bound(ys)  # the epilogue's max |y|
    assert bound is not None and abs(bound.item() - ys.abs().max().item()) <= 1e-6 * ys.abs().max().item()


@pytest.mark.parametrize('scale', [1e-25, 1e-4, 1.0, 1e6, 1e25])
def test_split_conv1x1_is_scale_invariant(gpu, scale):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 640, 12, 20, generator=g) * scale
    x[0, 3, 2, 2] *= 300.0  # an outlier sets the bound; everything else sits 2^-8 below it
    wt = torch.randn(128, 640, 1, 1, generator=g) * 0.05
    ref = F.conv2d(x.double(), wt.double())
    xg, wg = x.to(gpu), wt.to(gpu)
    y = ops.conv2d(xg, ops.pack_conv_weight(wg), None, 128, 1, wpk_f4s=ops.pack_conv_weight(wg, f4s=True), x_amax=ops.amax(xg) * 64.0)  # a loose bound is a bound
    assert torch.isfinite(y).all() and _rel(y, ref) < 2e-6, _rel(y, ref)


def test_split_conv1x1_falls_back_where_it_does_not_apply(gpu):
    """Fewer than 320 input channels, channel counts that are not multiples of 8, EDVR_CONV_DIRECT: the fp32 kernels, same results."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(6)
    for (ci, algo) in [(128, None), (324, None), (640, ops.CONV_DIRECT)]:
        x = torch.randn(1, ci, 8, 16, generator=g).to(gpu)
        wg = (torch.randn(64, ci, 1, 1, generator=g) * 0.05).to(gpu)
        ref = F.conv2d(x.double().cpu(), wg.double().cpu())
        seen = []
        ops.LAUNCH_HOOK = lambda name, flops, launch, *a: (seen.append(name), launch())
        try:
            y = ops.conv2d(x, ops.pack_conv_weight(wg), None, 64, 1, wpk_f4s=ops.pack_conv_weight(wg, f4s=True), algo=algo)
        finally:
            ops.LAUNCH_HOOK = None
        assert 'conv1x1_split_kernel' not in seen, (ci, algo, seen)
        assert _rel(y, ref) < 2e-6
