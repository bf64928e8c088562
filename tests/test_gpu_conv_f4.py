"""-m gpu: the F(4x4,3x3) Winograd conv kernel (csrc/winograd_f4.hip, inference path) through the C ABI against torch's CPU conv
in fp64.  Tolerance: F(4x4) transforms hold coefficients up to 8 (F(2x2): 1), so fp32 rounding is ~1e-6 of the output scale at
128 channels instead of ~2e-7; the bound below is 3e-5 of the output scale (the path's specification is 1e-3 dB PSNR)."""
import ctypes
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL_F4 = 3e-5


def _rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


CASES = [
    # n, c1, c2, h, w, co, act, res, out_mode, x2map, gate
    (2, 64, 0, 16, 64, 64, 'lrelu', 0, 0, None, None),        # exactly one 8 x 64 block per image and channel block
    (1, 128, 0, 45, 80, 128, 'relu', 1, 0, None, None),        # ragged: 45 rows, 80 columns
    (3, 64, 0, 64, 64, 64, 'lrelu', 0, 0, None, None),        # training-crop size
    (2, 32, 0, 19, 36, 70, 'lrelu', 2, 0, None, None),        # partial channel block (70), two residuals, ragged block
    (1, 16, 16, 10, 52, 64, 'relu', 0, 0, None, None),        # concat input
    (4, 64, 64, 18, 36, 64, 'none', 0, 0, (2, 2, 1), None),   # concat input through the frame map
    (1, 48, 0, 8, 36, 128, 'lrelu', 0, 1, None, None),        # pixel-shuffle epilogue
    (1, 32, 0, 12, 36, 216, 'sigmoid_from', 0, 0, None, None),  # offset / mask conv epilogue, 216 channels
    (2, 216, 0, 12, 40, 128, 'none', 0, 0, None, None),       # 216 input channels (27 chunks: odd)
    (1, 100, 20, 8, 36, 64, 'lrelu', 1, 0, None, None),       # concat boundary inside a chunk
    (1, 20, 0, 9, 68, 48, 'none', 0, 0, None, None),          # 20 -> 24 padded input channels, 68 columns (2 blocks, second nearly empty)
    (2, 64, 0, 14, 40, 64, 'none', 0, 0, None, 0.1),          # gate (LeakyReLU backward) epilogue
    (2, 64, 0, 16, 40, 64, 'relu', 1, 0, None, None),         # residual + y_scale (set below)
    (1, 8, 0, 8, 64, 64, 'none', 0, 0, None, None),           # a single chunk per item
    (1, 32, 0, 12, 64, 64, 'lrelu', 0, 1, None, None),        # pixel shuffle on the vector path, second block row half outside
    (2, 32, 0, 10, 128, 64, 'relu', 2, 0, None, None),        # two residuals on the vector path, rows 10..15 of the second block row outside
    (1, 64, 0, 6, 64, 128, 'none', 0, 0, None, 0.0),          # gate on the vector path, 6 rows
    (2, 64, 0, 32, 32, 64, 'lrelu', 1, 0, None, None),        # 16 x 32 blocks (4 x 8 tiles): a 32-wide pyramid level, residual
    (1, 32, 0, 30, 160, 128, 'lrelu', 0, 1, None, None),      # 16 x 32 blocks: 160 wide, pixel shuffle, last block row 14 rows
    (1, 128, 0, 23, 80, 64, 'relu', 2, 0, None, None),        # 16 x 32 blocks: 80 wide (third block column half outside), two residuals
]


@pytest.mark.parametrize('case', CASES)
def test_f4_conv_matches_fp64(gpu, case):
    from edvr_amd import _lib, ops
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
    wg = wt.to(gpu)
    wpk, wf4 = ops.pack_conv_weight(wg), ops.pack_conv_weight(wg, f4=True)
    kw = dict(x2=None if x2 is None else x2.to(gpu), x2_map=x2map, act=act, act_from=act_from, res1=res[0].to(gpu) if nres > 0 else None,
              res2=res[1].to(gpu) if nres > 1 else None, out_mode=out_mode, gate=None if gt is None else gt.to(gpu),
              gate_slope=gate or 0.0, y_scale=y_scale, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4)
    # the request must reach the F(4x4) kernel (not fall back)
    d = _lib.ConvDesc()
    d.c1, d.c2, d.n, d.h, d.w, d.co, d.ks, d.stride, d.algo = c1, c2, n, h, w, co, 3, 1, ops.CONV_WINOGRAD_F4
    x1g = x1.to(gpu)
    d.x1, d.wpk_f4, d.out_mode, d.act = x1g.data_ptr(), wf4.data_ptr(), out_mode, act
    if c2:
        d.x2 = kw['x2'].data_ptr()
    buf = ctypes.create_string_buffer(96)
    _lib.lib().edvr_conv2d_kernel_name(ctypes.byref(d), buf, 96)
    assert buf.value == b'conv3x3_winograd_f4_kernel', buf.value
    y = ops.conv2d(x1g, wpk, b.to(gpu), co, 3, **kw)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel(y, ref) < RTOL_F4, _rel(y, ref)


def test_f4_is_deterministic_and_repeated_launches_agree(gpu):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 128, 36, 128, generator=g).to(gpu)
    wt = (torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(gpu)
    b = torch.randn(128, generator=g).to(gpu)
    wpk, wf4 = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4=True)
    first = ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4)
    ref = ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, algo=ops.CONV_DIRECT)
    assert _rel(first, ref.double().cpu()) < RTOL_F4
    for _ in range(20):
        again = ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4)
        assert torch.equal(again, first)


def test_f4_falls_back_where_it_does_not_apply(gpu):
    """A width that is not a multiple of 4 (rows are fetched as aligned 16-byte pieces): the request runs on F(2x2) instead."""
    from edvr_amd import _lib, ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 32, 12, 38, generator=g)
    wt = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    wg = wt.to(gpu)
    d = _lib.ConvDesc()
    d.c1, d.n, d.h, d.w, d.co, d.ks, d.stride, d.algo = 32, 1, 12, 38, 64, 3, 1, ops.CONV_WINOGRAD_F4
    wf4 = ops.pack_conv_weight(wg, f4=True)
    xg = x.to(gpu)
    d.x1, d.wpk_f4 = xg.data_ptr(), wf4.data_ptr()
    buf = ctypes.create_string_buffer(96)
    _lib.lib().edvr_conv2d_kernel_name(ctypes.byref(d), buf, 96)
    assert buf.value == b'conv3x3_winograd_kernel', buf.value
    y = ops.conv2d(xg, ops.pack_conv_weight(wg), None, 64, 3, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4)
    assert _rel(y, ref) < RTOL_F4


def test_f4_data_gradient_packing(gpu):
    """transpose_flip packing: the F(4x4) kernel as the data gradient of the stride-1 conv."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 24, 10, 16, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(40, 24, 3, 3, generator=g, dtype=torch.float64) * 0.1
    dy = torch.randn(1, 40, 10, 16, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, 1, 1).backward(dy)
    wg = wt.float().to(gpu)
    dx = ops.conv2d(dy.float().to(gpu), ops.pack_conv_weight(wg, transpose_flip=True), None, 24, 3,
                    wpk_f4=ops.pack_conv_weight(wg, transpose_flip=True, f4=True), algo=ops.CONV_WINOGRAD_F4)
    assert _rel(dx, x.grad) < RTOL_F4


@pytest.mark.parametrize('flip', [False, True])
def test_f4_packed_weights_match_the_oracle_layout(gpu, flip):
    """edvr_conv2d_pack_weight_f4_f32 (U = G g G^T in MFMA operand order) against oracle/winograd_f4_oracle.py, element by element."""
    from edvr_amd import ops
    from oracle import winograd_f4_oracle as W
    g = torch.Generator().manual_seed(9)
    wt = torch.randn(70, 20, 3, 3, generator=g) * 0.1
    got = ops.pack_conv_weight(wt.to(gpu), transpose_flip=flip, f4=True).double().cpu()
    want = W.pack_operand_order(wt, transpose_flip=flip)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 2e-7 * want.abs().max().item()


@pytest.mark.parametrize('shape', [(3, 64, 20, 64, 216), (2, 32, 9, 68, 216), (5, 128, 45, 80, 216), (1, 32, 12, 36, 96)])
def test_f4_abs_sum_epilogue(gpu, shape):
    """edvr_conv2d_desc.abs_sum: per-image sums of |y| over the first channels (conv_offset's offsets: bias added, no activation
    below act_from), accumulated in the F(4x4) kernel's epilogue - vector rows, ragged blocks, blocks below the image, several
    images per workgroup walk - against the plain reduction of the output; and the fallback to the separate kernel."""
    from edvr_amd import ops
    n, c, h, w, co = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, c, h, w, generator=g).to(gpu)
    wt = (torch.randn(co, c, 3, 3, generator=g) * 0.1).to(gpu)
    b = torch.randn(co, generator=g).to(gpu)
    wpk, wf4 = ops.pack_conv_weight(wt), ops.pack_conv_weight(wt, f4=True)
    nch = 2 * co // 3
    kw = dict(act=ops.ACT_SIGMOID, act_from=nch)
    y, sums = ops.conv2d(x, wpk, b, co, 3, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4, abs_sum_channels=nch, **kw)
    y0 = ops.conv2d(x, wpk, b, co, 3, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4, **kw)
    assert torch.equal(y, y0)  # the extra epilogue does not touch the output
    want = y0[:, :nch].double().abs().sum((1, 2, 3))
    y4 = y0[:, :nch].double().view(n, nch, h, w // 4, 4)
    want_diff = (y4[..., 1:] - y4[..., :-1]).abs().sum((1, 2, 3, 4))  # the three neighbour pairs inside every 4-pixel row piece
    assert tuple(sums.shape) == (2, n)
    assert ((sums[0].double() - want).abs() / want).max().item() < 1e-5
    # the roughness row is an ESTIMATE from up to four images spread over the batch, each scaled to stand for its neighbours
    # (ops.conv2d: an accumulator for it inside the F(4x4) kernels' staging waves would spill): images 0, step, 2 step, ... carry
    # diff(image) * n / (number of sampled images), the others 0
    step = max(1, n // 4)
    picked = list(range(0, n, step))
    for i in range(n):
        if i in picked:
            exp = want_diff[i].item() * n / len(picked)
            assert abs(sums[1, i].double().item() - exp) / exp < 1e-5
        else:
            assert sums[1, i].item() == 0
    y2, sums2 = ops.conv2d(x, wpk, b, co, 3, algo=ops.CONV_DIRECT, abs_sum_channels=nch, **kw)  # no such epilogue: separate kernel
    assert ((sums2[0].double() - y2[:, :nch].double().abs().sum((1, 2, 3))).abs() / want).max().item() < 1e-5
    y24 = y2[:, :nch].double().view(n, nch, h, w // 4, 4)
    assert ((sums2[1].double() - (y24[..., 1:] - y24[..., :-1]).abs().sum((1, 2, 3, 4))).abs() / want_diff).max().item() < 1e-5
