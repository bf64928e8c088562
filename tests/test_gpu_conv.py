"""-m gpu: the fp32 MFMA conv kernel (through the C ABI) against torch's CPU conv in fp64."""
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# fp32 MFMA is a k-ordered fmaf chain: error ~1e-7 * sum|a*b|; tolerance relative to the output scale
RTOL = 2e-5


def _rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


CASES = [
    # n, c1, c2, h, w, co, ks, stride, act, res, out_mode
    (2, 64, 0, 16, 16, 64, 3, 1, 'lrelu', 0, 0),
    (1, 128, 0, 45, 80, 128, 3, 1, 'relu', 1, 0),
    (2, 3, 0, 20, 36, 64, 3, 1, 'lrelu', 0, 0),
    (1, 64, 64, 23, 41, 64, 3, 1, 'lrelu', 0, 0),
    (2, 64, 0, 32, 32, 64, 3, 2, 'lrelu', 0, 0),
    (1, 128, 0, 45, 81, 128, 3, 2, 'none', 0, 0),
    (1, 320, 0, 24, 40, 64, 1, 1, 'lrelu', 0, 0),
    (1, 64, 0, 16, 48, 216, 3, 1, 'sigmoid_from', 0, 0),
    (1, 64, 0, 12, 20, 256, 3, 1, 'lrelu', 0, 1),
    (1, 64, 0, 40, 72, 3, 3, 1, 'none', 2, 0),
    (3, 40, 24, 9, 7, 48, 3, 1, 'none', 2, 0),
    (1, 32, 0, 16, 16, 108, 3, 1, 'sigmoid_from', 0, 0),   # 97..127 output channels: 4-tile tail launch
    (1, 48, 0, 10, 12, 100, 1, 1, 'none', 0, 0),
    (1, 96, 0, 16, 32, 300, 1, 1, 'lrelu', 1, 0),           # 2 full 128-blocks + 2-tile tail, 1x1 with CK=32
    (2, 32, 0, 19, 37, 70, 3, 1, 'lrelu', 2, 0),            # winograd: odd sizes, partial tiles, two residuals
    (1, 16, 16, 10, 50, 64, 3, 1, 'relu', 0, 0),            # winograd: concat input
    (1, 48, 0, 8, 36, 128, 3, 1, 'lrelu', 0, 1),            # winograd: pixel-shuffle epilogue
    (1, 32, 0, 9, 33, 216, 3, 1, 'sigmoid_from', 0, 0),     # winograd: sigmoid-from-channel epilogue, odd width
    (2, 64, 0, 37, 131, 3, 3, 1, 'none', 0, 0),             # auto: <= 4 output channels -> VALU kernel (conv_last), ragged tiles
    (1, 24, 40, 9, 70, 4, 3, 1, 'lrelu', 1, 0),             # auto: VALU kernel, concat input, 64 channels in 8-chunks, residual
    (3, 20, 0, 5, 3, 1, 3, 1, 'relu', 0, 0),                # auto: VALU kernel, one output channel, channel count not a multiple of 8
    (2, 64, 64, 9, 13, 128, 1, 1, 'relu', 0, 0),            # 1x1 streaming kernel: concat input, 117 pixels (ragged 32-pixel groups)
    (1, 640, 0, 20, 36, 128, 1, 1, 'lrelu', 2, 0),          # 1x1 streaming kernel: TSA feat_fusion shape (5 x 128 channels), two residuals
    (3, 160, 0, 7, 5, 40, 1, 1, 'sigmoid_from', 0, 0),      # 1x1 streaming kernel: 2.5 weight slabs, 2-tile tail launch only
    (2, 216, 0, 12, 40, 128, 3, 1, 'none', 0, 0),           # winograd: ci not a multiple of 16 (data gradient of the offset conv)
    (1, 100, 20, 8, 34, 64, 3, 1, 'lrelu', 1, 0),           # winograd: concat boundary inside a chunk, 120 -> 128 padded channels
]


@pytest.mark.parametrize('algo', ['direct', 'winograd', 'auto'])
@pytest.mark.parametrize('case', CASES)
def test_conv2d_matches_fp64(gpu, case, algo):
    from edvr_amd import ops
    algo = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD, 'auto': ops.CONV_AUTO}[algo]  # winograd falls back where not applicable
    n, c1, c2, h, w, co, ks, stride, actn, nres, out_mode = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()))  # (hash() of a tuple holding strings changes from process to process)
    x1 = torch.randn(n, c1, h, w, generator=g)
    x2 = torch.randn(n, c2, h, w, generator=g) if c2 else None
    wt = torch.randn(co, c1 + c2, ks, ks, generator=g) * 0.1
    b = torch.randn(co, generator=g)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin.double(), wt.double(), b.double(), stride, ks // 2)
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
    if out_mode == 1:
        ref = F.pixel_shuffle(ref, 2)
    wpk = ops.pack_conv_weight(wt.to(gpu))
    y = ops.conv2d(x1.to(gpu), wpk, b.to(gpu), co, ks, x2=None if x2 is None else x2.to(gpu), stride=stride, act=act,
                   act_from=act_from, res1=res[0].to(gpu) if nres > 0 else None, res2=res[1].to(gpu) if nres > 1 else None,
                   out_mode=out_mode, algo=algo)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert _rel(y, ref) < RTOL


def test_conv2d_x2_image_map_and_strided_views(gpu):
    """cat(nbr, ref[clip centre]) without materialising either: x2 image map + channel-sliced input view."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(5)
    b, t, c, h, w, ctr = 2, 3, 32, 12, 20, 1
    feat = torch.randn(b * t, c, h, w, generator=g)
    wt = torch.randn(48, 2 * c, 3, 3, generator=g) * 0.1
    ref_in = torch.cat([feat, feat.view(b, t, c, h, w)[:, ctr:ctr + 1].expand(b, t, c, h, w).reshape(b * t, c, h, w)], 1)
    ref = F.conv2d(ref_in.double(), wt.double(), None, 1, 1)
    fg = feat.to(gpu)
    y = ops.conv2d(fg, ops.pack_conv_weight(wt.to(gpu)), None, 48, 3, x2=fg, x2_map=(t, t, ctr))
    assert _rel(y, ref) < RTOL
    # channel-sliced view as input (image stride larger than c*h*w)
    big = torch.randn(2, 96, h, w, generator=g)
    wt2 = torch.randn(32, 64, 3, 3, generator=g) * 0.1
    ref2 = F.conv2d(big[:, :64].double(), wt2.double(), None, 1, 1)
    y2 = ops.conv2d(big.to(gpu)[:, :64], ops.pack_conv_weight(wt2.to(gpu)), None, 32, 3)
    assert _rel(y2, ref2) < RTOL


def test_transpose_flip_pack_is_data_gradient(gpu):
    """conv with the transpose_flip packing == d/dx of the stride-1 conv."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 24, 10, 14, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(40, 24, 3, 3, generator=g, dtype=torch.float64) * 0.1
    dy = torch.randn(1, 40, 10, 14, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, 1, 1).backward(dy)
    wpk = ops.pack_conv_weight(wt.float().to(gpu), transpose_flip=True)
    dx = ops.conv2d(dy.float().to(gpu), wpk, None, 24, 3)
    assert _rel(dx, x.grad) < RTOL


def test_multi_tensor_packing_is_bit_identical(gpu):
    """ops.prepack_conv_weights (one edvr_conv2d_pack_weights_multi launch for a whole network's weights, the training path)
    against the per-tensor packing functions: every layout, both orientations, 1x1 / 3x3, channel counts that need padding -
    and the cache bookkeeping (nothing to do on a second call, only the touched weight after an in-place update)."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(77)
    shapes = [(128, 128, 3), (64, 3, 3), (3, 64, 3), (216, 128, 3), (70, 20, 3), (64, 640, 1), (48, 96, 1), (128, 256, 3)]
    ws = [torch.nn.Parameter((torch.randn(co, ci, k, k, generator=g) * 0.1).to(gpu)) for co, ci, k in shapes]
    want = {}
    for i, w in enumerate(ws):
        for flip in (False, True):
            want[(i, flip, False)] = ops.pack_conv_weight(w, transpose_flip=flip).clone()
            f4 = ops.f4_weight(w, w.shape[2], transpose_flip=flip)
            if f4 is not None:
                want[(i, flip, True)] = f4.clone()
    ops.invalidate_packed_weights()
    n_jobs = ops.prepack_conv_weights(ws)
    assert n_jobs == 2 * len(ws)
    for (i, flip, f4), ref in want.items():
        got = ops.pack_conv_weight(ws[i], transpose_flip=flip, f4=f4)  # a cache hit now: the buffer the multi launch filled
        assert torch.equal(got, ref), (shapes[i], flip, f4)
    assert ops.prepack_conv_weights(ws) == 0
    with torch.no_grad():
        ws[3].mul_(2.0)  # in-place update: the version moves, as after an optimizer step
    assert ops.prepack_conv_weights(ws) == 2
    assert torch.equal(ops.pack_conv_weight(ws[3], f4=True), 2.0 * want[(3, False, True)])
