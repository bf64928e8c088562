"""-m gpu: DCNv2 forward/backward through the C ABI against the CPU oracle (fp64 truth)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

FWD_RTOL = 2e-5   # fp32 vs fp64 oracle, relative to max|y|
BWD_RTOL = 1e-4   # gradients (dX goes through fp32 atomics, order not deterministic)


def _mk(B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma, seed, integer=False, half=False):
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(seed)
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C // groups, k, k, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g) * sigma
    if integer:
        off = off.round()
    if half:
        off = off.round() + 0.5
    m = torch.rand(B, dg * k * k, Ho, Wo, generator=g)
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    return x, off, m, w, b, dy


def _rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


CASES = [
    # B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma, kwargs
    (2, 64, 16, 16, 64, 3, 1, 1, 1, 1, 8, 1.0, {}),
    (1, 128, 23, 37, 128, 3, 1, 1, 1, 1, 8, 2.0, {}),
    (1, 64, 16, 20, 64, 3, 1, 1, 1, 1, 8, 0.0, {}),                 # zero offsets: integer grid (fresh model)
    (1, 64, 12, 12, 64, 3, 1, 1, 1, 1, 8, 2.0, {'integer': True}),  # integer taps: one-sided derivative
    (1, 64, 12, 12, 64, 3, 1, 1, 1, 1, 8, 2.0, {'half': True}),
    (1, 64, 12, 16, 64, 3, 1, 1, 1, 1, 8, 16.0, {}),                # mostly out of bounds
    (2, 16, 9, 11, 24, 3, 2, 1, 1, 2, 4, 1.5, {}),                  # generic: stride 2, groups 2, dg 4
    (1, 8, 10, 10, 12, 3, 1, 2, 2, 1, 2, 3.0, {}),                  # dilation 2
    (1, 12, 7, 9, 10, 1, 1, 0, 1, 1, 3, 1.0, {}),                   # 1x1 kernel, unaligned channel counts
    (1, 128, 64, 64, 128, 3, 1, 1, 1, 1, 8, 0.3, {}),               # the training layer: sub-pixel offsets, full-width waves
    (1, 48, 10, 13, 24, 3, 1, 1, 1, 1, 8, 0.4, {}),                 # 6 channels per deformable group (ragged channel quad)
    (1, 64, 8, 64, 32, 3, 1, 1, 1, 1, 2, 0.5, {}),                  # 32 channels per group: two quads per wave
    (1, 16, 1, 5, 16, 3, 1, 1, 1, 1, 4, 0.6, {}),                   # one row
    (1, 32, 12, 150, 32, 3, 1, 1, 1, 1, 8, 0.4, {}),                # wider than a wave: three 64-column strips, ragged last one
    (1, 16, 6, 128, 16, 3, 1, 1, 1, 1, 4, 1.2, {}),                 # two full strips, many taps outside the sub-pixel window
    # 16 channels per deformable group, Co <= 128: the backward without the dcol buffer (dcn_bwd_fused.hip) for 'lds' / 'strip'
    (2, 128, 17, 45, 128, 3, 1, 1, 1, 1, 8, 5.0, {}),               # ragged tiles, many cells beyond the LDS window (slow path)
    (1, 128, 9, 33, 63, 3, 1, 1, 1, 1, 8, 0.7, {}),                 # odd output-channel count (zero rows of the dY tile)
    (1, 128, 12, 12, 64, 3, 1, 1, 1, 1, 8, 2.0, {'integer': True}),
    (1, 128, 12, 12, 64, 3, 1, 1, 1, 1, 8, 2.0, {'half': True}),
    (1, 128, 12, 16, 64, 3, 1, 1, 1, 1, 8, 16.0, {}),               # mostly out of bounds
    (3, 32, 16, 40, 32, 3, 1, 1, 1, 1, 2, 1.0, {}),                 # two groups, three images
]


@pytest.mark.parametrize('scatter', ['lds', 'lds_wide', 'device', 'strip'])  # the dx accumulation strategies of the backward (EDVR_DCN_SCATTER_*)
@pytest.mark.parametrize('case', CASES)
def test_dcnv2_forward_backward_vs_oracle(gpu, case, scatter):
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    *dims, kw = case
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma = dims
    x, off, m, w, b, dy = _mk(*dims, seed=len(str(case)), **kw)
    cfg = (stride, pad, dil, groups, dg)
    ref_y = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), *cfg)
    ref_g = O.c_backward(x.double(), off.double(), m.double(), w.double(), dy.double(), True, *cfg)
    xg, og, mg, wg, bg, dyg = (t.to(gpu) for t in (x, off, m, w, b, dy))
    y = ops.dcnv2_forward(xg, og, mg, wg, bg, *cfg)
    grads = ops.dcnv2_backward(xg, og, mg, wg, dyg, True, *cfg,
                               scatter_hint={'lds': ops.DCN_SCATTER_LDS, 'lds_wide': ops.DCN_SCATTER_LDS_WIDE, 'device': ops.DCN_SCATTER_DEVICE,
                                             'strip': ops.DCN_SCATTER_STRIP}[scatter])
    torch.cuda.synchronize()
    assert _rel(y, ref_y) < FWD_RTOL
    # d(offset) is discontinuous where a sampling position sits on an integer: if fp32 rounding of `base + offset` crosses it,
    # floor() picks the other cell (the reference in fp32 does the same).  Such taps - where the ORACLE run in fp32 leaves its own
    # fp64 result - are excluded from the comparison (1 tap of 295 k in the 64x64 case).
    ref32 = O.c_backward(x, off, m, w, dy, True, *cfg)
    for name, a, r, r32 in zip(('dx', 'doffset', 'dmask', 'dweight', 'dbias'), grads, ref_g, ref32):
        if name == 'doffset':
            flip = (r32.double() - r).abs() > 1e-3 * r.abs().max()
            assert flip.sum().item() <= 2, 'fp32 floor flips should be rare'
            a, r = a.double().cpu().masked_fill(flip, 0.), r.masked_fill(flip, 0.)
        assert _rel(a, r) < BWD_RTOL, name


def test_dcnv2_zero_offset_unit_mask_is_conv2d(gpu):
    """Independent anchor: zero offsets + unit mask must reduce to an ordinary convolution."""
    import torch.nn.functional as F
    from edvr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 20, 24, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    off = torch.zeros(2, 144, 20, 24)
    m = torch.ones(2, 72, 20, 24)
    y = ops.dcnv2_forward(*(t.to(gpu) for t in (x, off, m, w, b)), 1, 1, 1, 1, 8)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    assert _rel(y, ref) < FWD_RTOL


def test_autograd_function_and_strided_offset_views(gpu):
    """modulated_deform_conv autograd == oracle autograd; offset/mask given as channel slices of one tensor."""
    from edvr_amd import modulated_deform_conv
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(4)
    B, C, H, W = 2, 64, 14, 18
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    om = torch.randn(B, 216, H, W, generator=g)
    om[:, 144:] = torch.sigmoid(om[:, 144:])
    dy = torch.randn(B, C, H, W, generator=g)
    cpu = [t.double().requires_grad_() for t in (x, om, w, b)]
    yo = O.dcnv2_torch(cpu[0], cpu[1][:, :144], cpu[1][:, 144:], cpu[2], cpu[3], 1, 1, 1, 1, 8)
    yo.backward(dy.double())
    dev = [t.to(gpu).requires_grad_() for t in (x, om, w, b)]
    y = modulated_deform_conv(dev[0], dev[1][:, :144], dev[1][:, 144:], dev[2], dev[3], 1, 1, 1, 1, 8)
    y.backward(dy.to(gpu))
    assert _rel(y.detach(), yo.detach()) < FWD_RTOL
    for a, r in zip(dev, cpu):
        assert _rel(a.grad, r.grad) < BWD_RTOL


@pytest.mark.parametrize('sigma', [0.3, 1.5])
def test_backward_without_dcol_buffer_writes_strided_gradient_slices(gpu, sigma):
    """The training path's call: offsets / masks are channel slices of conv_offset's output, d(offset) / d(mask) are written into
    slices of ONE gradient buffer (image strides != plane count x plane size), scatter hint STRIP -> csrc/dcn_bwd_fused.hip
    (16 channels per deformable group).  Against the CPU oracle and against the staged path (hint DEVICE) on the same inputs."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(11)
    B, C, H, W = 3, 128, 19, 40
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    om = torch.randn(B, 216, H, W, generator=g) * sigma
    om[:, 144:] = torch.sigmoid(om[:, 144:])
    dy = torch.randn(B, C, H, W, generator=g)
    ref = O.c_backward(x.double(), om[:, :144].double(), om[:, 144:].double(), w.double(), dy.double(), True, 1, 1, 1, 1, 8)
    xg, omg, wg, dyg = (t.to(gpu) for t in (x, om, w, dy))
    outs = {}
    for name, hint in (('fused', ops.DCN_SCATTER_STRIP), ('staged', ops.DCN_SCATTER_DEVICE)):
        dom = torch.full_like(omg, float('nan'))
        dx, _, _, dw, db = ops.dcnv2_backward(xg, omg[:, :144], omg[:, 144:], wg, dyg, True, 1, 1, 1, 1, 8, doffset=dom[:, :144],
                                              dmask=dom[:, 144:], scatter_hint=hint)
        torch.cuda.synchronize()
        assert torch.isfinite(dom).all(), 'every plane of both slices is written'
        outs[name] = (dx, dom[:, :144], dom[:, 144:], dw, db)
    ref32 = O.c_backward(x, om[:, :144], om[:, 144:], w, dy, True, 1, 1, 1, 1, 8)
    for name in outs:
        for what, a, r, r32 in zip(('dx', 'doffset', 'dmask', 'dweight', 'dbias'), outs[name], ref, ref32):
            if what == 'doffset':  # (fp32 floor flips, as in the test above)
                flip = (r32.double() - r).abs() > 1e-3 * r.abs().max()
                assert flip.sum().item() <= 2
                a, r = a.double().cpu().masked_fill(flip, 0.), r.masked_fill(flip, 0.)
            assert _rel(a, r) < BWD_RTOL, (name, what)
    for a, b_ in zip(outs['fused'][1:3], outs['staged'][1:3]):  # same arithmetic per (pixel, tap): no summation-order freedom
        assert (a - b_).abs().max().item() <= 2e-5 * b_.abs().max().item()


def test_cpu_tensor_is_refused():
    """Same behaviour as the reference op (deform_conv.py:133-134): no CPU path."""
    from edvr_amd import modulated_deform_conv
    x = torch.randn(1, 8, 6, 6)
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(x, torch.zeros(1, 18, 6, 6), torch.ones(1, 9, 6, 6), torch.randn(8, 8, 3, 3), None, 1, 1, 1, 1, 1)


@pytest.mark.parametrize('sigma', [0.5, 3.0, 12.0])
def test_fused_kernel_halo_classes_and_generic_path_agree(gpu, sigma):
    """halo_hint is a performance hint only: R=3, R=7 (fused, LDS halo + global slow path) and -1 (generic column-buffer
    path) must all match the oracle, whatever the offset magnitude."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    x, off, m, w, b, _ = _mk(2, 64, 21, 45, 96, 3, 1, 1, 1, 1, 8, sigma, seed=int(sigma * 10))
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, 8)
    dev = [t.to(gpu) for t in (x, off, m, w, b)]
    for hint in (3, 7, -1):
        y = ops.dcnv2_forward(*dev, 1, 1, 1, 1, 8, halo_hint=hint)
        assert _rel(y, ref) < FWD_RTOL, hint
    y_act = ops.dcnv2_forward(*dev, 1, 1, 1, 1, 8, act=ops.ACT_LRELU, halo_hint=3)
    assert _rel(y_act, torch.nn.functional.leaky_relu(ref, 0.1)) < FWD_RTOL


# ------------------------------------------------------------------------------------------------ per-tap shifted windows
def _smooth_field(B, dg, H, W, sigma, g, noise=0.15, low=0.5, outliers=0):
    """What a trained conv_offset produces: a constant per (group, tap, dy | dx) ~ N(0, sigma^2), a low-frequency motion component
    and a small white residual; `outliers` pixels per plane jump by +-9 px (object boundaries: the kernel's slow path)."""
    import torch.nn.functional as F
    bias = torch.randn(1, dg * 18, 1, 1, generator=g) * sigma
    coarse = torch.randn(B, dg * 18, (H + 15) // 16 + 1, (W + 15) // 16 + 1, generator=g) * low
    lowf = F.interpolate(coarse, scale_factor=16, mode='bilinear', align_corners=False)[:, :, :H, :W]
    off = bias + lowf + torch.randn(B, dg * 18, H, W, generator=g) * noise
    for _ in range(outliers):
        yy, xx = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
        off[:, :, yy, xx] += 9.0 * (1 if _ % 2 else -1)
    return off.contiguous()


def _piecewise_field(B, dg, H, W, jump, g, noise=0.05):
    """Two rigidly moving regions: per-(group, tap, dy | dx) constants ~ N(0, 3^2) on one side of a DIAGONAL boundary, the same plus
    a common motion difference of `jump` pixels (random direction per plane) on the other.  The boundary x + y = const crosses every
    8 x 32 tile of the image, so every workgroup of the tap-window kernels sees taps whose cells split between two windows."""
    base = torch.randn(1, dg * 18, 1, 1, generator=g) * 3.0
    step = (torch.rand(1, dg * 18, 1, 1, generator=g) * 2 - 1) * jump
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    side = (((xx + yy) // 24) % 2).float().view(1, 1, H, W)  # stripes of width 24 along the diagonal: boundaries in every tile
    return (base + side * step + torch.randn(B, dg * 18, H, W, generator=g) * noise).contiguous()


@pytest.mark.parametrize('jump', [4.0, 8.0])
@pytest.mark.parametrize('geom', [(2, 128, 40, 96, 128, 8), (1, 64, 24, 64, 64, 8)])
def test_tap_window_kernel_on_piecewise_constant_motion(gpu, geom, jump):
    """Object boundaries: a piecewise-constant field with jumps of 4 / 8 px along diagonals through every tile -
    the per-lane fix-up pass of the tap-window forward and, in the backward, the LDS-window scatter's slow path.  Forward against the
    C oracle, and all five gradients of the backward (whatever dX strategy the statistics pick, and every forced one) against it."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    B, C, H, W, Co, dg = geom
    g = torch.Generator().manual_seed(int(jump) * 100 + C)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    off = _piecewise_field(B, dg, H, W, jump, g)
    m = torch.rand(B, dg * 9, H, W, generator=g)
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, dg)
    dev = [t.to(gpu) for t in (x, off, m, w, b)]
    y = ops.dcnv2_forward(*dev, 1, 1, 1, 1, dg, halo_hint=ops.DCN_HALO_TAPWIN)
    assert _rel(y, ref) < FWD_RTOL
    dy = torch.randn(B, Co, H, W, generator=g)
    gref = O.c_backward(x.double(), off.double(), m.double(), w.double(), dy.double(), True, 1, 1, 1, 1, dg)
    g32 = O.c_backward(x, off, m, w, dy, True, 1, 1, 1, 1, dg)  # (taps whose fp32 position rounds across an integer: excluded, as above)
    for scatter in (ops.DCN_SCATTER_AUTO, ops.DCN_SCATTER_LDS, ops.DCN_SCATTER_LDS_WIDE, ops.DCN_SCATTER_DEVICE):
        got = ops.dcnv2_backward(dev[0], dev[1], dev[2], dev[3], dy.to(gpu), True, 1, 1, 1, 1, dg, scatter_hint=scatter)
        for name, a_, r_, r32 in zip(('dx', 'doffset', 'dmask', 'dw', 'db'), got, gref, g32):
            if name == 'doffset':
                flip = (r32.double() - r_).abs() > 1e-3 * r_.abs().max()
                assert flip.sum().item() <= 4, 'fp32 floor flips should be rare'
                a_, r_ = a_.double().cpu().masked_fill(flip, 0.), r_.masked_fill(flip, 0.)
            assert _rel(a_, r_) < BWD_RTOL, (name, scatter, _rel(a_, r_))


TAPWIN_CASES = [
    # B, C, H, W, Co, dg, sigma, outliers, act
    (2, 128, 21, 64, 128, 8, 4.0, 0, 0),    # EDVR-L geometry, ragged tile rows, multi-pixel smooth field
    (1, 128, 16, 32, 128, 8, 10.0, 0, 2),   # one tile column: most windows reach beyond the image (zero pieces) + LeakyReLU epilogue
    (2, 64, 24, 40, 64, 8, 3.0, 4, 0),      # EDVR-M geometry (8 channels per group, two channel tiles), ragged tile columns, outliers
    (1, 128, 9, 36, 96, 8, 2.0, 3, 0),      # three channel tiles, width % 32 != 0, outliers
    (1, 128, 40, 100, 40, 8, 6.0, 0, 1),    # tail channel tile (Co % 32 != 0), ReLU
    (1, 128, 8, 32, 128, 8, 0.0, 0, 0),     # zero offsets, exactly one tile
    (1, 256, 12, 48, 160, 16, 5.0, 2, 0),   # 16 groups: more than the kernel's 8 -> the request falls back to the R = 7 kernel
    (1, 128, 12, 48, 160, 8, 5.0, 2, 0),    # two launches (128 + 32 output channels)
    (3, 128, 33, 72, 128, 8, 64.0, 0, 0),   # displacements larger than the image: every window empty or clipped
]


@pytest.mark.parametrize('case', TAPWIN_CASES)
def test_tap_window_kernel_matches_oracle(gpu, case):
    """EDVR_DCN_HALO_TAPWIN (csrc/dcn_tapwin.hip): window origins follow each (group, tap)'s displacement.  Against the C oracle in
    fp64, and against the zero-centred halo kernel on the same inputs (same arithmetic per sample: only the MFMA summation order
    over channels differs)."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    B, C, H, W, Co, dg, sigma, outliers, act = case
    g = torch.Generator().manual_seed(1000 + TAPWIN_CASES.index(case))
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    off = _smooth_field(B, dg, H, W, sigma, g, outliers=outliers)
    m = torch.rand(B, dg * 9, H, W, generator=g)
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, dg)
    if act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.1)
    elif act == 1:
        ref = torch.relu(ref)
    dev = [t.to(gpu) for t in (x, off, m, w, b)]
    y = ops.dcnv2_forward(*dev, 1, 1, 1, 1, dg, act=act, halo_hint=ops.DCN_HALO_TAPWIN)
    assert _rel(y, ref) < FWD_RTOL
    y7 = ops.dcnv2_forward(*dev, 1, 1, 1, 1, dg, act=act, halo_hint=7)
    assert (y - y7).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('sigma', [0.5, 3.0, 12.0])
def test_tap_window_kernel_on_white_noise_offsets(gpu, sigma):
    """A rough field is the kernel's worst case, not a wrong one: every lane whose cell leaves its window goes through the fix-up
    pass (global gathers with the full bounds logic).  Same inputs as the halo-class test above."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    x, off, m, w, b, _ = _mk(2, 64, 21, 44, 96, 3, 1, 1, 1, 1, 8, sigma, seed=int(sigma * 10))
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, 8)
    y = ops.dcnv2_forward(*(t.to(gpu) for t in (x, off, m, w, b)), 1, 1, 1, 1, 8, halo_hint=ops.DCN_HALO_TAPWIN)
    assert _rel(y, ref) < FWD_RTOL


def test_tap_window_hint_falls_back_where_the_kernel_does_not_apply(gpu):
    """Widths that are not a multiple of 4, other group sizes and unaligned views run the R = 7 halo kernel under the same hint."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    for (C, H, W, dg) in ((64, 14, 45, 8), (48, 10, 36, 8), (64, 12, 20, 8)):
        x, off, m, w, b, _ = _mk(1, C, H, W, 64, 3, 1, 1, 1, 1, dg, 2.0, seed=C + W)
        ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, dg)
        y = ops.dcnv2_forward(*(t.to(gpu) for t in (x, off, m, w, b)), 1, 1, 1, 1, dg, halo_hint=ops.DCN_HALO_TAPWIN)
        assert _rel(y, ref) < FWD_RTOL, (C, H, W, dg)
    # a view whose storage offset is not 16-byte aligned
    x, off, m, w, b, _ = _mk(1, 64, 12, 36, 64, 3, 1, 1, 1, 1, 8, 2.0, seed=5)
    buf = torch.zeros(x.numel() + 1, device=gpu)
    xv = buf[1:].view(x.shape)
    xv.copy_(x)
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, 8)
    y = ops.dcnv2_forward(xv, *(t.to(gpu) for t in (off, m, w, b)), 1, 1, 1, 1, 8, halo_hint=ops.DCN_HALO_TAPWIN)
    assert _rel(y, ref) < FWD_RTOL


def test_offset_statistics_kernels(gpu):
    """edvr_abs_stats_f32 and conv_offset's statistics (abs_sum epilogue of the F(4x4) kernel + the roughness estimate) against torch, and the hints derived from them."""
    from edvr_amd import functional as F_, ops
    g = torch.Generator().manual_seed(21)
    t = torch.randn(3, 10, 12, 40, generator=g)
    st = ops.abs_stats_per_image(t.to(gpu)).cpu()
    want_sum = t.double().abs().sum((1, 2, 3))
    t4 = t.double().view(3, 10, 12, 10, 4)
    want_diff = (t4[..., 1:] - t4[..., :-1]).abs().sum((1, 2, 3, 4))
    assert _rel(st[0], want_sum) < 1e-5 and _rel(st[1], want_diff) < 1e-5
    absmean, rough = ops.offset_stats(st, t.numel())
    assert abs(absmean - t.abs().mean().item()) < 1e-4 and abs(rough - (t[..., 1:] - t[..., :-1]).abs().mean().item()) < 0.02
    odd = torch.randn(2, 3, 5, 7, generator=g)  # rows that are not whole 16-byte groups: sums only, roughness unknown
    st = ops.abs_stats_per_image(odd.to(gpu)).cpu()
    assert _rel(st[0], odd.double().abs().sum((1, 2, 3))) < 1e-5 and (st[1] == -1).all()
    assert ops.offset_stats(st, odd.numel())[1] is None
    # the conv epilogue: conv_offset-shaped layer on the F(4x4) kernel
    m = torch.nn.Conv2d(64, 216, 3, 1, 1)
    x = torch.randn(2, 64, 16, 64, generator=g)
    ref = m(x)[:, :144].double()
    with torch.no_grad():
        om, stats = F_.offset_mask_conv_stats(m.to(gpu), x.to(gpu))
    r4 = ref.view(2, 144, 16, 16, 4)
    assert _rel(stats[0], ref.abs().sum((1, 2, 3))) < 1e-4
    want_diff = (r4[..., 1:] - r4[..., :-1]).abs().sum((1, 2, 3, 4))
    # (n = 2: both images are sampled, each standing for itself - ops.conv2d samples up to four images spread over the batch)
    assert abs(stats[1].sum().item() - want_diff.sum().item()) < 1e-4 * want_diff.sum().item()
    assert F_.halo_hint_from_stats(3.0, 0.1) == ops.DCN_HALO_TAPWIN and F_.halo_hint_from_stats(51.0, 72.0) == -1
    assert F_.halo_hint_from_stats(None, None) == ops.DCN_HALO_TAPWIN and F_.halo_hint_from_stats(0.8, 1.1) == 7
    assert F_.scatter_hint_from_stats(3.0, 0.1) == ops.DCN_SCATTER_LDS and F_.scatter_hint_from_stats(0.2, 0.1) == ops.DCN_SCATTER_STRIP


@pytest.mark.parametrize('case', TAPWIN_CASES)
def test_split_tap_window_kernel_matches_oracle(gpu, case):
    """csrc/dcn_tapwin_s.hip (weights and sampled columns as f16 (hi, lo) pairs on the f16 matrix pipe, all four cross products) on the
    tap-window kernel's own cases, at the SAME tolerance against the C oracle in fp64, and not the less accurate of the two."""
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    B, C, H, W, Co, dg, sigma, outliers, act = case
    g = torch.Generator().manual_seed(1000 + TAPWIN_CASES.index(case))
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    off = _smooth_field(B, dg, H, W, sigma, g, outliers=outliers)
    m = torch.rand(B, dg * 9, H, W, generator=g)
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), 1, 1, 1, 1, dg)
    if act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.1)
    elif act == 1:
        ref = torch.relu(ref)
    dev = [t.to(gpu) for t in (x, off, m, w, b)]
    bound = ops.amax(dev[0])
    names = []
    ops.LAUNCH_HOOK = lambda name, flops, launch, nbytes, executed=None: (names.append(name), launch())
    try:
        y = ops.dcnv2_forward(*dev, 1, 1, 1, 1, dg, act=act, halo_hint=ops.DCN_HALO_TAPWIN, xm_bound=bound)
    finally:
        ops.LAUNCH_HOOK = None
    y32 = ops.dcnv2_forward(*dev, 1, 1, 1, 1, dg, act=act, halo_hint=ops.DCN_HALO_TAPWIN)
    if dg <= 8:
        assert names == ['dcnv2_fwd[dcn_tapwin_split_fwd_kernel]'], names
    assert _rel(y, ref) < FWD_RTOL, _rel(y, ref)
    assert _rel(y, ref) < 1.5 * _rel(y32, ref) + 2e-7, (_rel(y, ref), _rel(y32, ref))


@pytest.mark.parametrize('scale', [1e-20, 1e-3, 1.0, 1e4, 1e20])
def test_split_tap_window_kernel_is_scale_invariant(gpu, scale):
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, 128, 16, 64, generator=g) * scale
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.1
    off = _piecewise_field(1, 8, 16, 64, 6.0, g)
    m = torch.rand(1, 72, 16, 64, generator=g)
    ref = O.c_forward(x.double(), off.double(), m.double(), w.double(), None, 1, 1, 1, 1, 8)
    dev = [t.to(gpu) for t in (x, off, m, w)]
    y = ops.dcnv2_forward(*dev, None, 1, 1, 1, 1, 8, halo_hint=ops.DCN_HALO_TAPWIN, xm_bound=ops.amax(dev[0]))
    assert torch.isfinite(y).all() and _rel(y, ref) < FWD_RTOL, _rel(y, ref)


# ---- the dW product of the backward with split operands (csrc/gemm_nt_s.hip): every case of the backward's own test, dW against
#      the fp64 oracle at the fp32 kernel's tolerance and no further from it than the fp32 kernel is
@pytest.mark.parametrize('case', CASES)
def test_backward_split_dw_matches_oracle(gpu, case):
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    *dims, kw = case
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma = dims
    x, off, m, w, b, dy = _mk(*dims, seed=len(str(case)) + 5, **kw)
    cfg = (stride, pad, dil, groups, dg)
    ref = O.c_backward(x.double(), off.double(), m.double(), w.double(), dy.double(), True, *cfg)
    xg, og, mg, wg, dyg = (t.to(gpu) for t in (x, off, m, w, dy))
    g32 = ops.dcnv2_backward(xg, og, mg, wg, dyg, True, *cfg)
    gs = ops.dcnv2_backward(xg, og, mg, wg, dyg, True, *cfg, xm_bound=ops.amax(xg), dy_bound=ops.amax(dyg))
    torch.cuda.synchronize()
    e32, es = _rel(g32[3], ref[3]), _rel(gs[3], ref[3])
    assert es < BWD_RTOL and es < 1.5 * e32 + 2e-7, (es, e32)
    for a, c in zip(gs[:3] + gs[4:], g32[:3] + g32[4:]):  # everything else is the same code: equal up to the order of the dx atomics
        assert _rel(a, c.double().cpu()) < 1e-5


@pytest.mark.parametrize('scale', [(1e-20, 1e12), (1e-3, 1.0), (1e4, 1e-9), (1e18, 1e-30)])
def test_backward_split_dw_is_scale_invariant(gpu, scale):
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    sx, sd = scale
    x, off, m, w, b, dy = _mk(1, 128, 17, 45, 128, 3, 1, 1, 1, 1, 8, 0.4, seed=91)
    x, dy = x * sx, dy * sd
    ref = O.c_backward(x.double(), off.double(), m.double(), w.double(), dy.double(), False, 1, 1, 1, 1, 8)
    xg, og, mg, wg, dyg = (t.to(gpu) for t in (x, off, m, w, dy))
    # loose bounds (x 1000) are bounds too
    gs = ops.dcnv2_backward(xg, og, mg, wg, dyg, False, 1, 1, 1, 1, 8, xm_bound=ops.amax(xg) * 1000., dy_bound=ops.amax(dyg) * 1000.)
    assert torch.isfinite(gs[3]).all() and _rel(gs[3], ref[3]) < BWD_RTOL, _rel(gs[3], ref[3])


def test_conv1x1_weight_gradient_split_gemm(gpu):
    """The 1x1 weight gradient runs the same product: split when both tensors carry a bound, the fp32 kernel otherwise."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(3)
    for (n, ci, h, w, co) in [(3, 640, 20, 36, 128), (2, 64, 17, 23, 96), (1, 128, 8, 8, 64)]:
        x = torch.randn(n, ci, h, w, generator=g)
        dz = torch.randn(n, co, h, w, generator=g) * 1e-3
        ref = torch.einsum('nohw,nihw->oi', dz.double(), x.double()).reshape(co, ci, 1, 1)
        xg, dzg = x.to(gpu), dz.to(gpu)
        seen = []
        ops.LAUNCH_HOOK = lambda name, flops, launch, *a: (seen.append(name), launch())
        try:
            d32 = ops.conv2d_wgrad(xg, None, None, dzg, co, 1, 1)
            ops.set_bound(xg, ops.amax(xg))
            ops.set_bound(dzg, ops.amax(dzg))
            ds = ops.conv2d_wgrad(xg, None, None, dzg, co, 1, 1)
        finally:
            ops.LAUNCH_HOOK = None
        assert 'gemm_nt_kernel' in seen and ('gemm_nt_split_kernel' in seen) == bool(ops.F4S_TRAINING), seen
        e32, es = _rel(d32, ref), _rel(ds, ref)
        assert es < 2e-6 and es < 1.5 * e32 + 2e-7, (es, e32)
