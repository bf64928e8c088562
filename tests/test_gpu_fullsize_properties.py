"""-m gpu: size-independent properties at BASELINE.json's full sizes (EDVR-L, 180x320 LR, 4 clips x 5 frames = 20 images,
128 channels), where the CPU oracle would take minutes.  Each property ties a kernel to an identity that does not depend on
the tensor size, or to a second, independently written kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu
N, C, H, W = 20, 128, 180, 320


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def data(gpu):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, C, H, W, generator=g).to(gpu)
    y = torch.randn(N, C, H, W, generator=g).to(gpu)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to(gpu)
    b = torch.randn(C, generator=g).to(gpu)
    return x, y, w, b


@pytest.mark.parametrize('algo', ['direct', 'winograd'])
def test_conv_is_linear(gpu, data, algo):
    from edvr_amd import ops
    x, y, w, _ = data
    a = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[algo]
    wpk = ops.pack_conv_weight(w)
    lhs = ops.conv2d(1.5 * x - 0.25 * y, wpk, None, C, 3, algo=a)
    rhs = 1.5 * ops.conv2d(x, wpk, None, C, 3, algo=a) - 0.25 * ops.conv2d(y, wpk, None, C, 3, algo=a)
    assert _rel(lhs, rhs) < 2e-5


def test_winograd_and_direct_kernels_agree(gpu, data):
    """Two independently written kernels (different algorithm, tiling and epilogue code) on the full-size trunk conv."""
    from edvr_amd import ops
    x, y, w, b = data
    wpk = ops.pack_conv_weight(w)
    for kw in (dict(act=ops.ACT_LRELU), dict(act=ops.ACT_RELU, res1=y), dict()):
        d = ops.conv2d(x, wpk, b, C, 3, algo=ops.CONV_DIRECT, **kw)
        wg = ops.conv2d(x, wpk, b, C, 3, algo=ops.CONV_WINOGRAD, **kw)
        assert _rel(wg, d) < 2e-5


@pytest.mark.parametrize('algo', ['direct', 'winograd'])
def test_data_and_weight_gradients_are_adjoint(gpu, data, algo):
    """<dy, conv_W(x)> == <dx, x> == <dW, W> for the linear conv: checks dgrad and wgrad at full size against the forward."""
    from edvr_amd import ops
    x, dy, w, _ = data
    a = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[algo]
    out = ops.conv2d(x, ops.pack_conv_weight(w), None, C, 3, algo=a)
    lhs = (dy.double() * out.double()).sum().item()
    dx = ops.conv2d(dy, ops.pack_conv_weight(w, transpose_flip=True), None, C, 3, algo=a)
    dw = ops.conv2d_wgrad(x, None, None, dy, C, 3, 1)
    assert abs((dx.double() * x.double()).sum().item() - lhs) / abs(lhs) < 1e-4
    assert abs((dw.double() * w.double()).sum().item() - lhs) / abs(lhs) < 1e-4


@pytest.mark.parametrize('hint', [3, 7, -1])
def test_dcn_zero_offset_unit_mask_is_the_conv(gpu, data, hint):
    from edvr_amd import ops
    x, _, w, b = data
    off = torch.zeros(N, 144, H, W, device=gpu)
    m = torch.ones(N, 72, H, W, device=gpu)
    y = ops.dcnv2_forward(x, off, m, w, b, 1, 1, 1, 1, 8, halo_hint=hint)
    ref = ops.conv2d(x, ops.pack_conv_weight(w), b, C, 3, algo=ops.CONV_DIRECT)
    assert _rel(y, ref) < 2e-5


def test_dcn_integer_offset_is_a_shifted_conv_and_mask_scales(gpu, data):
    """A constant integer offset (dy, dx) = (2, -3) on every tap samples the input translated by that amount: away from
    the borders the result equals the plain convolution of the shifted image; a constant mask m scales the linear part."""
    from edvr_amd import ops
    x, _, w, _ = data
    off = torch.zeros(N, 144, H, W, device=gpu)
    off[:, 0::2] = 2.0
    off[:, 1::2] = -3.0
    m = torch.full((N, 72, H, W), 0.5, device=gpu)
    y = ops.dcnv2_forward(x, off, m, w, None, 1, 1, 1, 1, 8)
    shifted = torch.zeros_like(x)
    shifted[:, :, :H - 2, 3:] = x[:, :, 2:, :W - 3]
    ref = 0.5 * ops.conv2d(shifted, ops.pack_conv_weight(w), None, C, 3, algo=ops.CONV_DIRECT)
    assert _rel(y[:, :, 4:H - 4, 5:W - 5], ref[:, :, 4:H - 4, 5:W - 5]) < 2e-5


def test_dcn_backward_adjoint_at_full_size(gpu, data):
    """DCN is linear in x and in W for fixed offsets/masks: <dy, dcn(x)> == <dx, x> == <dW, W>."""
    from edvr_amd import ops
    x, dy, w, _ = data
    g = torch.Generator().manual_seed(9)
    n = 4  # 4 images keep the (generic-path) column workspace of the backward modest
    off = (torch.randn(n, 144, H, W, generator=g) * 1.5).to(gpu)
    m = torch.rand(n, 72, H, W, generator=g).to(gpu)
    xs, dys = x[:n].contiguous(), dy[:n].contiguous()
    out = ops.dcnv2_forward(xs, off, m, w, None, 1, 1, 1, 1, 8)
    lhs = (dys.double() * out.double()).sum().item()
    dx, doff, dm, dw, _ = ops.dcnv2_backward(xs, off, m, w, dys, False, 1, 1, 1, 1, 8)
    assert abs((dx.double() * xs.double()).sum().item() - lhs) / abs(lhs) < 1e-4
    assert abs((dw.double() * w.double()).sum().item() - lhs) / abs(lhs) < 1e-4
    assert abs((dm.double() * m.double()).sum().item() - lhs) / abs(lhs) < 1e-4  # also linear in the mask


@pytest.mark.parametrize('sigma', [0.3, 0.8])
def test_dcn_backward_without_dcol_buffer_at_the_training_launch_shape(gpu, sigma):
    """BASELINE configs[3]: 32 clips x 5 frames of 64 x 64 crops in ONE DCN backward launch (160 x 128 x 64 x 64, the shape
    csrc/dcn_bwd_fused.hip is timed on).  The fused kernel (scatter hint STRIP) against the staged path (hint DEVICE: dcol buffer,
    device atomics) on the same inputs, and the adjoint identities <dy, dcn(x)> == <dx, x> == <dW, W> == <dmask, mask>."""
    from edvr_amd import ops
    g = torch.Generator(device=gpu).manual_seed(31)
    n, c, h, w_ = 160, 128, 64, 64
    x = torch.randn(n, c, h, w_, device=gpu, generator=g)
    wt = torch.randn(c, c, 3, 3, device=gpu, generator=g) * 0.03
    off = torch.randn(n, 144, h, w_, device=gpu, generator=g) * sigma
    m = torch.rand(n, 72, h, w_, device=gpu, generator=g)
    dy = torch.randn(n, c, h, w_, device=gpu, generator=g)
    out = ops.dcnv2_forward(x, off, m, wt, None, 1, 1, 1, 1, 8)
    lhs = (dy.double() * out.double()).sum().item()
    fused = ops.dcnv2_backward(x, off, m, wt, dy, True, 1, 1, 1, 1, 8, scatter_hint=ops.DCN_SCATTER_STRIP)
    staged = ops.dcnv2_backward(x, off, m, wt, dy, True, 1, 1, 1, 1, 8, scatter_hint=ops.DCN_SCATTER_DEVICE)
    torch.cuda.synchronize()
    for name, a, b, tol in zip(('dx', 'doffset', 'dmask', 'dweight', 'dbias'), fused, staged, (2e-5, 2e-5, 2e-5, 1e-4, 1e-5)):
        assert torch.isfinite(a).all(), name
        assert _rel(a, b) < tol, name  # (dx, dW: different summation orders; doffset / dmask: the same arithmetic per tap)
    dx, _, dm, dw, _ = fused
    assert abs((dx.double() * x.double()).sum().item() - lhs) / abs(lhs) < 1e-4
    assert abs((dw.double() * wt.double()).sum().item() - lhs) / abs(lhs) < 1e-4
    assert abs((dm.double() * m.double()).sum().item() - lhs) / abs(lhs) < 1e-4


def test_edvr_l_full_size_forward_is_finite_and_batch_consistent(gpu):
    """EDVR-L at the bench shape: clips are independent, so a 2-clip batch equals the two 1-clip runs."""
    from util_edvr import randomize_offsets
    from edvr_amd import EDVR
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(num_feat=128, num_reconstruct_block=40, center_frame_idx=None)).eval().to(gpu)
    x = torch.rand(2, 5, 3, H, W, generator=torch.Generator().manual_seed(0)).to(gpu)
    with torch.no_grad():
        both = net(x)
        one = torch.cat([net(x[:1]), net(x[1:])], 0)
    assert torch.isfinite(both).all() and both.shape == (2, 3, 4 * H, 4 * W)
    assert _rel(both, one) < 1e-5


# ---- the F(4x4,3x3) Winograd kernel at the launch shape of the headline workload: 10 clips x 5 frames = 50 images, 128 -> 128
#      channels, 180 x 320 (11 250 items of 64 channels x 32 tiles walked by 256 persistent workgroups)
N50 = 50


@pytest.fixture(scope='module')
def data50(gpu):
    g = torch.Generator(device=gpu).manual_seed(11)
    x = torch.randn(N50, C, H, W, device=gpu, generator=g)
    y = torch.randn(N50, C, H, W, device=gpu, generator=g)
    w = torch.randn(C, C, 3, 3, device=gpu, generator=g) * 0.03
    b = torch.randn(C, device=gpu, generator=g)
    return x, y, w, b


def _f4(ops, x, w, b=None, flip=False, **kw):
    import ctypes
    from edvr_amd import _lib
    wpk, wf4 = ops.pack_conv_weight(w, transpose_flip=flip), ops.pack_conv_weight(w, transpose_flip=flip, f4=True)
    d = _lib.ConvDesc()
    d.c1, d.n, d.h, d.w, d.co, d.ks, d.stride, d.algo = x.shape[1], x.shape[0], x.shape[2], x.shape[3], C, 3, 1, ops.CONV_WINOGRAD_F4
    d.x1, d.wpk_f4 = x.data_ptr(), wf4.data_ptr()
    buf = ctypes.create_string_buffer(96)
    _lib.lib().edvr_conv2d_kernel_name(ctypes.byref(d), buf, 96)
    assert buf.value == b'conv3x3_winograd_f4_kernel', buf.value
    return ops.conv2d(x, wpk, b, C, 3, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4, **kw)


def test_f4_agrees_with_the_direct_kernel_at_the_bench_launch_shape(gpu, data50):
    """Independent algorithm, tiling, staging and epilogue code on the full 50-image launch, three epilogues."""
    from edvr_amd import ops
    x, y, w, b = data50
    wpk = ops.pack_conv_weight(w)
    for kw in (dict(act=ops.ACT_LRELU), dict(act=ops.ACT_RELU, res1=y), dict()):
        d = ops.conv2d(x, wpk, b, C, 3, algo=ops.CONV_DIRECT, **kw)
        f = _f4(ops, x, w, b, **kw)
        assert _rel(f, d) < 3e-5
        del d, f


def test_f4_is_linear_at_the_bench_launch_shape(gpu, data50):
    from edvr_amd import ops
    x, y, w, _ = data50
    lhs = _f4(ops, 1.5 * x - 0.25 * y, w)
    rhs = 1.5 * _f4(ops, x, w) - 0.25 * _f4(ops, y, w)
    assert _rel(lhs, rhs) < 3e-5


def test_f4_data_gradient_is_the_adjoint_at_the_bench_launch_shape(gpu, data50):
    """<dy, conv_W(x)> == <conv_{W^T flipped}(dy), x>: the transpose_flip packing run through the same kernel is the adjoint
    of the forward - ties the F(4x4) data gradient of the training path to its forward at full size."""
    from edvr_amd import ops
    x, y, w, _ = data50
    out = _f4(ops, x, w)
    # dy correlated with the output, so that <dy, out> is ~ |out|^2 and not the near-cancelling sum of 3.7e8 random products (a
    # purely random dy gives |<dy, out>| ~ 1e-7 of sum |dy . out|: relative errors of THAT number measure nothing)
    dy = 0.5 * out + y
    lhs = (dy.double() * out.double()).sum().item()
    del out
    dx = _f4(ops, dy, w, flip=True)
    assert abs((dx.double() * x.double()).sum().item() - lhs) / abs(lhs) < 1e-5
