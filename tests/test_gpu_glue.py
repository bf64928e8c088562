"""-m gpu: the bandwidth-bound glue kernels against stock torch ops on the CPU (fp64)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-6  # elementwise fp32 work: a few ulps, relative to max|ref|


def _rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('shape', [(2, 5, 64, 16, 16), (1, 7, 128, 9, 20), (1, 3, 8, 5, 7)])
def test_tsa_temporal(gpu, shape):
    from edvr_amd import ops
    b, t, c, h, w = shape
    g = torch.Generator().manual_seed(1)
    emb, al = torch.randn(shape, generator=g) * 0.3, torch.randn(shape, generator=g)
    er = torch.randn(b, c, h, w, generator=g) * 0.3
    prob = torch.sigmoid((emb.double() * er.double().unsqueeze(1)).sum(2))
    ref = al.double() * prob.unsqueeze(2)
    out, p = ops.tsa_temporal(emb.to(gpu), er.to(gpu), al.to(gpu), want_prob=True)
    assert _rel(out, ref) < 1e-5 and _rel(p, prob) < 1e-5


@pytest.mark.parametrize('hw', [(16, 16), (45, 80), (9, 7)])
def test_pool_maxavg(gpu, hw):
    from edvr_amd import ops
    x = torch.randn(2, 6, *hw, generator=torch.Generator().manual_seed(2))
    ref = torch.cat([F.max_pool2d(x.double(), 3, 2, 1), F.avg_pool2d(x.double(), 3, 2, 1)], 1)
    assert _rel(ops.pool_maxavg(x.to(gpu)), ref) < TOL


@pytest.mark.parametrize('hw', [(8, 8), (45, 80), (5, 3), (6, 2), (1, 4), (3, 130), (7, 256), (5, 260), (2, 12)])
def test_upsample2x_and_4x_add(gpu, hw):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, *hw, generator=g)
    ref2 = F.interpolate(x.double(), scale_factor=2, mode='bilinear', align_corners=False) * 2
    assert _rel(ops.upsample2x(x.to(gpu), 2.0), ref2) < TOL
    # the three x2 kernels (16-byte groups with lane exchanges / 8-byte pairs / generic) evaluate the same expression per output:
    # the same values from an 8-byte aligned and from a 4-byte aligned view of the input are bit-identical
    got = ops.upsample2x(x.to(gpu), 2.0)
    for shift in (2, 1):
        flat = torch.zeros(x.numel() + shift, device=gpu)
        flat[shift:] = x.to(gpu).reshape(-1)
        assert torch.equal(ops.upsample2x(flat[shift:].view(x.shape), 2.0), got)
    y = torch.randn(2, 3, 4 * hw[0], 4 * hw[1], generator=g)
    ref4 = y.double() + F.interpolate(x.double(), scale_factor=4, mode='bilinear', align_corners=False)
    assert _rel(ops.upsample4x_add_(y.to(gpu), x.to(gpu)), ref4) < TOL


def test_combine_add_abs_sum_act_bwd(gpu):
    from edvr_amd import ops
    g = torch.Generator().manual_seed(4)
    a, b, c = (torch.randn(2, 8, 6, 10, generator=g) for _ in range(3))
    assert _rel(ops.tsa_combine(a.to(gpu), b.to(gpu), c.to(gpu)), a.double() * torch.sigmoid(b.double()) * 2 + c.double()) < 1e-5
    assert _rel(ops.add(a.to(gpu), b.to(gpu)), a.double() + b.double()) < TOL
    big = torch.randn(3, 12, 6, 10, generator=g)
    assert _rel(ops.abs_sum_per_image(big.to(gpu)[:, :8]), big[:, :8].double().abs().sum((1, 2, 3))) < 1e-5
    odd = torch.randn(2, 3, 7, 9, generator=g)  # element count not a multiple of 4: scalar path
    assert _rel(ops.abs_sum_per_image(odd.to(gpu)), odd.double().abs().sum((1, 2, 3))) < 1e-5
    wide = torch.randn(2, 144, 45, 80, generator=g)  # many blocks per image, 16-byte loads
    assert _rel(ops.abs_sum_per_image(wide.to(gpu)), wide.double().abs().sum((1, 2, 3))) < 1e-5
    y = F.leaky_relu(a, 0.1)
    assert _rel(ops.act_backward(b.to(gpu), y.to(gpu), ops.ACT_LRELU), torch.where(a > 0, b, 0.1 * b).double()) < TOL
    s = torch.sigmoid(a)
    ref = torch.cat([b[:, :5], (b * s * (1 - s))[:, 5:]], 1).double()
    assert _rel(ops.act_backward(b.to(gpu), s.to(gpu), ops.ACT_SIGMOID, act_from=5), ref) < 1e-5


@pytest.mark.parametrize('act', ['none', 'relu', 'lrelu'])
def test_pixel_unshuffle2_with_the_activation_backward_folded_in(gpu, act):
    """Gradient of PixelShuffle(2)(act(z)) w.r.t. z in one launch (the two up-convolutions of the tail, edvr_arch.py:403-404): equals
    the two-launch form bit for bit and torch's pixel_unshuffle of the gated gradient."""
    from edvr_amd import ops
    code = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'lrelu': ops.ACT_LRELU}[act]
    g = torch.Generator().manual_seed(21)
    for shape in [(2, 3, 10, 14), (1, 64, 64, 96), (3, 1, 2, 2)]:
        dy, z = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
        y = {'none': z, 'relu': torch.relu(z), 'lrelu': F.leaky_relu(z, 0.1)}[act]
        gate = {'none': torch.ones_like(z), 'relu': (z > 0).float(), 'lrelu': torch.where(z > 0, 1.0, 0.1)}[act]
        ref = F.pixel_unshuffle((dy * gate).double(), 2)
        got = ops.pixel_unshuffle2_act_backward(dy.to(gpu), y.to(gpu), code)
        assert _rel(got, ref) < TOL
        two = ops.pixel_unshuffle2(ops.act_backward(dy.to(gpu), y.to(gpu), code) if act != 'none' else dy.to(gpu))
        assert torch.equal(got, two)


@pytest.mark.parametrize('hw', [(8, 8), (45, 80), (7, 256), (5, 260), (1, 2), (3, 130), (5, 3), (1, 1), (2, 6)])
def test_upsample2x_backward(gpu, hw):
    """Adjoint of the x2 bilinear upsampling against torch autograd in fp64: the 16-byte kernel with lane exchanges (even widths; rows
    that end inside a wave, waves that end inside a row, single-row / two-column borders) and the generic one (odd widths)."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, *hw, generator=g).double().requires_grad_()
    dy = torch.randn(2, 3, 2 * hw[0], 2 * hw[1], generator=g)
    (F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) * 1.5).backward(dy.double())
    assert _rel(ops.upsample2x_backward(dy.to(gpu), 1.5), x.grad) < TOL


def test_tsa_temporal_backward_two_pass(gpu):
    """TSA temporal attention backward (two parallel passes through a scratch plane) against torch autograd in fp64 on a shape with
    several workgroups per pass."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(32)
    b, t, c, h, w = 3, 5, 24, 20, 36
    emb, al = torch.randn(b, t, c, h, w, generator=g) * 0.3, torch.randn(b, t, c, h, w, generator=g)
    er = torch.randn(b, c, h, w, generator=g) * 0.3
    leaves = [v.double().requires_grad_() for v in (emb, er, al)]
    out = leaves[2] * torch.sigmoid((leaves[0] * leaves[1].unsqueeze(1)).sum(2)).unsqueeze(2)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy.double())
    d_emb, d_ref, d_al = ops.tsa_temporal_backward(emb.to(gpu), er.to(gpu), al.to(gpu), dy.to(gpu))
    for a, r in zip((d_emb, d_ref, d_al), leaves):
        assert _rel(a, r.grad) < 1e-4
