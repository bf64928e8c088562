"""-m gpu: DCNv1 (DeformConv / DeformConvPack / deform_conv, reference deform_conv.py:12-108,188-292) through edvr_dcnv1_*_f32,
against the C oracle run with an all-ones mask - which tests/test_oracle_vs_ref.py pins to the reference's own v1 kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma
    (2, 32, 12, 20, 32, 3, 1, 1, 1, 1, 4, 1.0),   # fused-kernel signature
    (1, 128, 9, 33, 64, 3, 1, 1, 1, 1, 8, 2.5),   # EDVR-like, larger offsets
    (2, 16, 9, 11, 24, 3, 2, 1, 1, 2, 4, 1.5),    # generic path: stride 2, groups 2
    (1, 8, 10, 10, 12, 3, 1, 2, 2, 1, 2, 3.0),    # dilation 2
    (1, 12, 7, 9, 10, 1, 1, 0, 1, 1, 3, 1.0),     # 1x1
]


def _mk(case):
    from oracle import dcn_oracle as O
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma = case
    g = torch.Generator().manual_seed(sum(map(int, case[:11])))
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C // groups, k, k, generator=g) * 0.1
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g) * sigma
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    return x, off, w, dy, (stride, pad, dil, groups, dg)


def _rel(a, r):
    return ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('case', CASES)
def test_dcnv1_forward_backward_vs_oracle(gpu, case):
    from edvr_amd import ops
    from oracle import dcn_oracle as O
    x, off, w, dy, cfg = _mk(case)
    ref_y = O.c_dcn1_forward(x.double(), off.double(), w.double(), *cfg)
    ref_g = O.c_dcn1_backward(x.double(), off.double(), w.double(), dy.double(), *cfg)
    xg, og, wg, dyg = (t.to(gpu) for t in (x, off, w, dy))
    assert _rel(ops.dcnv1_forward(xg, og, wg, *cfg), ref_y) < 2e-5
    for hint in (ops.DCN_SCATTER_LDS, ops.DCN_SCATTER_DEVICE):
        for name, a, r in zip(('dx', 'doffset', 'dweight'), ops.dcnv1_backward(xg, og, wg, dyg, *cfg, scatter_hint=hint), ref_g):
            assert _rel(a, r) < 1e-4, (name, hint)


RECT_CASES = [  # B, C, H, W, Co, (kh, kw), stride, pad, dil, groups, dg, sigma: (h, w) pairs as deform_conv.py:33-36 takes them
    (2, 8, 9, 11, 6, (3, 3), (2, 1), (1, 2), (1, 2), 2, 2, 1.5),   # the geometry of tests/golden/dcn1_rect.pt
    (1, 16, 12, 10, 8, (3, 3), (1, 2), (0, 1), (1, 1), 1, 4, 1.0),
    (2, 12, 8, 13, 12, (1, 3), (1, 1), (0, 1), (1, 2), 1, 3, 2.0),  # rectangular kernel as well
]


@pytest.mark.parametrize('case', RECT_CASES)
def test_dcnv1_rectangular_geometry(gpu, case):
    """Pair-valued stride / padding / dilation through the op, the autograd Function and the module (generic column-buffer
    kernels; EDVR_HW pairs in the C ABI) against the floor/gather oracle, which tests/test_golden.py pins to the reference's
    own kernels on exactly such a geometry."""
    from edvr_amd import DeformConv, deform_conv, ops
    from oracle import dcn_oracle as O
    B, C, H, W, Co, (kh, kw), stride, pad, dil, groups, dg, sigma = case
    g = torch.Generator().manual_seed(B + C + H + W + Co + kh + kw)
    Ho, Wo = O._out_hw(H, W, kh, kw, stride, pad, dil)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C // groups, kh, kw, generator=g) * 0.1
    off = torch.randn(B, dg * 2 * kh * kw, Ho, Wo, generator=g) * sigma
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    cfg = (stride, pad, dil, groups, dg)
    ref_y = O.torch_dcn1_forward(x.double(), off.double(), w.double(), *cfg)
    ref_g = O.torch_dcn1_backward(x.double(), off.double(), w.double(), dy.double(), *cfg)
    xg, og, wg = (t.to(gpu).requires_grad_() for t in (x, off, w))
    assert _rel(ops.dcnv1_forward(xg.detach(), og.detach(), wg.detach(), *cfg), ref_y) < 2e-5
    y = deform_conv(xg, og, wg, stride, pad, dil, groups, dg)
    assert tuple(y.shape) == (B, Co, Ho, Wo) and _rel(y.detach(), ref_y) < 2e-5
    y.backward(dy.to(gpu))
    for name, t, r in zip(('dx', 'doffset', 'dweight'), (xg, og, wg), ref_g):
        assert _rel(t.grad, r) < 1e-4, name
    m = DeformConv(C, Co, (kh, kw), stride=stride, padding=pad, dilation=dil, groups=groups, deformable_groups=dg).to(gpu)
    with torch.no_grad():
        m.weight.copy_(w)
        assert _rel(m(x.to(gpu), off.to(gpu)), ref_y) < 2e-5


def test_deform_conv_autograd_and_module_contract(gpu):
    from edvr_amd import DeformConv, DeformConvPack, deform_conv
    from oracle import dcn_oracle as O
    x, off, w, dy, cfg = _mk(CASES[0])
    stride, pad, dil, groups, dg = cfg
    xg, og, wg = (t.to(gpu).requires_grad_() for t in (x, off, w))
    y = deform_conv(xg, og, wg, stride, pad, dil, groups, dg)
    y.backward(dy.to(gpu))
    ref_g = O.c_dcn1_backward(x.double(), off.double(), w.double(), dy.double(), *cfg)
    for name, t, r in zip(('dx', 'doffset', 'dweight'), (xg, og, wg), ref_g):
        assert _rel(t.grad, r) < 1e-4, name
    # module: parameter names / shapes / init of the reference (deform_conv.py:188-292)
    torch.manual_seed(3)
    m = DeformConvPack(32, 48, 3, stride=1, padding=1, deformable_groups=4).to(gpu)
    assert sorted(m.state_dict()) == ['conv_offset.bias', 'conv_offset.weight', 'weight']
    assert m.conv_offset.weight.shape == (4 * 2 * 9, 32, 3, 3) and m.weight.shape == (48, 32, 3, 3)
    assert float(m.conv_offset.weight.detach().abs().sum()) == 0.0 and float(m.conv_offset.bias.detach().abs().sum()) == 0.0
    xin = torch.randn(2, 32, 10, 14, generator=torch.Generator().manual_seed(4))
    out = m(xin.to(gpu))  # zero offsets: a plain convolution
    ref = torch.nn.functional.conv2d(xin.double(), m.weight.detach().double().cpu(), None, 1, 1)
    assert _rel(out, ref) < 2e-5
    with pytest.raises(AssertionError):
        DeformConv(8, 8, 3, bias=True)
    with pytest.raises(NotImplementedError):
        deform_conv(x, off, w, stride, pad, dil, groups, dg)  # CPU tensors are refused like the reference op
    with pytest.raises(AssertionError):
        deform_conv(xg.detach()[:2].repeat(3, 1, 1, 1)[:3], og.detach()[:2].repeat(3, 1, 1, 1)[:3], wg.detach(), stride, pad, dil, groups, dg, 2)


def test_small_input_padding_shim(gpu):
    """deform_conv.py:234-250: an input smaller than the kernel is zero-padded, convolved, and the output cropped."""
    from edvr_amd import DeformConv
    torch.manual_seed(5)
    m = DeformConv(4, 6, 3, padding=1, deformable_groups=1).to(gpu)
    x = torch.randn(1, 4, 2, 5, generator=torch.Generator().manual_seed(6))
    off = torch.zeros(1, 18, 2, 5)
    out = m(x.to(gpu), off.to(gpu))
    xp = torch.nn.functional.pad(x.double(), (0, 0, 0, 1))
    ref = torch.nn.functional.conv2d(xp, m.weight.detach().double().cpu(), None, 1, 1)[:, :, :2, :]
    assert out.shape == (1, 6, 2, 5) and _rel(out, ref) < 2e-5
