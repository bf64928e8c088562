"""CPU: pin the oracle.  The C restatement and the torch restatement of DCNv2 must agree with the reference's
OWN kernels (deform_conv_cuda_kernel.cu compiled serially for the CPU into oracle/_ref) on random cases."""
import pytest
import torch

from oracle import dcn_oracle as O

needs_ref = pytest.mark.skipif(not O.have_ref(), reason='oracle/_ref not built (needs /root/reference)')

CASES = [
    # B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma, mode
    (2, 16, 9, 11, 8, 3, 1, 1, 1, 1, 8, 2.0, 'rand'),
    (1, 16, 8, 8, 16, 3, 1, 1, 1, 1, 8, 0.0, 'rand'),
    (1, 16, 8, 8, 16, 3, 1, 1, 1, 1, 8, 2.0, 'int'),
    (1, 16, 8, 8, 16, 3, 1, 1, 1, 1, 8, 2.0, 'half'),
    (2, 8, 7, 9, 6, 3, 2, 1, 1, 2, 2, 1.5, 'rand'),
    (1, 8, 10, 10, 4, 3, 1, 2, 2, 1, 4, 8.0, 'rand'),
    (1, 4, 6, 6, 4, 1, 1, 0, 1, 1, 1, 1.0, 'rand'),
    (1, 6, 5, 4, 9, 3, 1, 0, 1, 3, 3, 1.0, 'rand'),
]


def _mk(case, dt):
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma, mode = case
    g = torch.Generator().manual_seed(sum(case[:12]).__hash__() % 1000)
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    x = torch.randn(B, C, H, W, generator=g, dtype=dt)
    w = torch.randn(Co, C // groups, k, k, generator=g, dtype=dt) * 0.1
    b = torch.randn(Co, generator=g, dtype=dt)
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g, dtype=dt) * sigma
    if mode == 'int':
        off = off.round()
    if mode == 'half':
        off = off.round() + 0.5
    m = torch.rand(B, dg * k * k, Ho, Wo, generator=g, dtype=dt)
    dy = torch.randn(B, Co, Ho, Wo, generator=g, dtype=dt)
    return x, off, m, w, b, dy, (stride, pad, dil, groups, dg)


@needs_ref
@pytest.mark.parametrize('case', CASES)
def test_c_restatement_equals_reference_kernels_fp64(case):
    x, off, m, w, b, dy, cfg = _mk(case, torch.float64)
    assert torch.equal(O.c_forward(x, off, m, w, b, *cfg), O.ref_forward(x, off, m, w, b, *cfg))  # bit-exact
    for a, r in zip(O.c_backward(x, off, m, w, dy, True, *cfg), O.ref_backward(x, off, m, w, dy, True, *cfg)):
        assert (a - r).abs().max().item() <= 1e-13 * max(1.0, r.abs().max().item())


@needs_ref
@pytest.mark.parametrize('case', CASES)
def test_torch_restatement_autograd_equals_reference_backward(case):
    x, off, m, w, b, dy, cfg = _mk(case, torch.float64)
    t = [v.clone().requires_grad_() for v in (x, off, m, w, b)]
    y = O.dcnv2_torch(*t, *cfg)
    y.backward(dy)
    assert (y.detach() - O.ref_forward(x, off, m, w, b, *cfg)).abs().max().item() < 1e-12
    for a, r in zip(t, O.ref_backward(x, off, m, w, dy, True, *cfg)):
        assert (a.grad - r).abs().max().item() <= 1e-12 * max(1.0, r.abs().max().item())


@needs_ref
def test_fp32_paths_agree():
    x, off, m, w, b, dy, cfg = _mk(CASES[0], torch.float32)
    assert torch.equal(O.c_forward(x, off, m, w, b, *cfg), O.ref_forward(x, off, m, w, b, *cfg))


def test_zero_offset_unit_mask_is_conv2d():
    """Anchor that needs no reference build: DCNv2 with zero offsets and unit mask IS a convolution."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 9, 7, generator=g, dtype=torch.float64)
    w = torch.randn(6, 8, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(6, generator=g, dtype=torch.float64)
    y = O.c_forward(x, torch.zeros(2, 36, 9, 7, dtype=torch.float64), torch.ones(2, 18, 9, 7, dtype=torch.float64), w, b, 1, 1, 1, 1, 2)
    assert (y - F.conv2d(x, w, b, 1, 1)).abs().max().item() < 1e-12


def test_gradcheck_of_the_torch_restatement():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 5, 5, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 4, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    off = (torch.randn(1, 36, 5, 5, generator=g, dtype=torch.float64) * 1.3 + 0.017).requires_grad_()  # away from integers
    m = torch.rand(1, 18, 5, 5, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: O.dcnv2_torch(*a, None, 1, 1, 1, 1, 2), (x, off, m, w), eps=1e-6, atol=1e-6)


@needs_ref
@pytest.mark.parametrize('case', CASES)
def test_dcnv1_oracle_matches_the_references_own_v1_kernels(case):
    """DCNv1: the C oracle run with an all-ones mask equals the reference's deformable_im2col /
    col2im / col2im_coord kernels (.cu:190-465) driven as deform_conv_cuda.cpp:152-488 does - forward and all gradients."""
    x, off, _, w, _, dy, cfg = _mk(case, torch.float64)
    y_ref = O.ref_dcn1_forward(x, off, w, *cfg)
    y_c = O.c_dcn1_forward(x, off, w, *cfg)
    assert torch.equal(y_ref, y_c) or (y_ref - y_c).abs().max().item() <= 1e-13 * max(1.0, y_ref.abs().max().item())
    g_ref = O.ref_dcn1_backward(x, off, w, dy, *cfg)
    g_c = O.c_dcn1_backward(x, off, w, dy, *cfg)
    for name, a, r in zip(('dx', 'doffset', 'dweight'), g_c, g_ref):
        assert (a - r).abs().max().item() <= 1e-12 * max(1.0, r.abs().max().item()), name
