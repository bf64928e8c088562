"""-m gpu: the DCN operators in float64 and float16 - the other legs of the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF
(deform_conv_cuda_kernel.cu:781,811,843) - through edvr_dcnv2_{fwd,bwd}_any (csrc/dcn_any.hip), against the fp64 C oracle (pinned to
the reference's own kernels, tests/test_oracle_vs_ref.py) and, for float64, torch.autograd.gradcheck as a user of `ops/dcn` would
run it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, C, H, W, Co, k, stride, pad, dil, groups, dg, bias
    (2, 16, 9, 11, 16, 3, 1, 1, 1, 1, 8, True),    # the EDVR signature
    (2, 8, 7, 9, 6, 3, 2, 1, 1, 2, 2, True),       # stride 2, groups 2
    (1, 8, 10, 10, 4, 3, 1, 2, 2, 1, 4, False),    # dilation 2, no bias
    (1, 6, 5, 7, 5, 1, 1, 0, 1, 1, 3, True),       # 1x1
]


def _mk(case, dtype):
    from oracle import dcn_oracle as O
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, with_bias = case
    g = torch.Generator().manual_seed(sum(map(int, case)))
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    t = dict(x=torch.randn(B, C, H, W, generator=g, dtype=torch.float64),
             offset=torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g, dtype=torch.float64) * 1.5,
             mask=torch.rand(B, dg * k * k, Ho, Wo, generator=g, dtype=torch.float64),
             weight=torch.randn(Co, C // groups, k, k, generator=g, dtype=torch.float64) * 0.2,
             bias=torch.randn(Co, generator=g, dtype=torch.float64) if with_bias else None,
             dy=torch.randn(B, Co, Ho, Wo, generator=g, dtype=torch.float64))
    t = {k_: (None if v is None else v.to(dtype).double()) for k_, v in t.items()}  # the values the low-precision run actually sees
    return t, (stride, pad, dil, groups, dg)


def _rel(a, r):
    return ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float16, 3e-3)], ids=['f64', 'f16'])
def test_dcnv2_forward_backward_in_f64_and_f16(gpu, case, dtype, tol):
    from edvr_amd import modulated_deform_conv
    from oracle import dcn_oracle as O
    t, cfg = _mk(case, dtype)
    ref_y = O.c_forward(t['x'], t['offset'], t['mask'], t['weight'], t['bias'], *cfg)
    ref_g = O.c_backward(t['x'], t['offset'], t['mask'], t['weight'], t['dy'], t['bias'] is not None, *cfg)
    dev = {k: (None if v is None else v.to(gpu, dtype)) for k, v in t.items()}
    leaves = [dev[k] if dev[k] is None else dev[k].requires_grad_() for k in ('x', 'offset', 'mask', 'weight', 'bias')]
    y = modulated_deform_conv(*leaves, *cfg)
    assert y.dtype == dtype and _rel(y.detach(), ref_y) < tol
    y.backward(dev['dy'])
    for name, leaf, r in zip(('dx', 'doffset', 'dmask', 'dweight', 'dbias'), leaves, ref_g):
        if leaf is not None:
            assert leaf.grad.dtype == dtype and _rel(leaf.grad, r) < tol, name


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float16, 3e-3)], ids=['f64', 'f16'])
def test_dcnv1_rectangular_in_f64_and_f16(gpu, dtype, tol):
    from edvr_amd import deform_conv
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(21)
    stride, pad, dil, groups, dg = (2, 1), (1, 2), (1, 2), 2, 2
    Ho, Wo = O._out_hw(9, 11, 3, 3, stride, pad, dil)
    x = torch.randn(2, 8, 9, 11, generator=g, dtype=torch.float64).to(dtype).double()
    w = (torch.randn(6, 4, 3, 3, generator=g, dtype=torch.float64) * 0.2).to(dtype).double()
    off = (torch.randn(2, dg * 18, Ho, Wo, generator=g, dtype=torch.float64) * 1.5).to(dtype).double()
    dy = torch.randn(2, 6, Ho, Wo, generator=g, dtype=torch.float64).to(dtype).double()
    ref_y = O.torch_dcn1_forward(x, off, w, stride, pad, dil, groups, dg)
    ref_g = O.torch_dcn1_backward(x, off, w, dy, stride, pad, dil, groups, dg)
    leaves = [v.to(gpu, dtype).requires_grad_() for v in (x, off, w)]
    y = deform_conv(*leaves, stride, pad, dil, groups, dg)
    assert y.dtype == dtype and _rel(y.detach(), ref_y) < tol
    y.backward(dy.to(gpu, dtype))
    for name, leaf, r in zip(('dx', 'doffset', 'dweight'), leaves, ref_g):
        assert _rel(leaf.grad, r) < tol, name


def test_gradcheck_float64(gpu):
    """What a maintainer of ops/dcn runs: finite differences against the analytic backward, float64.  Offsets are kept away from
    integer sampling positions (the op is only piecewise differentiable there, like the reference's)."""
    from edvr_amd import deform_conv, modulated_deform_conv
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 4, 5, 6, generator=g, dtype=torch.float64)
    off = torch.rand(1, 2 * 2 * 9, 5, 6, generator=g, dtype=torch.float64) * 0.6 + 0.2  # fractional parts in (0.2, 0.8)
    m = torch.rand(1, 2 * 9, 5, 6, generator=g, dtype=torch.float64)
    w = torch.randn(3, 4, 3, 3, generator=g, dtype=torch.float64) * 0.3
    b = torch.randn(3, generator=g, dtype=torch.float64)
    args = [v.to(gpu).requires_grad_() for v in (x, off, m, w, b)]
    assert torch.autograd.gradcheck(lambda *a: modulated_deform_conv(*a, 1, 1, 1, 1, 2), args, eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-10)
    args1 = [v.to(gpu).requires_grad_() for v in (x, off, w)]
    assert torch.autograd.gradcheck(lambda *a: deform_conv(*a, 1, 1, 1, 1, 2), args1, eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=1e-10)


def test_mixed_dtypes_and_unsupported_types_are_refused(gpu):
    from edvr_amd import modulated_deform_conv
    x = torch.randn(1, 8, 6, 6, device=gpu, dtype=torch.float64)
    off = torch.zeros(1, 18, 6, 6, device=gpu, dtype=torch.float64)
    m = torch.ones(1, 9, 6, 6, device=gpu, dtype=torch.float64)
    w = torch.randn(8, 8, 3, 3, device=gpu, dtype=torch.float64)
    with pytest.raises(RuntimeError):
        modulated_deform_conv(x, off, m, w.float(), None, 1, 1, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(x.bfloat16(), off.bfloat16(), m.bfloat16(), w.bfloat16(), None, 1, 1, 1, 1, 1)
    y = modulated_deform_conv(x, off, m, w, None, 1, 1, 1, 1, 1)  # zero offsets, unit mask: the plain convolution, in float64
    ref = torch.nn.functional.conv2d(x.cpu(), w.cpu(), None, 1, 1)
    assert _rel(y, ref) < 1e-13
