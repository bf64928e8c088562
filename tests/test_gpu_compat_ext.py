"""-m gpu: INTEGRATION.md Level 2 - edvr_amd/compat/deform_conv_ext.py, the stand-in for the reference's pybind11 module
`deform_conv_ext`, driven with EXACTLY the argument lists of the reference's Python call sites
(basicsr/models/ops/dcn/deform_conv.py:49-54,74-91 for DCNv1, :141-146,159-165 for DCNv2): caller-allocated outputs, the
`columns` / `ones` dummies, pre-zeroed gradient buffers.  Results against the CPU oracle (pinned to the reference's kernels)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, r):
    return ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()


def _reference_mdcn_call_sites(ext, input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups, grad_output):
    """ModulatedDeformConvFunction.forward / .backward of the reference restated line by line around the FFI calls."""
    with_bias = bias is not None
    if not with_bias:
        bias = input.new_empty(1)  # fake tensor (:130-131)
    kh, kw = weight.shape[2:4]
    ho = (input.shape[2] + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (input.shape[3] + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    output = input.new_empty((input.size(0), weight.size(0), ho, wo))
    bufs = [input.new_empty(0), input.new_empty(0)]
    ext.modulated_deform_conv_forward(input, weight, bias, bufs[0], offset, mask, output, bufs[1], weight.shape[2], weight.shape[3],
                                      stride, stride, padding, padding, dilation, dilation, groups, deformable_groups, with_bias)
    grad_input, grad_offset, grad_mask = torch.zeros_like(input), torch.zeros_like(offset), torch.zeros_like(mask)
    grad_weight, grad_bias = torch.zeros_like(weight), torch.zeros_like(bias)
    ext.modulated_deform_conv_backward(input, weight, bias, bufs[0], offset, mask, bufs[1], grad_input, grad_weight, grad_bias,
                                       grad_offset, grad_mask, grad_output, weight.shape[2], weight.shape[3], stride, stride, padding,
                                       padding, dilation, dilation, groups, deformable_groups, with_bias)
    return output, grad_input, grad_offset, grad_mask, grad_weight, (grad_bias if with_bias else None)


MDCN_CASES = [  # B, C, H, W, Co, k, stride, pad, dil, groups, dg, bias
    (2, 64, 12, 20, 64, 3, 1, 1, 1, 1, 8, True),    # the EDVR signature (fused forward, tile / strip backward)
    (1, 16, 9, 11, 12, 3, 2, 1, 1, 2, 4, True),     # generic path: stride 2, groups 2
    (2, 8, 10, 10, 8, 3, 1, 2, 2, 1, 2, False),     # dilation 2, no bias (the fake bias tensor)
]


@pytest.mark.parametrize('case', MDCN_CASES)
def test_modulated_deform_conv_ext_call_sites(gpu, case):
    from edvr_amd.compat import deform_conv_ext as ext
    from oracle import dcn_oracle as O
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, with_bias = case
    g = torch.Generator().manual_seed(sum(map(int, case)))
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g) * 1.5
    m = torch.rand(B, dg * k * k, Ho, Wo, generator=g)
    w = torch.randn(Co, C // groups, k, k, generator=g) * 0.1
    b = torch.randn(Co, generator=g) if with_bias else None
    dy = torch.randn(B, Co, Ho, Wo, generator=g)
    cfg = (stride, pad, dil, groups, dg)
    ref_y = O.c_forward(x.double(), off.double(), m.double(), w.double(), None if b is None else b.double(), *cfg)
    ref_g = O.c_backward(x.double(), off.double(), m.double(), w.double(), dy.double(), with_bias, *cfg)
    dev = [t.to(gpu) for t in (x, off, m, w)]
    out = _reference_mdcn_call_sites(ext, dev[0], dev[1], dev[2], dev[3], None if b is None else b.to(gpu), *cfg, dy.to(gpu))
    assert _rel(out[0], ref_y) < 2e-5
    for name, a, r in zip(('dx', 'doffset', 'dmask', 'dweight', 'dbias'), out[1:], ref_g):
        if r is not None:
            assert _rel(a, r) < 1e-4, name
    assert (out[5] is None) == (not with_bias)


def test_ext_backward_takes_its_dx_strategy_from_the_forward_of_the_same_offsets(gpu):
    """The FFI has no argument for the scatter hint: the forward notes the mean |offset| of its call (pinned memory + event), the
    backward of the same offset tensor picks it up - sub-pixel offsets on a 16-channels-per-group layer then run the kernel
    without the dcol buffer.  Results against the oracle either way; the statistic is consumed exactly once."""
    from edvr_amd import ops
    from edvr_amd.compat import deform_conv_ext as ext
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(77)
    B, C, H, W, dg = 2, 128, 16, 40, 8
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 18, H, W, generator=g) * 0.3
    m = torch.rand(B, dg * 9, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    cfg = (1, 1, 1, 1, dg)
    ref_y = O.c_forward(x.double(), off.double(), m.double(), w.double(), b.double(), *cfg)
    ref_g = O.c_backward(x.double(), off.double(), m.double(), w.double(), dy.double(), True, *cfg)
    dev = [t.to(gpu) for t in (x, off, m, w)]
    seen = []
    orig = ops.dcnv2_backward

    def spy(*a, **kw):
        seen.append(kw.get('scatter_hint'))
        return orig(*a, **kw)

    ops.dcnv2_backward = spy
    try:
        ext._OFFSET_STATS.clear()
        # (the reference's Function runs forward and backward in separate calls: the statistic has long arrived by then)
        out_y = torch.empty(B, C, H, W, device=gpu)
        ext.modulated_deform_conv_forward(dev[0], dev[3], b.to(gpu), dev[0].new_empty(0), dev[1], dev[2], out_y, dev[0].new_empty(0),
                                          3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True)
        torch.cuda.synchronize()
        assert ext._stat_key(dev[1]) in ext._OFFSET_STATS
        grads = [torch.zeros_like(t) for t in (dev[0], dev[3], b.to(gpu), dev[1], dev[2])]
        ext.modulated_deform_conv_backward(dev[0], dev[3], b.to(gpu), dev[0].new_empty(0), dev[1], dev[2], dev[0].new_empty(0), grads[0],
                                           grads[1], grads[2], grads[3], grads[4], dy.to(gpu), 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, True)
        torch.cuda.synchronize()
    finally:
        ops.dcnv2_backward = orig
    assert seen == [ops.DCN_SCATTER_STRIP] and not ext._OFFSET_STATS
    assert _rel(out_y, ref_y) < 2e-5
    for name, a, r in zip(('dx', 'dweight', 'dbias', 'doffset', 'dmask'), grads, (ref_g[0], ref_g[3], ref_g[4], ref_g[1], ref_g[2])):
        assert _rel(a, r) < 1e-4, name


def test_ext_accumulates_where_the_reference_does(gpu):
    """grad_input / grad_weight / grad_bias are accumulated into (atomicAdd / addmm_ beta 1 in the reference), grad_offset and
    grad_mask overwritten: buffers that are NOT zero on entry keep their contents in the first three, lose them in the others."""
    from edvr_amd.compat import deform_conv_ext as ext
    g = torch.Generator().manual_seed(5)
    x, w, b = torch.randn(1, 16, 8, 8, generator=g).to(gpu), (torch.randn(16, 16, 3, 3, generator=g) * 0.1).to(gpu), torch.randn(16, generator=g).to(gpu)
    off, m, dy = torch.randn(1, 144, 8, 8, generator=g).to(gpu), torch.rand(1, 72, 8, 8, generator=g).to(gpu), torch.randn(1, 16, 8, 8, generator=g).to(gpu)
    e = x.new_empty(0)

    def run(fill):
        gi, gw, gb = torch.full_like(x, fill), torch.full_like(w, fill), torch.full_like(b, fill)
        go, gm = torch.full_like(off, fill), torch.full_like(m, fill)
        ext.modulated_deform_conv_backward(x, w, b, e, off, m, e, gi, gw, gb, go, gm, dy, 3, 3, 1, 1, 1, 1, 1, 1, 1, 8, True)
        return gi, gw, gb, go, gm
    z, o = run(0.0), run(1.0)
    for a, c in zip(z[:3], o[:3]):
        assert torch.allclose(c, a + 1.0, rtol=0, atol=1e-4 * float(a.abs().max()))
    for a, c in zip(z[3:], o[3:]):
        assert torch.equal(a, c)


def test_deform_conv_ext_v1_call_sites(gpu):
    """DeformConvFunction.forward / .backward (deform_conv.py:37-95): W-before-H argument order, `scale`, im2col_step."""
    from edvr_amd.compat import deform_conv_ext as ext
    from oracle import dcn_oracle as O
    g = torch.Generator().manual_seed(8)
    stride, padding, dilation, groups, dg = (2, 1), (1, 2), (1, 2), 2, 2
    x = torch.randn(2, 8, 9, 11, generator=g)
    w = torch.randn(6, 4, 3, 3, generator=g) * 0.1
    Ho, Wo = O._out_hw(9, 11, 3, 3, stride, padding, dilation)
    off = torch.randn(2, dg * 18, Ho, Wo, generator=g) * 1.5
    dy = torch.randn(2, 6, Ho, Wo, generator=g)
    ref_y = O.torch_dcn1_forward(x.double(), off.double(), w.double(), stride, padding, dilation, groups, dg)
    ref_g = O.torch_dcn1_backward(x.double(), off.double(), w.double(), dy.double(), stride, padding, dilation, groups, dg)
    input, offset, weight, grad_output = (t.to(gpu) for t in (x, off, w, dy))
    output = input.new_empty((2, 6, Ho, Wo))
    bufs_ = [input.new_empty(0), input.new_empty(0)]  # columns, ones
    cur_im2col_step = min(64, input.shape[0])
    ext.deform_conv_forward(input, weight, offset, output, bufs_[0], bufs_[1], weight.size(3), weight.size(2), stride[1], stride[0],
                            padding[1], padding[0], dilation[1], dilation[0], groups, dg, cur_im2col_step)
    assert _rel(output, ref_y) < 2e-5
    grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
    ext.deform_conv_backward_input(input, offset, grad_output, grad_input, grad_offset, weight, bufs_[0], weight.size(3), weight.size(2),
                                   stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0], groups, dg, cur_im2col_step)
    grad_weight = torch.zeros_like(weight)
    ext.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, bufs_[0], bufs_[1], weight.size(3), weight.size(2),
                                        stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0], groups, dg, 1, cur_im2col_step)
    for name, a, r in zip(('dx', 'doffset', 'dweight'), (grad_input, grad_offset, grad_weight), ref_g):
        assert _rel(a, r) < 1e-4, name
    ext.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, bufs_[0], bufs_[1], weight.size(3), weight.size(2),
                                        stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0], groups, dg, 0.5, cur_im2col_step)
    assert _rel(grad_weight, 1.5 * ref_g[2]) < 1e-4  # accumulated with `scale`


def test_ext_refuses_what_the_reference_refuses(gpu):
    from edvr_amd.compat import deform_conv_ext as ext
    x, w = torch.randn(1, 8, 6, 6), torch.randn(8, 8, 3, 3)
    e = x.new_empty(0)
    with pytest.raises(RuntimeError):  # "not implemented on CPU"
        ext.modulated_deform_conv_forward(x, w, e, e, e, e, e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, False)
    xg, wg = x.to(gpu), w.to(gpu)
    with pytest.raises(RuntimeError):  # kernel shape mismatch (deform_conv_cuda.cpp:507-509)
        ext.modulated_deform_conv_forward(xg, wg, e, e, e, e, e, e, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, False)
    with pytest.raises(RuntimeError):  # non-contiguous input (:497)
        ext.modulated_deform_conv_forward(xg.transpose(2, 3), wg, e, e, e, e, e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, False)
