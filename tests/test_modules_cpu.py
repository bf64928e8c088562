"""CPU: host-side contract of the module API (no compute: there is no CPU path)."""
import os

import pytest
import torch

from oracle import ref_import

EXPECTED = {  # measured on the instantiated reference classes
    'M': (dict(num_feat=64, num_reconstruct_block=10), 144, 3300131),
    'L': (dict(num_feat=128, num_reconstruct_block=40, center_frame_idx=None), 264, 20633827),
    'L_T7': (dict(num_feat=128, num_frame=7, num_reconstruct_block=40, center_frame_idx=None), 264, 20699363),
    'L_deblur': (dict(num_feat=128, num_reconstruct_block=40, hr_in=True, with_predeblur=True), None, 23602019),
}


@pytest.mark.parametrize('name', list(EXPECTED))
def test_state_dict_contract(name):
    from edvr_amd import EDVR
    kwargs, n_keys, n_params = EXPECTED[name]
    net = EDVR(**kwargs)
    if n_keys is not None:
        assert len(net.state_dict()) == n_keys
    assert sum(p.numel() for p in net.parameters()) == n_params
    names = [n for n, _ in net.named_parameters()]
    assert any('dcn' in n for n in names) and any(n.startswith('fusion') for n in names)  # LR groups / TSA freeze rely on these
    assert net.pcd_align.dcn_pack['l3'].conv_offset.weight.shape == (216, kwargs['num_feat'], 3, 3)
    assert net.pcd_align.cas_dcnpack._version == 2
    assert net.center_frame_idx == kwargs.get('num_frame', 5) // 2


@pytest.mark.skipif(not ref_import.available(), reason='/root/reference not present')
@pytest.mark.parametrize('kwargs', [dict(num_feat=64, num_reconstruct_block=10),
                                    dict(num_feat=32, num_frame=3, num_reconstruct_block=2, center_frame_idx=1, with_tsa=False),
                                    dict(num_feat=32, num_reconstruct_block=2, hr_in=True, with_predeblur=True)])
def test_same_seed_same_parameters_as_reference(kwargs):
    """Key names, order, shapes AND initial values equal the reference's classes under the same seed."""
    from edvr_amd import EDVR
    ea, _ = ref_import.load()
    torch.manual_seed(7)
    ours = EDVR(**kwargs).state_dict()
    torch.manual_seed(7)
    theirs = ea.EDVR(**kwargs).state_dict()
    assert list(ours) == list(theirs)
    assert all(torch.equal(ours[k], theirs[k]) for k in ours)
    ea.EDVR(**kwargs).load_state_dict(ours, strict=True)


def test_cpu_tensors_are_refused_like_the_reference():
    from edvr_amd import EDVR, ModulatedDeformConvPack, modulated_deform_conv
    with pytest.raises(NotImplementedError):  # deform_conv.py:133-134
        modulated_deform_conv(torch.randn(1, 8, 6, 6), torch.zeros(1, 18, 6, 6), torch.ones(1, 9, 6, 6), torch.randn(8, 8, 3, 3),
                              None, 1, 1, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        EDVR(num_feat=16, num_reconstruct_block=1)(torch.rand(1, 5, 3, 16, 16))
    with pytest.raises(NotImplementedError):
        ModulatedDeformConvPack(8, 8, 3, padding=1)(torch.randn(1, 8, 6, 6))
    # the glue operators of the training path have no CPU route either: a CPU tensor is refused, never computed on the host
    from edvr_amd import ops
    x = torch.randn(1, 2, 4, 4)
    for call in (lambda: ops.upsample2x(x), lambda: ops.upsample2x_backward(x), lambda: ops.pixel_unshuffle2(x),
                 lambda: ops.pixel_unshuffle2_act_backward(x, x, ops.ACT_LRELU), lambda: ops.act_backward(x, x, ops.ACT_RELU),
                 lambda: ops.tsa_temporal_backward(torch.randn(1, 3, 2, 4, 4), torch.randn(1, 2, 4, 4), torch.randn(1, 3, 2, 4, 4),
                                                   torch.randn(1, 3, 2, 4, 4))):
        with pytest.raises(NotImplementedError):
            call()


def test_input_size_assertions():
    from edvr_amd import EDVR
    with pytest.raises(AssertionError):
        EDVR(num_feat=16, num_reconstruct_block=1)(torch.rand(1, 5, 3, 18, 16))
    with pytest.raises(AssertionError):
        EDVR(num_feat=16, num_reconstruct_block=1, hr_in=True, with_predeblur=True)(torch.rand(1, 5, 3, 24, 32))


def test_module_attributes_mirror_reference():
    from edvr_amd import ModulatedDeformConv
    m = ModulatedDeformConv(8, 12, 3, stride=1, padding=1, dilation=1, groups=1, deformable_groups=2, bias=True)
    assert (m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation) == (8, 12, (3, 3), 1, 1, 1)
    assert m.with_bias and not m.transposed and m.output_padding == (0,)
    assert m.weight.shape == (12, 8, 3, 3) and torch.all(m.bias == 0)
    bound = 1.0 / (8 * 9) ** 0.5
    assert m.weight.abs().max().item() <= bound


def test_kernel_hints_from_offset_statistics():
    """The two performance hints derived from mean |offset| (results do not depend on them; GPU tests run every variant)."""
    from edvr_amd import functional as F_, ops
    assert F_.scatter_hint_from_absmean(None) == ops.DCN_SCATTER_LDS       # unknown field: the offset-independent strategy
    assert F_.scatter_hint_from_absmean(0.05) == ops.DCN_SCATTER_STRIP     # fresh conv_offset: sub-pixel, no scatter at all
    assert F_.scatter_hint_from_absmean(0.6) == ops.DCN_SCATTER_DEVICE
    assert F_.scatter_hint_from_absmean(3.0) == ops.DCN_SCATTER_LDS
    # forward: the per-tap-window kernel except on rough fields of about a pixel (R = 7 halo) or of tens of pixels (column buffer)
    T = ops.DCN_HALO_TAPWIN
    assert T == 16 and F_.halo_hint_from_stats(None, None) == T and F_.halo_hint_from_stats(0.4, 0.56) == T
    assert F_.halo_hint_from_stats(8.0, 0.2) == T and F_.halo_hint_from_stats(3.2, 4.5) == T and F_.halo_hint_from_stats(12.8, 18.0) == T
    assert F_.halo_hint_from_stats(0.8, 1.13) == 7 and F_.halo_hint_from_stats(51.0, 72.0) == -1
    assert F_.halo_hint_from_stats(0.4, None) == 3 and F_.halo_hint_from_stats(8.0, None) == -1
    # backward
    assert F_.scatter_hint_from_stats(8.0, 0.2) == ops.DCN_SCATTER_LDS_WIDE and F_.scatter_hint_from_stats(8.0, 1.5) == ops.DCN_SCATTER_LDS_WIDE  # (>= 4 px: the 6 px margin)
    assert F_.scatter_hint_from_stats(3.9, 0.2) == ops.DCN_SCATTER_LDS
    assert F_.scatter_hint_from_stats(0.1, 0.2) == ops.DCN_SCATTER_STRIP and F_.scatter_hint_from_stats(0.5, 0.17) == ops.DCN_SCATTER_STRIP
    assert F_.scatter_hint_from_stats(0.6, 1.0) == ops.DCN_SCATTER_DEVICE and F_.scatter_hint_from_stats(1.6, 0.17) == ops.DCN_SCATTER_LDS
    st = torch.tensor([[6.0, 2.0], [3.0, 0.0]])  # (2, n) sums of abs_stats_per_image over 16 elements in all
    assert ops.offset_stats(st, 16) == (0.5, 0.25) and ops.offset_stats(torch.tensor([[8.0], [-1.0]]), 16) == (0.5, None)
    assert (ops.DCN_SCATTER_AUTO, ops.DCN_SCATTER_DEVICE, ops.DCN_SCATTER_LDS, ops.DCN_SCATTER_STRIP, ops.DCN_SCATTER_LDS_WIDE) == (0, 1, 2, 3, 4)
    hdr = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'edvr_amd.h')).read()
    for name, val in (('AUTO', 0), ('DEVICE', 1), ('LDS', 2), ('STRIP', 3)):
        assert f'#define EDVR_DCN_SCATTER_{name} {val}' in hdr


def test_compat_ext_matches_the_reference_call_sites():
    """edvr_amd/compat/deform_conv_ext.py takes exactly the positional arguments the reference's Python passes to its pybind
    module at every call site of basicsr/models/ops/dcn/deform_conv.py (authoring container only: needs /root/reference)."""
    import ast
    import inspect
    import os
    src = '/root/reference/basicsr/models/ops/dcn/deform_conv.py'
    if not os.path.exists(src):
        pytest.skip('reference sources not present')
    from edvr_amd.compat import deform_conv_ext as ext
    calls = {}
    for node in ast.walk(ast.parse(open(src).read())):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and getattr(node.func.value, 'id', None) == 'deform_conv_ext':
            calls[node.func.attr] = len(node.args)
    assert sorted(calls) == ['deform_conv_backward_input', 'deform_conv_backward_parameters', 'deform_conv_forward',
                             'modulated_deform_conv_backward', 'modulated_deform_conv_forward']
    for name, n in calls.items():
        assert len(inspect.signature(getattr(ext, name)).parameters) == n, name
