"""Shared helpers for whole-network parity tests."""
import torch

CONFIGS = {
    # name: (ctor kwargs, input shape (b, t, c, h, w))
    'M_T5': (dict(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2), (1, 5, 3, 32, 32)),
    'M_T5_b2_rect': (dict(num_feat=64, num_frame=5, num_reconstruct_block=4, center_frame_idx=2), (2, 5, 3, 24, 44)),
    'L_T7': (dict(num_feat=128, num_frame=7, num_reconstruct_block=4, center_frame_idx=None), (1, 7, 3, 32, 48)),
    'L_deblur_hr': (dict(num_feat=64, num_frame=5, num_reconstruct_block=4, center_frame_idx=2, hr_in=True,
                         with_predeblur=True), (1, 5, 3, 64, 64)),
    'M_noTSA': (dict(num_feat=32, num_frame=3, num_reconstruct_block=2, center_frame_idx=1, with_tsa=False,
                     deformable_groups=4), (2, 3, 3, 16, 16)),
}


def randomize_offsets(net, seed=123):
    """Default init zeroes conv_offset (deform_conv.py:377-381): every tap would sit on the integer grid
    and the bilinear gather would never be exercised (SURVEY.md finding 5)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('conv_offset.weight'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif n.endswith('conv_offset.bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    return net


def build(name, seed=10):
    from edvr_amd import EDVR
    kwargs, shape = CONFIGS[name]
    torch.manual_seed(seed)
    net = randomize_offsets(EDVR(**kwargs)).eval()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(0))
    return net, x, kwargs


def oracle_kwargs(kwargs):
    return dict(center=kwargs.get('center_frame_idx'), hr_in=kwargs.get('hr_in', False),
                with_predeblur=kwargs.get('with_predeblur', False), with_tsa=kwargs.get('with_tsa', True),
                dg=kwargs.get('deformable_groups', 8))
