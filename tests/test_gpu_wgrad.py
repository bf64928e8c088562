"""-m gpu: weight gradient of the fused conv through the C ABI (edvr_conv2d_wgrad_f32): the Winograd-domain kernel and the
direct kernel against torch's fp64 conv2d_weight on the CPU (tolerance: 2e-5 of max |dW|, fp32 accumulation over <= 1e5 pixels)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, c1, c2, h, w, co, x2_map
    (4, 64, 0, 32, 32, 64, None),
    (2, 128, 0, 16, 48, 128, None),      # 24 tiles per row: 3 chunks
    (3, 64, 64, 20, 36, 96, None),       # concat input, 18 tiles per row (phantom tiles in the last chunk), partial co block
    (2, 48, 0, 8, 8, 48, None),          # partial ci and co blocks
    (1, 64, 0, 4, 6, 64, None),          # 3 chunks: odd count, masked tail chunk
    (6, 64, 64, 16, 16, 64, (3, 1, 0)),  # x2 is a broadcast reference frame: image i of x2 is i // 3
    (5, 192, 0, 12, 20, 216, None),      # 3 ci blocks x 4 co blocks (the offset-conv shape)
    (32, 128, 0, 64, 64, 128, None),     # EDVR-L training trunk layer at full size
    (2, 64, 0, 20, 44, 3, None),         # auto: <= 4 output channels -> VALU kernel (conv_last), ragged 4 x 32 tiles
    (3, 100, 0, 7, 9, 4, None),          # auto: VALU kernel, two 64-channel blocks (second partial), tiny image
    (8, 64, 0, 256, 256, 3, None),       # conv_last at the training resolution
]


@pytest.mark.parametrize('algo', ['winograd', 'direct', 'auto'])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'n%d_c%d+%d_%dx%d_co%d%s' % (c[0], c[1], c[2], c[3], c[4], c[5], '_map' if c[6] else ''))
def test_wgrad_matches_fp64_reference(gpu, case, algo):
    from edvr_amd import ops
    n, c1, c2, h, w, co, x2_map = case
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + co)
    x1 = torch.randn(n, c1, h, w, generator=g)
    n2 = n if x2_map is None else n // x2_map[0]
    x2 = torch.randn(n2, c2, h, w, generator=g) if c2 else None
    dz = torch.randn(n, co, h, w, generator=g)
    if c2:
        x2_full = x2 if x2_map is None else x2[(torch.arange(n) // x2_map[0]) * x2_map[1] + x2_map[2]]
        xcat = torch.cat([x1, x2_full], 1)
    else:
        xcat = x1
    ref = torch.nn.grad.conv2d_weight(xcat.double(), (co, c1 + c2, 3, 3), dz.double(), padding=1)
    prev = ops.set_wgrad_algo({'winograd': ops.CONV_WINOGRAD, 'direct': ops.CONV_DIRECT, 'auto': ops.CONV_AUTO}[algo])
    try:
        dw, db = ops.conv2d_wgrad(x1.to(gpu), x2.to(gpu) if c2 else None, x2_map, dz.to(gpu), co, 3, 1, want_db=True)
        dw2 = ops.conv2d_wgrad(x1.to(gpu), x2.to(gpu) if c2 else None, x2_map, dz.to(gpu), co, 3, 1)
    finally:
        ops.set_wgrad_algo(prev)
    assert torch.equal(dw, dw2), 'split-K reduction must be deterministic'
    err = (dw.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, err
    ref_db = dz.double().sum((0, 2, 3))  # the bias gradient produced by the same launches
    assert (db.cpu().double() - ref_db).abs().max().item() <= 2e-5 * max(1.0, ref_db.abs().max().item()) + 1e-6 * (n * h * w) ** 0.5


def test_wgrad_algo_setter_rejects_unknown(gpu):
    from edvr_amd import ops
    with pytest.raises(Exception):
        ops.set_wgrad_algo(17)
    assert ops.set_wgrad_algo(ops.CONV_AUTO) == ops.CONV_AUTO


@pytest.mark.parametrize('shape', [(32, 128, 64, 64), (5, 128, 16, 16), (3, 216, 12, 20), (2, 64, 9, 7), (160, 128, 64, 64), (2, 3, 40, 40)])
def test_channel_sum_bias_gradient(gpu, shape):
    """db = sum over (n, h, w): single-stage kernel (aligned, <= 1M elements per channel) and the two-stage one."""
    from edvr_amd import ops
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[1]))
    ref = x.double().sum((0, 2, 3))
    got = ops.channel_sum(x.to(gpu)).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) + 1e-3 * (shape[0] * shape[2] * shape[3]) ** 0.5 * 1e-3
    sl = x.to(gpu)[:, 1:shape[1] - 1] if shape[1] > 8 else None  # channel-sliced view: image stride != c * hw
    if sl is not None:
        assert torch.allclose(ops.channel_sum(sl).cpu().double(), ref[1:-1], rtol=0, atol=2e-5 * max(1.0, ref.abs().max().item()) + 1e-4)


def test_direct_split_weight_gradient_opt_in(gpu):
    """csrc/wgrad_direct_s.hip (EDVR_WGRAD_DIRECT_SPLIT=1, read once per process: a child process): every `auto` case of this file - it
    takes the ones with w % 4 == 0, concat inputs, the frame map and partial channel blocks included - at the same tolerance."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, EDVR_WGRAD_DIRECT_SPLIT='1')
    code = ('import ctypes, sys; sys.path.insert(0, "tests"); from edvr_amd import _lib; L = _lib.lib(); '
            'assert L.edvr_conv2d_wgrad_split_is_direct(64, 64) == 1 and L.edvr_conv2d_wgrad_split_is_direct(7, 9) == 0; '
            'import pytest; sys.exit(pytest.main(["-q", "-x", "tests/test_gpu_wgrad.py", "-k", "matches_fp64_reference and auto"]))')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout
