"""Golden vectors (tests/golden/, made by oracle/make_golden.py from the reference's own code).
CPU: the oracle reproduces them.  GPU (-m gpu): the HIP path reproduces them."""
import glob
import os

import pytest
import torch

from oracle import dcn_oracle as O, edvr_oracle as EO
from util_edvr import CONFIGS, oracle_kwargs, randomize_offsets

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DCN_FILES = sorted(glob.glob(os.path.join(GOLD, 'dcn_*.pt')))
DCN1_FILES = sorted(glob.glob(os.path.join(GOLD, 'dcn1_*.pt')))
EDVR_FILES = sorted(glob.glob(os.path.join(GOLD, 'edvr_*.pt')))
GRADS = ('dx', 'doffset', 'dmask', 'dweight', 'dbias')


def _load(f):
    return torch.load(f, weights_only=False)


def test_fixtures_exist():
    assert len(DCN_FILES) >= 7 and len(EDVR_FILES) >= 3 and len(DCN1_FILES) >= 4


@pytest.mark.parametrize('f', DCN_FILES, ids=os.path.basename)
def test_oracle_reproduces_dcn_golden(f):
    d = _load(f)
    args = (d['x'], d['offset'], d['mask'], d['weight'])
    assert (O.c_forward(*args, d['bias'], *d['cfg']) - d['y']).abs().max().item() < 1e-12
    for name, got in zip(GRADS, O.c_backward(*args, d['dy'], True, *d['cfg'])):
        assert (got - d[name]).abs().max().item() <= 1e-12 * max(1.0, d[name].abs().max().item()), name
    t = [v.clone().requires_grad_() for v in (*args, d['bias'])]
    O.dcnv2_torch(*t, *d['cfg']).backward(d['dy'])
    for name, v in zip(GRADS, t):
        assert (v.grad - d[name]).abs().max().item() <= 1e-11 * max(1.0, d[name].abs().max().item()), name


@pytest.mark.parametrize('f', DCN1_FILES, ids=os.path.basename)
def test_oracle_reproduces_dcn1_golden(f):
    d = _load(f)
    # the plain-C restatement takes square geometries; rectangular stride / padding / dilation pairs (dcn1_rect.pt, from the
    # reference's own kernels like the others) pin the floor/gather restatement, the oracle of the rectangular GPU tests
    rect = isinstance(d['cfg'][0], tuple)
    fwd, bwd = (O.torch_dcn1_forward, O.torch_dcn1_backward) if rect else (O.c_dcn1_forward, O.c_dcn1_backward)
    assert (fwd(d['x'], d['offset'], d['weight'], *d['cfg']) - d['y']).abs().max().item() < 1e-12
    for name, got in zip(('dx', 'doffset', 'dweight'), bwd(d['x'], d['offset'], d['weight'], d['dy'], *d['cfg'])):
        assert (got - d[name]).abs().max().item() <= 1e-11 * max(1.0, d[name].abs().max().item()), name


def _rebuild(d):
    from edvr_amd import EDVR
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(**d['kwargs'])).eval()
    checksum = float(sum(p.detach().double().abs().sum() for p in net.parameters()))
    assert len(net.state_dict()) == d['n_keys'] and sum(p.numel() for p in net.parameters()) == d['n_params']
    assert abs(checksum - d['param_checksum']) <= 1e-9 * d['param_checksum']  # same seed -> same weights as the reference
    return net


@pytest.mark.parametrize('f', EDVR_FILES, ids=os.path.basename)
def test_oracle_reproduces_edvr_golden(f):
    d = _load(f)
    net = _rebuild(d)
    with torch.no_grad():
        y = EO.edvr_forward(net.state_dict(), d['x'], **oracle_kwargs(d['kwargs']))
    assert (y - d['y']).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('f', DCN_FILES, ids=os.path.basename)
def test_hip_reproduces_dcn_golden(gpu, f):
    from edvr_amd import ops
    d = _load(f)
    dev = [d[k].float().to(gpu) for k in ('x', 'offset', 'mask', 'weight', 'bias', 'dy')]
    y = ops.dcnv2_forward(*dev[:5], *d['cfg'])
    grads = ops.dcnv2_backward(*dev[:4], dev[5], True, *d['cfg'])
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
    assert rel(y, d['y']) < 2e-5
    for name, got in zip(GRADS, grads):
        assert rel(got, d[name]) < 1e-4, name


@pytest.mark.gpu
@pytest.mark.parametrize('f', DCN1_FILES, ids=os.path.basename)
def test_hip_reproduces_dcn1_golden(gpu, f):
    from edvr_amd import ops
    d = _load(f)
    x, off, w, dy = (d[k].float().to(gpu) for k in ('x', 'offset', 'weight', 'dy'))
    y = ops.dcnv1_forward(x, off, w, *d['cfg'])
    grads = ops.dcnv1_backward(x, off, w, dy, *d['cfg'])
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
    assert rel(y, d['y']) < 2e-5
    for name, got in zip(('dx', 'doffset', 'dweight'), grads):
        assert rel(got, d[name]) < 1e-4, name


@pytest.mark.gpu
@pytest.mark.parametrize('f', EDVR_FILES, ids=os.path.basename)
def test_hip_reproduces_edvr_golden(gpu, f):
    d = _load(f)
    net = _rebuild(d).to(gpu)
    with torch.no_grad():
        y = net(d['x'].to(gpu)).cpu()
    assert ((y - d['y']).abs().max() / d['y'].abs().max()).item() < 2e-4
    gt = torch.rand(d['y'].shape, generator=torch.Generator().manual_seed(1))
    assert abs(EO.psnr(y, gt) - EO.psnr(d['y'], gt)) <= 1e-3  # north_star: within 1e-3 dB PSNR
