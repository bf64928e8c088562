"""-m gpu: the training path.  Gradients of the fused HIP launches against autograd of the CPU oracle (fp64)."""
import pytest
import torch
import torch.nn.functional as F

from util_edvr import CONFIGS, build, oracle_kwargs

pytestmark = pytest.mark.gpu
GRAD_RTOL = 5e-4  # fp32 vs fp64, relative to max|ref grad| of each tensor


def _rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


CONV_CASES = [
    # n, c1, c2, h, w, co, ks, stride, act, nres, out_mode
    (2, 32, 0, 12, 20, 48, 3, 1, 'lrelu', 0, 0),
    (1, 24, 40, 9, 13, 32, 3, 1, 'relu', 1, 0),
    (2, 32, 0, 16, 12, 32, 3, 2, 'lrelu', 0, 0),
    (1, 32, 0, 15, 11, 40, 3, 2, 'none', 0, 0),
    (1, 96, 0, 10, 14, 32, 1, 1, 'lrelu', 1, 0),
    (1, 16, 0, 8, 12, 216, 3, 1, 'sigmoid_from', 0, 0),
    (1, 32, 0, 6, 10, 64, 3, 1, 'lrelu', 0, 1),
    (1, 130, 0, 7, 9, 140, 3, 1, 'none', 2, 0),
]


@pytest.fixture(params=['direct', 'winograd'])
def conv_algo(request):
    """Run the test with conv2d's default algorithm forced to each kernel (Winograd falls back where not applicable)."""
    from edvr_amd import ops
    ops.CONV_ALGO = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[request.param]
    yield request.param
    ops.CONV_ALGO = ops.CONV_AUTO


KINK_BAND = 1e-4  # |pre-activation| below this is "on the kink" of ReLU / LeakyReLU (fp32 forward error is ~1e-6 of max|z|)


def _move_off_the_kink(m64, xin, act_from):
    """Round-1 flake, root-caused (profiles/r2/flake_rootcause_*.log, scripts/flake_hunt.py): the derivative of ReLU / LeakyReLU
    jumps at z = 0; the HIP path takes it from the sign of its fp32 output, fp64 autograd from its own.  When a pre-activation is
    smaller than the forward rounding error the two pick different sides and ONE such element changes dz by 0.9 dy there (5e-2
    relative gradient error; seed 659 of 1500: |z| = 3.2e-8; the oracle run with the HIP forward's sides agrees to 3e-7).  The
    kernels themselves are bit-reproducible (profiles/r2/repeat_hunt_400x16.log).  A gradient comparison is only meaningful away
    from the kink, so the bias of any channel with a pre-activation inside KINK_BAND is nudged until none is left."""
    with torch.no_grad():
        for _ in range(50):
            near = m64(xin).abs() < KINK_BAND
            near[:, :act_from] = False
            ch = near.any(0).any(-1).any(-1)
            if not ch.any():
                return
            m64.bias[ch] += 3.7e-3
    raise AssertionError('could not move the test inputs off the activation kink')


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_gradients(gpu, case, conv_algo):
    """Forward + all gradients of one fused conv launch against fp64 autograd (strict: no retry)."""
    from edvr_amd import functional as F_
    n, c1, c2, h, w, co, ks, stride, actn, nres, out_mode = case
    torch.manual_seed(1234 + CONV_CASES.index(case))  # nn.Conv2d's init draws from the GLOBAL generator: pin it
    g = torch.Generator().manual_seed(11)
    m64 = torch.nn.Conv2d(c1 + c2, co, ks, stride, ks // 2).double()
    x1 = torch.randn(n, c1, h, w, generator=g)
    x2 = torch.randn(n, c2, h, w, generator=g) if c2 else None
    act, act_from = {'none': (0, 0), 'relu': (1, 0), 'lrelu': (2, 0), 'sigmoid_from': (3, 2 * co // 3)}[actn]
    ho, wo = (h + 2 * (ks // 2) - ks) // stride + 1, (w + 2 * (ks // 2) - ks) // stride + 1
    res = [torch.randn(n, co, ho, wo, generator=g) for _ in range(nres)]
    # oracle (fp64 autograd) on parameters that are exactly representable in fp32
    with torch.no_grad():
        for p_ in m64.parameters():
            p_.copy_(p_.float().double())
    leaves = [t.double().requires_grad_() for t in [x1] + ([x2] if c2 else []) + res]
    xin = torch.cat(leaves[:2], 1) if c2 else leaves[0]
    if actn in ('relu', 'lrelu'):
        _move_off_the_kink(m64, xin.detach(), act_from)
        with torch.no_grad():
            m64.bias.copy_(m64.bias.float().double())
    m = torch.nn.Conv2d(c1 + c2, co, ks, stride, ks // 2)
    m.load_state_dict({k: v.float() for k, v in m64.state_dict().items()})
    y = m64(xin)
    if actn == 'relu':
        y = F.relu(y)
    elif actn == 'lrelu':
        y = F.leaky_relu(y, 0.1)
    elif actn == 'sigmoid_from':
        y = torch.cat([y[:, :act_from], torch.sigmoid(y[:, act_from:])], 1)
    for r in leaves[(2 if c2 else 1):]:
        y = y + r
    if out_mode == 1:
        y = F.pixel_shuffle(y, 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    # HIP
    m = m.to(gpu)
    dev = [t.to(gpu).requires_grad_() for t in [x1] + ([x2] if c2 else []) + res]
    rs = dev[(2 if c2 else 1):]
    out = F_.conv(m, dev[0], x2=dev[1] if c2 else None, act=act, act_from=act_from, res1=rs[0] if nres > 0 else None,
                  res2=rs[1] if nres > 1 else None, out_mode=out_mode)
    e = _rel(out.detach(), y.detach())
    assert e < 2e-5, f'forward {e}'
    out.backward(dy.to(gpu))
    for i, (a, r) in enumerate(zip(dev, leaves)):
        e = _rel(a.grad, r.grad)
        assert e < GRAD_RTOL, f'grad of input {i}: {e}'
    e = _rel(m.weight.grad, m64.weight.grad)
    assert e < GRAD_RTOL, f'dweight {e}'
    e = _rel(m.bias.grad, m64.bias.grad)
    assert e < GRAD_RTOL, f'dbias {e}'


@pytest.mark.parametrize('shape', [(2, 64, 20, 36), (1, 128, 9, 23), (1, 48, 16, 18)])
@pytest.mark.parametrize('slope', [0.0, 0.1])
def test_conv_gate_epilogue(gpu, shape, slope):
    """conv2d(gate=g, gate_slope=s) == conv2d(...) * (g > 0 ? 1 : s): the activation backward fused into a data-gradient conv."""
    from edvr_amd import ops
    n, c, h, w = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, h, w, generator=g).to(gpu)
    wgt = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(gpu)
    gate = torch.randn(n, c, h, w, generator=g).relu().to(gpu)  # ~half exact zeros, like a saved ReLU output
    wpk = ops.pack_conv_weight(wgt)
    plain = ops.conv2d(x, wpk, None, c, 3)
    gated = ops.conv2d(x, wpk, None, c, 3, gate=gate, gate_slope=slope)
    ref = torch.where(gate > 0, plain, slope * plain)
    assert torch.equal(gated, ref)


@pytest.mark.parametrize('res', [False, True])
def test_conv_gate_on_the_direct_kernel(gpu, res):
    """Where the Winograd kernel does not apply (or is switched off) the direct kernel's generic store variant takes the gate."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 64, 9, 14, generator=g).to(gpu)
    wpk = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(gpu))
    gate = torch.randn(2, 64, 9, 14, generator=g).relu().to(gpu)
    r1 = torch.randn(2, 64, 9, 14, generator=g).to(gpu) if res else None
    plain = ops.conv2d(x, wpk, None, 64, 3, algo=ops.CONV_DIRECT)
    gated = ops.conv2d(x, wpk, None, 64, 3, gate=gate, gate_slope=0.1, res1=r1, algo=ops.CONV_DIRECT)
    ref = torch.where(gate > 0, plain, 0.1 * plain)
    assert torch.equal(gated, ref + r1 if res else ref)
    assert not ops.conv_gate_supported(2, 64, 9, 14, 64, ops.CONV_DIRECT) and ops.conv_gate_supported(2, 64, 20, 36, 64)


@pytest.mark.parametrize('algo', ['direct', 'winograd'])
@pytest.mark.parametrize('scale', [0.2, -1.5])
def test_conv_y_scale(gpu, algo, scale):
    """y = y_scale * (conv + bias) + res1 (+ gate): the scalar lives in the residual / gate epilogues of both kernels."""
    from edvr_amd import ops
    a = {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD}[algo]
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 64, 20, 36, generator=g).to(gpu)
    wpk = ops.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(gpu))
    b = torch.randn(64, generator=g).to(gpu)
    r1 = torch.randn(2, 64, 20, 36, generator=g).to(gpu)
    gate = torch.randn(2, 64, 20, 36, generator=g).relu().to(gpu)
    plain = ops.conv2d(x, wpk, b, 64, 3, algo=a)
    assert _rel(ops.conv2d(x, wpk, b, 64, 3, res1=r1, y_scale=scale, algo=a).cpu(), (scale * plain + r1).double().cpu()) < 1e-6
    assert _rel(ops.conv2d(x, wpk, b, 64, 3, y_scale=scale, algo=a).cpu(), (scale * plain).double().cpu()) < 1e-6
    got = ops.conv2d(x, wpk, b, 64, 3, gate=gate, gate_slope=0.1, y_scale=scale, algo=a)
    assert _rel(got.cpu(), (scale * torch.where(gate > 0, plain, 0.1 * plain)).double().cpu()) < 1e-6
    with pytest.raises(RuntimeError):  # 1x1 kernels take no scale
        ops.conv2d(x, ops.pack_conv_weight(torch.randn(64, 64, 1, 1, device=gpu)), None, 64, 1, y_scale=scale)


@pytest.mark.parametrize('shape', [(2, 64, 20, 36), (1, 128, 10, 24)])
@pytest.mark.parametrize('frozen', [False, True])
@pytest.mark.parametrize('res_scale', [1, 0.2])
def test_residual_block_fused_backward(gpu, shape, frozen, res_scale):
    """ResidualBlockNoBN as one autograd node (ReLU backward + identity gradient fused into the dgrad launches) vs fp64 autograd;
    res_scale != 1 (arch_util.py:95) rides in the same epilogues."""
    from edvr_amd.arch_util import ResidualBlockNoBN
    n, c, h, w = shape
    g = torch.Generator().manual_seed(17)
    blk = ResidualBlockNoBN(c, res_scale=res_scale)
    for p_ in blk.parameters():
        p_.data = torch.randn(p_.shape, generator=g) * 0.05
    x = torch.randn(n, c, h, w, generator=g)
    dy = torch.randn(n, c, h, w, generator=g)
    w1, b1, w2, b2 = [p_.detach().double().requires_grad_() for p_ in (blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias)]
    x64 = x.double().requires_grad_()
    y64 = x64 + res_scale * F.conv2d(F.relu(F.conv2d(x64, w1, b1, padding=1)), w2, b2, padding=1)
    y64.backward(dy.double())
    blk = blk.to(gpu)
    if frozen:  # TSA-only phase: trunk weights frozen, gradient still flows to the input
        for p_ in blk.parameters():
            p_.requires_grad_(False)
    xd = x.to(gpu).requires_grad_()
    y = blk(xd)
    assert type(y.grad_fn).__name__.startswith('ResBlockFn')
    assert _rel(y.detach(), y64.detach()) < 2e-5
    y.backward(dy.to(gpu))
    assert _rel(xd.grad, x64.grad) < GRAD_RTOL
    if frozen:
        assert all(p_.grad is None for p_ in blk.parameters())
    else:
        for ours, ref in [(blk.conv1.weight, w1), (blk.conv1.bias, b1), (blk.conv2.weight, w2), (blk.conv2.bias, b2)]:
            assert _rel(ours.grad, ref.grad) < GRAD_RTOL


@pytest.mark.parametrize('res_scale', [1, 0.3])
def test_residual_block_unfused_path(gpu, res_scale):
    """Sizes the Winograd kernel does not take (w <= 16): two ConvFn nodes on the direct kernel, res_scale through ConvFn."""
    from edvr_amd.arch_util import ResidualBlockNoBN
    g = torch.Generator().manual_seed(18)
    blk = ResidualBlockNoBN(32, res_scale=res_scale)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
    x, dy = torch.randn(2, 32, 10, 12, generator=g), torch.randn(2, 32, 10, 12, generator=g)
    ps = [p_.detach().double().requires_grad_() for p_ in blk.parameters()]
    x64 = x.double().requires_grad_()
    y64 = x64 + res_scale * F.conv2d(F.relu(F.conv2d(x64, ps[0], ps[1], padding=1)), ps[2], ps[3], padding=1)
    y64.backward(dy.double())
    blk = blk.to(gpu)
    xd = x.to(gpu).requires_grad_()
    y = blk(xd)
    assert not type(y.grad_fn).__name__.startswith('ResBlockFn')
    assert _rel(y.detach(), y64.detach()) < 2e-5
    y.backward(dy.to(gpu))
    assert _rel(xd.grad, x64.grad) < GRAD_RTOL
    for ours, ref in zip(blk.parameters(), ps):
        assert _rel(ours.grad, ref.grad) < GRAD_RTOL


def test_training_backward_with_the_winograd_kernel_switched_off():
    """EDVR_CONV_WINOGRAD=0 (the documented fallback switch) is read once per process: a child process trains one residual block
    on the direct kernels only (round-1 advisor finding: the fused ReLU-backward gate used to raise there)."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    code = (
        'import torch, torch.nn.functional as F\n'
        'from edvr_amd.arch_util import ResidualBlockNoBN\n'
        'from edvr_amd import ops\n'
        'assert not ops.conv_gate_supported(2, 64, 20, 36, 64)\n'
        'torch.manual_seed(3)\n'
        'blk = ResidualBlockNoBN(64); x = torch.randn(2, 64, 20, 36); dy = torch.randn(2, 64, 20, 36)\n'
        'ps = [p.detach().double().requires_grad_() for p in blk.parameters()]\n'
        'x64 = x.double().requires_grad_()\n'
        '(x64 + F.conv2d(F.relu(F.conv2d(x64, ps[0], ps[1], padding=1)), ps[2], ps[3], padding=1)).backward(dy.double())\n'
        'blk = blk.cuda(); xd = x.cuda().requires_grad_(); blk(xd).backward(dy.cuda())\n'
        'rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max()).item()\n'
        'errs = [rel(xd.grad, x64.grad)] + [rel(p.grad, r.grad) for p, r in zip(blk.parameters(), ps)]\n'
        'assert max(errs) < 5e-4, errs\n'
        'print("ok", max(errs))\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EDVR_CONV_WINOGRAD='0', PYTHONPATH=root), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


def test_conv_gradient_with_reference_frame_map(gpu):
    """x2 = the same tensor read through the clip-centre image map: its gradient is summed over the clip's frames."""
    from edvr_amd import functional as F_
    g = torch.Generator().manual_seed(12)
    b, t, c, h, w, ctr = 2, 3, 32, 10, 12, 1
    m = torch.nn.Conv2d(2 * c, 32, 3, 1, 1)
    feat = torch.randn(b * t, c, h, w, generator=g)
    dy = torch.randn(b * t, 32, h, w, generator=g)
    f64 = feat.double().requires_grad_()
    ref_in = torch.cat([f64, f64.view(b, t, c, h, w)[:, ctr:ctr + 1].expand(b, t, c, h, w).reshape(b * t, c, h, w)], 1)
    m64 = torch.nn.Conv2d(2 * c, 32, 3, 1, 1).double()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    m64(ref_in).backward(dy.double())
    m = m.to(gpu)
    fg = feat.to(gpu).requires_grad_()
    F_.conv(m, fg, x2=fg, x2_map=(t, t, ctr)).backward(dy.to(gpu))
    assert _rel(fg.grad, f64.grad) < GRAD_RTOL
    assert _rel(m.weight.grad, m64.weight.grad) < GRAD_RTOL


def test_glue_gradients(gpu):
    from edvr_amd import functional as F_
    g = torch.Generator().manual_seed(13)
    # upsample x2 (scale 2), pooling, tsa temporal, tsa combine
    x = torch.randn(2, 5, 9, 7, generator=g)
    x64 = x.double().requires_grad_()
    dy = torch.randn(2, 5, 18, 14, generator=g)
    (F.interpolate(x64, scale_factor=2, mode='bilinear', align_corners=False) * 2).backward(dy.double())
    xg = x.to(gpu).requires_grad_()
    F_.upsample2x(xg, 2.0).backward(dy.to(gpu))
    assert _rel(xg.grad, x64.grad) < 1e-5

    x = torch.randn(2, 4, 11, 8, generator=g)
    x64 = x.double().requires_grad_()
    y64 = torch.cat([F.max_pool2d(x64, 3, 2, 1), F.avg_pool2d(x64, 3, 2, 1)], 1)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    xg = x.to(gpu).requires_grad_()
    F_.pool_maxavg(xg).backward(dy.to(gpu))
    assert _rel(xg.grad, x64.grad) < 1e-5

    b, t, c, h, w = 2, 5, 16, 6, 9
    emb, al = torch.randn(b, t, c, h, w, generator=g) * 0.3, torch.randn(b, t, c, h, w, generator=g)
    er = torch.randn(b, c, h, w, generator=g) * 0.3
    leaves = [v.double().requires_grad_() for v in (emb, er, al)]
    out = leaves[2] * torch.sigmoid((leaves[0] * leaves[1].unsqueeze(1)).sum(2)).unsqueeze(2)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy.double())
    dev = [v.to(gpu).requires_grad_() for v in (emb, er, al)]
    F_.tsa_temporal(*dev).backward(dy.to(gpu))
    for a, r in zip(dev, leaves):
        assert _rel(a.grad, r.grad) < 1e-4

    a3 = [torch.randn(2, 6, 5, 7, generator=g) for _ in range(3)]
    l3 = [v.double().requires_grad_() for v in a3]
    dy = torch.randn(2, 6, 5, 7, generator=g)
    (l3[0] * torch.sigmoid(l3[1]) * 2 + l3[2]).backward(dy.double())
    d3 = [v.to(gpu).requires_grad_() for v in a3]
    F_.tsa_combine(*d3).backward(dy.to(gpu))
    for a, r in zip(d3, l3):
        assert _rel(a.grad, r.grad) < 1e-5


def test_charbonnier(gpu):
    from edvr_amd.autograd import charbonnier_loss
    from oracle import edvr_oracle as EO
    g = torch.Generator().manual_seed(14)
    p, t = torch.rand(2, 3, 20, 24, generator=g), torch.rand(2, 3, 20, 24, generator=g)
    p64 = p.double().requires_grad_()
    l64 = EO.charbonnier_sum(p64, t.double())
    l64.backward()
    pg = p.to(gpu).requires_grad_()
    l = charbonnier_loss(pg, t.to(gpu))
    l.backward()
    assert abs(l.item() - l64.item()) / l64.item() < 1e-5
    assert _rel(pg.grad, p64.grad) < 1e-5


@pytest.mark.parametrize('name', ['M_T5', 'L_deblur_hr', 'M_noTSA', 'L_T7'])
def test_edvr_parameter_gradients_match_oracle(gpu, name, conv_algo):
    _followed_gradient_check(gpu, name)


def test_edvr_l_full_depth_parameter_gradients_match_oracle(gpu):
    """BASELINE configs[3] at its real depth and crop size - EDVR-L as trained (128 features, 40 reconstruction blocks, 5 frames,
    64x64 LR crop, one clip) - on the default kernel selection (F(4x4) forward / data gradient, Winograd-domain weight gradient,
    fused DCN backward): every parameter gradient against the fp64 CPU oracle following the HIP run's discrete decisions, same
    bounds as the small stand-ins above."""
    _followed_gradient_check(gpu, 'L_full_T5')


@pytest.mark.parametrize('name,hint_wait', [('M_T5_b2_rect', True), ('M_T5_b2_rect', False), ('L_T7', True)])
def test_edvr_parameter_gradients_with_multi_pixel_offsets(gpu, name, hint_wait, monkeypatch):
    """The same whole-network check on the offsets of a TRAINED model (`conv_offset.bias ~ N(0, 4^2)` per channel: per-tap
    displacements of several pixels, the `trained_like` legs of bench.py): the forward runs on the tap-window kernel where the level
    is wide enough (44 / 48 columns: L1; the narrower levels on the halo / column kernels), the backward on the dX strategies
    the statistics of THIS forward select (hint_wait, the deterministic mode: LDS window / device atomics) or on the default
    strategy a first iteration gets (statistics not yet examined)."""
    from edvr_amd import ops
    monkeypatch.setattr(ops, 'HINT_WAIT', hint_wait)
    launches = []

    def hook(name_, flops, launch, *rest):
        launches.append(name_)
        launch()

    monkeypatch.setattr(ops, 'LAUNCH_HOOK', hook)
    _followed_gradient_check(gpu, name, bias_sigma=4.0)
    assert any('dcn_tapwin' in n for n in launches), sorted(set(launches))  # (fp32 or split-operand form of the tap-window kernel)


def _followed_gradient_check(gpu, name, bias_sigma=0.5):
    """Whole network: d(Charbonnier sum)/d(every parameter), HIP fp32 vs oracle autograd fp64.
    The bound is calibrated per tensor against the fp32 noise floor of the oracle itself (same algorithm in fp32 on the
    CPU vs fp64): ours must be within max(1e-3, 4 x that floor) of the fp64 truth, relative to max|grad|.

    EDVR is only piecewise differentiable: ReLU / LeakyReLU sides, the element a max-pool window routes to, the cell floor()
    picks for a deformable tap.  An fp32 run and the fp64 oracle (or two fp32 kernels) decide differently wherever a value sits
    within the forward rounding error of a boundary, and the gradients then differ by a finite jump, not by rounding: EDVR-L /
    T7 under the direct kernel had ONE LeakyReLU flip right below `aligned`, which moved every PCD gradient by ~1e-3 (median over
    tensors 4.3e-4 vs 2e-6 for the Winograd run of the same network; scripts/grad_bisect.py, scripts/grad_taps.py,
    profiles/r2/grad_bisect_L_T7_direct_vs_winograd.log).  The oracle therefore takes the HIP run's discrete decisions
    (tests/util_edvr.py::DecisionRecorder -> edvr_oracle._follow / _ACT_SIDES) and computes its own fp64 derivative on them:
    what is compared is the derivative arithmetic of every kernel, on identical branches."""
    from edvr_amd.autograd import charbonnier_loss
    from oracle import dcn_oracle as O, edvr_oracle as EO
    from util_edvr import DecisionRecorder
    net, x, kwargs = build(name, bias_sigma=bias_sigma)
    net.train()
    state = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(gpu)
    with DecisionRecorder(net) as rec:
        out = net(x.to(gpu))
    rec.bind(net)
    assert len(rec.pool_inputs) == (2 if kwargs.get('with_tsa', True) else 0) and len(rec.oms) == 4 and len(rec.sides) > 20

    def oracle_grads(dt, follow_stats=None):
        sd = {k: v.to(dt).requires_grad_() for k, v in state.items()}
        o = EO.edvr_forward(sd, x.to(dt), dcn=O.dcnv2_c, follow_stats=follow_stats, **rec.oracle_kwargs(), **oracle_kwargs(kwargs))
        gt = torch.rand(o.shape, generator=torch.Generator().manual_seed(1))
        EO.charbonnier_sum(o, gt.to(dt)).backward()
        return o.detach(), gt, {k: v.grad for k, v in sd.items()}

    fstats = []
    out64, gt, g64 = oracle_grads(torch.float64, fstats)
    _, _, g32 = oracle_grads(torch.float32)
    assert _rel(out.detach(), out64) < 2e-4
    # Following must not be able to hide a forward bug: the HIP run's decisions are held against the oracle's OWN at every point
    # they are taken over.  (a) the followed tensors (DCN offsets, max-pool inputs) equal the oracle's own values of them to
    # forward-parity tolerance; (b) activation sides differ from the oracle's on a vanishing fraction of the elements, and only
    # where the oracle's pre-activation is itself within rounding distance of the kink.
    assert len(fstats) > 20
    vals = [s for s in fstats if s[0] == 'value']
    acts = [s for s in fstats if s[0] == 'act']
    assert vals and max(s[4] for s in vals) < 2e-4, max(s[4] for s in vals)
    flipped, total = sum(s[2] for s in acts), sum(s[3] for s in acts)
    assert flipped <= 1e-5 * total + 2, (flipped, total)
    assert max(s[4] for s in acts) < KINK_BAND, max(acts, key=lambda s: s[4])
    charbonnier_loss(out, gt.to(gpu)).backward()
    # What is left after sharing the decisions: the fp32 ORACLE itself differs from fp64 by up to ~6e-3 on the conv_offset
    # tensors (sampling positions are formed in the compute precision), so each tensor is held to the calibrated bound
    # max(1e-3, 4 x its own fp32 floor) - ALL of them (round 1 allowed 5 % outliers) - with a tight median.
    ours_all, within = [], 0
    worst = (0.0, 0.0, '')
    for k, p in net.named_parameters():
        ref = g64[k]
        assert p.grad is not None, k
        if ref.abs().max() == 0:
            assert p.grad.abs().max().item() == 0, k
            continue
        ours, floor = _rel(p.grad, ref), _rel(g32[k], ref)
        ours_all.append(ours)
        within += ours < max(1e-3, 4 * floor)
        if ours > worst[0]:
            worst = (ours, floor, k)
        assert ours < 2e-2, (k, ours, floor)
    ours_all.sort()
    assert ours_all[len(ours_all) // 2] < 5e-5, ours_all[len(ours_all) // 2]
    assert within == len(ours_all), (within, len(ours_all), worst)
    print(f'{name}: median {ours_all[len(ours_all) // 2]:.1e}; worst {worst[0]:.2e} (fp32-oracle floor {worst[1]:.2e}) at {worst[2]}; '
          f'{within}/{len(ours_all)} tensors within max(1e-3, 4 x floor)')


@pytest.mark.parametrize('name', ['M_T5', 'L_T7', 'L_full_T5'])
def test_edvr_parameter_gradients_unfollowed(gpu, name):
    """The same comparison with NOTHING shared: the fp64 oracle takes its own activation sides, pooling routes and DCN cells.
    A handful of elements may then sit on different sides of a kink in the two runs and shift individual tensors by a finite
    jump (see the test above), so only the forward output and the MEDIAN over the parameter tensors are bounded - loose enough
    for such flips, tight enough that a wrong derivative anywhere upstream (which moves most tensors) cannot pass."""
    from edvr_amd.autograd import charbonnier_loss
    from oracle import dcn_oracle as O, edvr_oracle as EO
    net, x, kwargs = build(name)
    net.train()
    sd = {k: v.detach().double().requires_grad_() for k, v in net.state_dict().items()}
    o = EO.edvr_forward(sd, x.double(), dcn=O.dcnv2_c, **oracle_kwargs(kwargs))
    gt = torch.rand(o.shape, generator=torch.Generator().manual_seed(1))
    EO.charbonnier_sum(o, gt.double()).backward()
    net = net.to(gpu)
    out = net(x.to(gpu))
    assert _rel(out.detach(), o.detach()) < 2e-4
    charbonnier_loss(out, gt.to(gpu)).backward()
    errs = sorted(_rel(p.grad, sd[k].grad) for k, p in net.named_parameters() if sd[k].grad.abs().max() > 0)
    median, p90 = errs[len(errs) // 2], errs[int(len(errs) * 0.9)]
    print(f'{name} unfollowed: median {median:.1e}, 90th percentile {p90:.1e}, max {errs[-1]:.1e} over {len(errs)} tensors')
    assert median < 1e-3, (median, p90, errs[-1])
    assert errs[-1] < 0.2, errs[-1]


def test_training_trajectory_matches_oracle_adam(gpu):
    """Ten optimizer steps, not one (sr_model.py:88-112 in a loop), three arms from identical weights on identical batches (two
    alternating batches of two clips, EDVR-M with 4 reconstruction blocks, the reference's lr 4e-4 / betas (0.9, 0.99), both
    dcn_lr_mul groups):
      O  the fp64 CPU oracle + torch.optim.Adam                                   (the absolute anchor)
      A  the product path: HIP network + edvr_amd.optim.FusedAdam, every cache live
      B  the HIP network + torch.optim.Adam with the packed-weight cache dropped and the kernel hints reset before EVERY forward
    What only a trajectory catches is state that outlives an iteration: the packed-weight cache (ops.prepack_conv_weights rewrites
    its buffers in place after every step), the per-layer kernel hints, FusedAdam's multi-tensor table and moments.
    The sharp detector is the LOSS CURVE: a forward pass on weights that are one step stale leaves it by ~1e-3 in the first
    iterations.  Bounds: A vs O 1e-4 relative at every iteration, A vs B 1e-5 (same kernels up to the hint-selected variants: ~10
    ulps of an fp32 sum measured).
    The final weights are compared through the accumulated UPDATE, per tensor e = || (w - w_0) - (w_ref - w_0) || / || w_ref - w_0 ||.
    Adam divides by sqrt(v): an element whose gradient is small against its tensor's largest moves by a sizeable fraction of lr per
    step on rounding-level differences, and ten steps on random targets amplify those to per cent: the fp32 CPU oracle itself ends
    5 % (median over tensors) to 10.5 % (worst) away from the fp64 one on this very trajectory (2.8e-3 = seven steps of lr at single
    elements), and the two HIP arms - which differ only by FusedAdam's ulps, the order of the dX atomics and the hint-selected kernel
    variants - 7.6 % (worst).  A 1e-4 max-norm bound on the weights is therefore not what this optimizer preserves; asserted are
    median e < 15 %, worst e < 40 % (both pairs), and that no element is further apart than the 2 * lr * steps two copies can
    drift at all.  Wrong moments, a wrong step count or lr, or a group that is not updated give e of order 1."""
    from edvr_amd import EDVR, ops
    from edvr_amd.autograd import charbonnier_loss
    from edvr_amd.optim import FusedAdam
    from oracle import dcn_oracle as O, edvr_oracle as EO
    from util_edvr import randomize_offsets
    kwargs = dict(num_feat=64, num_frame=5, num_reconstruct_block=4, center_frame_idx=2)
    steps, lr = 10, 4e-4
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(**kwargs)).train()
    state0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    batches = [(torch.rand(2, 5, 3, 32, 32, generator=g), torch.rand(2, 3, 128, 128, generator=g)) for _ in range(2)]
    # arm O
    sd = {k: v.detach().double().clone().requires_grad_() for k, v in state0.items()}
    opt64 = torch.optim.Adam(list(sd.values()), lr=lr, betas=(0.9, 0.99))
    loss64 = []
    for it in range(steps):
        x, gt = batches[it % 2]
        opt64.zero_grad()
        l = EO.charbonnier_sum(EO.edvr_forward(sd, x.double(), dcn=O.dcnv2_c, **oracle_kwargs(kwargs)), gt.double())
        l.backward()
        opt64.step()
        loss64.append(l.item())
    assert abs(loss64[2] - loss64[0]) / loss64[0] > 1e-3, 'the test needs a loss that moves'
    dev = [(x.to(gpu), gt.to(gpu)) for x, gt in batches]

    def hip_arm(fused):
        torch.manual_seed(10)
        m = randomize_offsets(EDVR(**kwargs)).train().to(gpu)
        dcn = [p for n, p in m.named_parameters() if 'dcn' in n]
        rest = [p for n, p in m.named_parameters() if 'dcn' not in n]
        groups = [{'params': rest, 'lr': lr}, {'params': dcn, 'lr': lr}]
        opt = (FusedAdam if fused else torch.optim.Adam)(groups, lr=lr, betas=(0.9, 0.99))
        losses = []
        for it in range(steps):
            x, gt = dev[it % 2]
            if not fused:  # arm B: nothing survives an iteration but the weights and torch's own optimizer state
                ops.invalidate_packed_weights()
                for d in m.pcd_align.dcn_modules():
                    d.last_offset_absmean = d.last_offset_rough = None
            opt.zero_grad(set_to_none=True)
            l = charbonnier_loss(m(x), gt)
            l.backward()
            opt.step()
            losses.append(l.item())
        return losses, {k: p.detach().double().cpu() for k, p in m.named_parameters()}

    loss_a, w_a = hip_arm(True)
    loss_b, w_b = hip_arm(False)
    w0 = {k: v.detach().double() for k, v in state0.items()}

    def update_errors(w, ref):
        out = []
        for k in w:
            du = (ref[k] - w0[k]).norm().item()
            assert du > 0, k
            out.append((((w[k] - w0[k]) - (ref[k] - w0[k])).norm().item() / du, k))
        return sorted(out)

    # A vs B: the state that outlives an iteration
    assert max(abs(a - b) / b for a, b in zip(loss_a, loss_b)) < 1e-5, (loss_a, loss_b)  # (measured 1.2e-6: ~10 ulps of an fp32 sum)
    ab = update_errors(w_a, w_b)
    # A vs O: the arithmetic
    curve = [abs(a - b) / b for a, b in zip(loss_a, loss64)]
    assert max(curve) < 1e-4, (curve, loss_a, loss64)
    w_o = {k: v.detach() for k, v in sd.items()}
    ao = update_errors(w_a, w_o)
    for k in w_a:
        assert (w_a[k] - w_o[k]).abs().max().item() <= 2.1 * lr * steps, (k, (w_a[k] - w_o[k]).abs().max().item())
    print(f'trajectory: loss rel err vs oracle {max(curve):.1e}; update error vs oracle median {ao[len(ao) // 2][0]:.2e}, worst {ao[-1][0]:.2e} '
          f'at {ao[-1][1]}; cached vs uncached HIP arms: median {ab[len(ab) // 2][0]:.2e}, worst {ab[-1][0]:.2e} at {ab[-1][1]}')
    for name, errs in (('A vs B', ab), ('A vs O', ao)):
        assert errs[len(errs) // 2][0] < 0.15 and errs[-1][0] < 0.40, (name, errs[len(errs) // 2], errs[-3:])


def test_steady_state_training_reuses_packed_buffers_and_the_job_table(gpu):
    """ops.prepack_conv_weights rewrites a stale packed layout IN PLACE when the cache is its only owner (`_sole_owner`: a reference
    count and torch's private use count - interpreter / torch-version dependent; when they disagree the code takes the safe branch
    and silently allocates ~480 fresh buffers and a new job table every iteration).  This is the assertion that the fast branch is the
    one that runs: from the second optimizer step on, every packed buffer keeps its address and the ONE job table is reused."""
    from edvr_amd import EDVR, ops
    from edvr_amd.autograd import charbonnier_loss
    from edvr_amd.optim import FusedAdam
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    ops.invalidate_packed_weights()
    ops._PACK_TABLES.clear()
    m = randomize_offsets(EDVR(num_feat=64, num_frame=5, num_reconstruct_block=4, center_frame_idx=2)).train().to(gpu)
    opt = FusedAdam(m.parameters(), lr=4e-4, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(7)
    x, gt = torch.rand(2, 5, 3, 32, 32, generator=g).to(gpu), torch.rand(2, 3, 128, 128, generator=g).to(gpu)

    def snapshot():
        ptrs = {(wid, key): hit[1].data_ptr() for wid, ent in ops._PACKED.items() for key, hit in ent[1].items()}
        return ptrs, {k: v[0].data_ptr() for k, v in ops._PACK_TABLES.items()}

    snaps = []
    for it in range(4):
        opt.zero_grad(set_to_none=True)
        charbonnier_loss(m(x), gt).backward()
        opt.step()
        snaps.append(snapshot())
    torch.cuda.synchronize()
    assert len(snaps[1][0]) > 100, 'the network should have filled the cache'
    for later in snaps[2:]:
        assert later[0] == snaps[1][0], 'packed buffers were re-allocated in steady state'
        assert later[1] == snaps[1][1] and len(later[1]) <= 2, 'the job table was rebuilt in steady state'  # (<= 2: the first iteration's, which also allocates, may differ)
