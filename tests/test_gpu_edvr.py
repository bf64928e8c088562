"""-m gpu: whole-network parity of the HIP EDVR against the CPU oracle on identical weights/inputs."""
import pytest
import torch

from util_edvr import CONFIGS, build, oracle_kwargs

pytestmark = pytest.mark.gpu

INTERMEDIATE_RTOL = 2e-4   # fp32 HIP vs fp64 oracle on aligned / fused / trunk features, relative to max|ref|
PSNR_TOL_DB = 1e-3         # north_star: outputs within 1e-3 dB PSNR (fp32)


def _rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('algo', ['direct', 'winograd', 'auto'])  # auto: F(4x4) Winograd wherever it applies (w >= 32, w % 4 == 0)
@pytest.mark.parametrize('name', list(CONFIGS))
def test_edvr_forward_matches_oracle(gpu, name, algo, monkeypatch):
    from edvr_amd import ops
    from oracle import edvr_oracle as EO
    monkeypatch.setattr(ops, 'CONV_ALGO', {'direct': ops.CONV_DIRECT, 'winograd': ops.CONV_WINOGRAD, 'auto': ops.CONV_AUTO}[algo])
    net, x, kwargs = build(name)
    sd64 = {k: v.double() for k, v in net.state_dict().items()}
    taps_ref = {}
    with torch.no_grad():
        ref = EO.edvr_forward(sd64, x.double(), taps=taps_ref, **oracle_kwargs(kwargs))
        net = net.to(gpu)
        net.taps = {}
        out = net(x.to(gpu))
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    for key in ('aligned', 'fused', 'trunk'):
        assert _rel(net.taps[key].reshape(taps_ref[key].shape), taps_ref[key]) < INTERMEDIATE_RTOL, key
    assert _rel(out, ref) < INTERMEDIATE_RTOL
    gt = torch.rand(ref.shape, generator=torch.Generator().manual_seed(1))
    p_ours, p_ref = EO.psnr(out.cpu(), gt), EO.psnr(ref.float(), gt)
    assert abs(p_ours - p_ref) <= PSNR_TOL_DB, (p_ours, p_ref)


def test_pcd_and_tsa_public_forward(gpu):
    """The reference-style entry points: PCDAlignment.forward(nbr_l, ref_l), TSAFusion.forward(aligned)."""
    from edvr_amd import PCDAlignment, TSAFusion
    from oracle import edvr_oracle as EO
    from util_edvr import randomize_offsets
    torch.manual_seed(3)
    pcd = randomize_offsets(PCDAlignment(num_feat=64, deformable_groups=8)).eval()
    tsa = TSAFusion(num_feat=64, num_frame=3, center_frame_idx=1).eval()
    g = torch.Generator().manual_seed(2)
    nbr = [torch.randn(2, 64, 32 >> i, 48 >> i, generator=g) for i in range(3)]
    ref = [torch.randn(2, 64, 32 >> i, 48 >> i, generator=g) for i in range(3)]
    al = torch.randn(2, 3, 64, 16, 24, generator=g)
    with torch.no_grad():
        sd = {'p.' + k: v.double() for k, v in pcd.state_dict().items()}
        want = EO.pcd_align(sd, 'p.', [t.double() for t in nbr], [t.double() for t in ref], 8, EO.dcn_oracle.dcnv2_c)
        got = pcd.to(gpu)([t.to(gpu) for t in nbr], [t.to(gpu) for t in ref])
        assert _rel(got, want) < INTERMEDIATE_RTOL
        sd = {'f.' + k: v.double() for k, v in tsa.state_dict().items()}
        want = EO.tsa_fusion(sd, 'f.', al.double(), 1)
        got = tsa.to(gpu)(al.to(gpu))
        assert _rel(got, want) < INTERMEDIATE_RTOL


def _kernels_of(fn):
    """Names of the kernels edvr_amd.ops launches while fn() runs (the measurement hook of bench.py)."""
    from edvr_amd import ops
    names = []

    def hook(name, flops, launch, nbytes, executed=None):
        names.append(name)
        launch()

    ops.LAUNCH_HOOK = hook
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        ops.LAUNCH_HOOK = None
    return names


def test_default_paths_run_the_f4_kernel(gpu):
    """No silent fallback: with default settings the 3x3 convs of a no-grad forward AND of a training step (forward + data
    gradient) run on the split-operand F(4x4) kernel where it applies (here: the 32x48 level of EDVR-L / 7 frames), on the fp32
    F(4x4) kernel with that form switched off - and the bound of the input magnitude travels with the tensors: only a few inputs
    (outputs of kernels without the epilogue) need the reduction pass."""
    from edvr_amd import ops
    net, x, _ = build('L_T7')
    net = net.to(gpu)
    xg = x.to(gpu)
    with torch.no_grad():
        infer = _kernels_of(lambda: net(xg))
    n_split = infer.count('conv3x3_winograd_f4s_kernel')
    assert n_split >= 10 and infer.count('conv3x3_winograd_f4_kernel') == 0, sorted(set(infer))
    assert infer.count('amax') <= n_split // 2, (infer.count('amax'), n_split)
    prev = ops.set_f4s(inference=False, training=False)
    try:
        with torch.no_grad():
            infer32 = _kernels_of(lambda: net(xg))
        assert infer32.count('conv3x3_winograd_f4_kernel') == n_split and 'conv3x3_winograd_f4s_kernel' not in infer32
        net.train()
        train32 = _kernels_of(lambda: net(xg).sum().backward())
        assert train32.count('conv3x3_winograd_f4_kernel') >= 20, sorted(set(train32))  # forward + data gradient
    finally:
        ops.set_f4s(*prev)
    train = _kernels_of(lambda: net(xg).sum().backward())
    assert train.count('conv3x3_winograd_f4s_kernel') == train32.count('conv3x3_winograd_f4_kernel'), sorted(set(train))


def test_f4_switch_off(gpu):
    """EDVR_WINOGRAD_F4=0 / EDVR_WINOGRAD_F4_TRAIN=0 (read at import): F(2x2) / direct kernels only."""
    import os
    import subprocess
    import sys
    code = '''
import torch, sys
sys.path.insert(0, "tests")
from test_gpu_edvr import _kernels_of
from util_edvr import build
net, x, _ = build("L_T7")
net = net.cuda(); x = x.cuda()
with torch.no_grad():
    a = _kernels_of(lambda: net(x))
net.train()
b = _kernels_of(lambda: net(x).sum().backward())
assert "conv3x3_winograd_f4_kernel" not in a + b, sorted(set(a + b))
assert "conv3x3_winograd_kernel" in a and "conv3x3_winograd_kernel" in b
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EDVR_WINOGRAD_F4='0', EDVR_WINOGRAD_F4_TRAIN='0', PYTHONPATH=root),
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


def test_offset_check_never_waits_for_the_gpu(gpu, caplog, monkeypatch):
    """arch_util.py:248-253's `Offset abs mean is ..., larger than 50` warning: no host synchronisation in a no-grad forward
    or in a training forward (the statistics are examined when the next forward starts, on check_offsets() or on train() / eval());
    evaluated right away in grad mode only under EDVR_DCN_HINT_WAIT=1 (deterministic choice of the backward's dX strategy)."""
    import logging
    net, x, _ = build('M_T5')
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('cas_dcnpack.conv_offset.bias'):
                p.fill_(80.0)  # every offset of the cascade DCN ~80 px; its masks saturate, the output stays finite
    net = net.to(gpu)
    xg = x.to(gpu)
    cas = net.pcd_align.cas_dcnpack
    with caplog.at_level(logging.WARNING, logger='basicsr'):
        with torch.no_grad():
            out = net(xg)
        assert len(net._pending_offset_stats) == 1 and cas.last_offset_absmean is None  # nothing examined, nothing waited for
        assert not [r for r in caplog.records if 'larger than 50' in r.getMessage()]
        net.check_offsets()
        assert not net._pending_offset_stats and cas.last_offset_absmean > 50
        n_inf = len([r for r in caplog.records if 'larger than 50' in r.getMessage()])
        assert n_inf == x.shape[1]  # one per frame of the clip, like the reference's per-call check of that layer
        with torch.no_grad():
            net(xg)
            torch.cuda.synchronize()
            net(xg)  # the start of this forward examines the previous one (its copy has landed)
        assert len(net._pending_offset_stats) == 1
        assert len([r for r in caplog.records if 'larger than 50' in r.getMessage()]) == 2 * n_inf
        net.check_offsets()
        caplog.clear()
        net.train()  # (train() / eval() also flush whatever is pending: nothing here)
        net(xg).sum().backward()  # grad mode: queued as well - the backward picks its dX strategy from what has arrived
        assert len(net._pending_offset_stats) == 1
        assert not [r for r in caplog.records if 'larger than 50' in r.getMessage()]
        net.eval()  # flushes the queue
        assert not net._pending_offset_stats
        assert len([r for r in caplog.records if 'larger than 50' in r.getMessage()]) == n_inf
        caplog.clear()
        from edvr_amd import ops
        monkeypatch.setattr(ops, 'HINT_WAIT', True)  # EDVR_DCN_HINT_WAIT=1: evaluated inside the training forward
        net.train()
        net(xg).sum().backward()
        assert not net._pending_offset_stats
        assert len([r for r in caplog.records if 'larger than 50' in r.getMessage()]) == n_inf
    assert torch.isfinite(out).all()


def test_pending_offset_statistics_do_not_live_in_the_module(gpu, caplog):
    """The queue of not-yet-examined offset statistics (pinned tensors + torch.cuda.Event objects, which can be neither pickled nor
    deep-copied) is kept outside the module: an EMA copy or torch.save(net) right after an eval forward works, the copy starts with
    an empty queue, and train() / eval() flush the queue - the final clip's `larger than 50` warning is not lost."""
    import copy
    import io
    import logging
    net, x, _ = build('M_T5')
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('cas_dcnpack.conv_offset.bias'):
                p.fill_(80.0)
    net = net.to(gpu).eval()
    with caplog.at_level(logging.WARNING, logger='basicsr'):
        with torch.no_grad():
            net(x.to(gpu))
        assert len(net._pending_offset_stats) == 1 and '_pending_offset_stats' not in net.__dict__
        ema = copy.deepcopy(net)
        assert not ema._pending_offset_stats and len(net._pending_offset_stats) == 1
        torch.save(net, io.BytesIO())
        assert not [r for r in caplog.records if 'larger than 50' in r.getMessage()]
        net.train()  # flushes: the warning of the last (only) clip appears
        assert not net._pending_offset_stats
        assert len([r for r in caplog.records if 'larger than 50' in r.getMessage()]) == x.shape[1]
        cas = net.pcd_align.cas_dcnpack
        assert cas.last_offset_absmean > 50 and cas.last_offset_rough is not None and cas.last_offset_rough < 0.45


@pytest.mark.parametrize('cfg', ['M_T5', 'L_T7'])
def test_magnitude_bounds_reaching_the_split_kernel_are_bounds(gpu, cfg):
    """Every `x_amax` the split-operand conv kernel is given - measured by a producer's epilogue, derived through a layer's weight
    norms, carried through a gate / interpolation / pooling - is compared with the data (ops.BOUND_CHECK): never below max |x| (an
    overflow to infinity otherwise), and never more than 2^12 above it (the f16 pair keeps full accuracy down to 2^-18 of the bound)."""
    from edvr_amd import ops
    net, x, _ = build(cfg)
    net = net.to(gpu)
    xg = x.to(gpu)
    ops.BOUND_CHECK, ops.BOUND_CHECK_LOG[:] = True, []
    try:
        with torch.no_grad():
            y = net(xg)
        n_infer = len(ops.BOUND_CHECK_LOG)
        net.train()
        net(xg).sum().backward()
    finally:
        ops.BOUND_CHECK = False
    assert torch.isfinite(y).all()
    assert n_infer >= 10 and len(ops.BOUND_CHECK_LOG) >= 3 * n_infer - 4  # (forward + data gradient in training; the split 1x1 conv has no data-gradient form)
    worst = max(r for _, r in ops.BOUND_CHECK_LOG)
    assert worst < 4096.0, sorted(ops.BOUND_CHECK_LOG, key=lambda r: -r[1])[:5]


# ------------------------------------------------------------------------------------------------ the split path's guards
def test_forward_under_inference_mode(gpu):
    """inference tensors have no version counter; the bound bookkeeping must not read it."""
    from oracle import edvr_oracle as EO
    net, x, kwargs = build('M_T5')
    with torch.no_grad():
        ref = EO.edvr_forward(net.state_dict(), x, **oracle_kwargs(kwargs))
    net = net.to(gpu)
    with torch.inference_mode():
        out = net(x.to(gpu))
        out2 = net(x.to(gpu))  # (the second forward examines the first one's guard flag)
    assert _rel(out, ref.double()) < INTERMEDIATE_RTOL and torch.equal(out, out2)


def test_heavy_tailed_input_and_weights_stay_inside_the_bounds(gpu):
    """One 1e4 outlier pixel in the clip and every conv weight x30 (activations grow by orders of magnitude from layer to layer): every
    bound that reaches a split kernel still bounds its data (BOUND_CHECK compares each with the tensor), nothing overflows (the
    guard stays quiet) and the output matches the fp64 oracle at the whole-network tolerance."""
    from edvr_amd import ops
    from oracle import edvr_oracle as EO
    net, x, kwargs = build('M_T5')
    with torch.no_grad():
        for n, p in net.named_parameters():
            if p.dim() == 4 and 'conv_offset' not in n:
                p.mul_(30.0)
    x = x.clone()
    x[0, 2, 1, 7, 9] = 1e4
    sd64 = {k: v.double() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = EO.edvr_forward(sd64, x.double(), **oracle_kwargs(kwargs))
    net = net.to(gpu)
    trips = ops.GUARD_TRIPS
    ops.BOUND_CHECK, ops.BOUND_CHECK_LOG[:] = True, []
    try:
        with torch.no_grad():
            out = net(x.to(gpu))
    finally:
        ops.BOUND_CHECK = False
    ops.split_guard_check(wait=True)
    assert ops.GUARD_TRIPS == trips and len(ops.BOUND_CHECK_LOG) >= 10
    assert torch.isfinite(out).all() and ref.abs().max() > 1e6  # (the weights really did blow the activations up)
    assert _rel(out, ref) < INTERMEDIATE_RTOL


def test_batch_composition_changes_a_clip_only_within_the_conv_tolerance(gpu):
    """The split kernels scale by max |x| over the WHOLE (b t) tensor, so the low bits of clip A's output depend on which clips share
    its batch (the reference has no such coupling).  Clip A alone vs clip A beside a 100x brighter clip B: equal
    within the conv tolerance, and the brighter neighbour costs clip A no accuracy against the oracle."""
    from oracle import edvr_oracle as EO
    net, x, kwargs = build('M_T5')
    xb = torch.rand(x.shape, generator=torch.Generator().manual_seed(5)) * 100.0
    with torch.no_grad():
        ref = EO.edvr_forward({k: v.double() for k, v in net.state_dict().items()}, x.double(), **oracle_kwargs(kwargs))
    net = net.to(gpu)
    with torch.no_grad():
        alone = net(x.to(gpu))
        both = net(torch.cat([x, xb]).to(gpu))
    scale = ref.abs().max().item()
    assert (alone - both[:1]).abs().max().item() / scale < 3e-5  # the F(4x4) conv tolerance (tests/test_gpu_conv_f4s.py)
    assert _rel(alone, ref) < INTERMEDIATE_RTOL and _rel(both[:1], ref) < INTERMEDIATE_RTOL


def test_overflow_guard_catches_a_stale_bound(gpu):
    """A bound that is too small overflows the f16 operands to inf / NaN.  The conv's own y_amax slot keeps a non-finite maximum
    sticky, and the guard reads the slots without stalling the forward: the error surfaces at the next check."""
    from edvr_amd import ops
    ops.split_guard_check(wait=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 16, 64, generator=g).to(gpu)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.1).to(gpu)
    wpk, wf4s = ops.pack_conv_weight(w), ops.pack_conv_weight(w, f4s=True)
    ops.split_guard_submit(gpu)  # (whatever earlier tests left)
    ops.split_guard_check(wait=True)
    good = ops.conv2d(x, wpk, None, 64, 3, wpk_f4s=wf4s, x_amax=ops.amax(x), algo=ops.CONV_WINOGRAD_F4S)
    ops.split_guard_submit(gpu)
    ops.split_guard_check(wait=True)  # quiet
    assert torch.isfinite(good).all() and torch.isfinite(ops.get_bound(good)).all()
    stale = torch.full((1,), 1e-4, device=gpu)  # "max |x|" 1e4 times too small
    bad = ops.conv2d(x, wpk, None, 64, 3, wpk_f4s=wf4s, x_amax=stale, algo=ops.CONV_WINOGRAD_F4S)
    assert not torch.isfinite(bad).all()
    assert not torch.isfinite(ops.get_bound(bad)).all()  # sticky in the slot although v_max_f32 would have dropped the NaNs
    ops.split_guard_submit(gpu)
    with pytest.raises(ops.SplitOperandOverflow):
        ops.split_guard_check(wait=True)
    # the reduction kernel is sticky too: a consumer measuring a tensor with NaNs does not get a finite "bound"
    assert not torch.isfinite(ops.amax(bad)).all()
    ops.split_guard_submit(gpu)
    with pytest.raises(ops.SplitOperandOverflow):
        ops.split_guard_check(wait=True)
    # EDVR_SPLIT_GUARD=fallback: a warning, and the fp32 kernels from there on
    prev = (ops.F4S_INFERENCE, ops.F4S_TRAINING)
    ops.SPLIT_GUARD = 'fallback'
    try:
        ops.amax(bad)
        ops.split_guard_submit(gpu)
        with pytest.warns(UserWarning, match='fp32 kernels'):
            ops.split_guard_check(wait=True)
        assert (ops.F4S_INFERENCE, ops.F4S_TRAINING) == (False, False)
    finally:
        ops.SPLIT_GUARD = 'raise'
        ops.set_f4s(*prev)


def test_out_buffers_and_in_place_kernels_void_stale_bounds(gpu):
    """kernels write through raw pointers (no version bump), so every function that rewrites a caller's tensor must drop
    or replace the bound attached to it: dcnv2_forward(out=), dcnv1_forward(out=), upsample4x_add_, frame_reduce_add_."""
    from edvr_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, 8, 8, generator=g).to(gpu)
    off = (torch.randn(1, 18, 8, 8, generator=g) * 0.5).to(gpu)
    msk = torch.rand(1, 9, 8, 8, generator=g).to(gpu)
    w = torch.randn(8, 16, 3, 3, generator=g).to(gpu)
    out = torch.zeros(1, 8, 8, 8, device=gpu)
    ops.set_bound(out, torch.zeros(1, device=gpu))
    ops.dcnv2_forward(x, off, msk, w, None, 1, 1, 1, 1, 1, out=out)
    assert ops.get_bound(out) is None and out.abs().max() > 0
    ops.set_bound(out, torch.zeros(1, device=gpu))
    ops.dcnv1_forward(x, off, w, 1, 1, 1, 1, 1, out=out)
    assert ops.get_bound(out) is None
    y = torch.randn(1, 3, 16, 16, generator=g).to(gpu)
    base = torch.randn(1, 3, 4, 4, generator=g).to(gpu) * 5
    ops.set_bound(y, ops.amax(y))
    ops.upsample4x_add_(y, base)  # base has no bound: y must lose its own
    assert ops.get_bound(y) is None
    y2 = torch.randn(1, 3, 16, 16, generator=g).to(gpu)
    ops.set_bound(y2, ops.amax(y2))
    ops.set_bound(base, ops.amax(base))
    ops.upsample4x_add_(y2, base)
    assert float(ops.get_bound(y2)) >= float(y2.abs().max())  # replaced by bound(y) + bound(base)
    src = torch.randn(6, 4, 8, 8, generator=g).to(gpu)
    dst = torch.randn(6, 4, 8, 8, generator=g).to(gpu)
    ops.set_bound(dst, ops.amax(dst))
    ops.frame_reduce_add_(src, dst, 3, 1)
    assert ops.get_bound(dst) is None


def test_motion_like_offset_field_varies_in_space_and_matches_the_oracle(gpu):
    """bench.py's `motion` legs: structured clips + rescaled offset convs give every DCN layer offsets of a few
    pixels whose mean |horizontal neighbour difference| is ~0.5 px (a trained model's offsets follow objects, arch_util.py:243-257;
    white-noise frames with constant biases give 0.00-0.01) - and on that field the forward still equals the oracle."""
    from oracle import edvr_oracle as EO
    from util_edvr import motion_frames, motion_like_offsets
    kwargs = dict(num_feat=64, num_frame=5, num_reconstruct_block=4, center_frame_idx=2)
    torch.manual_seed(10)
    from edvr_amd import EDVR
    net = EDVR(**kwargs).eval().to(gpu)
    x = motion_frames(2, (5, 3, 64, 64), seed=3)
    stats = motion_like_offsets(net, x.to(gpu), target_rough=0.5, bias_sigma=3.0)
    for absmean, rough in stats:
        assert 0.1 < rough < 1.0, stats   # (white-noise frames + constant biases: 0.00-0.01)
        assert 1.0 < absmean < 7.0, stats  # single-digit displacements: the cap of motion_like_offsets
    sd64 = {k: v.double().cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = EO.edvr_forward(sd64, x.double(), **oracle_kwargs(kwargs))
        out = net(x.to(gpu))
    assert _rel(out, ref) < INTERMEDIATE_RTOL
