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
    # EDVR-L AS TRAINED (options/train/EDVR/train_EDVR_L_x4_SR_REDS.yml:21-32: 128 features, 40 reconstruction blocks, 5 frames,
    # 64x64 LR crops): BASELINE configs[3] at its real depth and size, one clip
    'L_full_T5': (dict(num_feat=128, num_frame=5, num_reconstruct_block=40, center_frame_idx=None), (1, 5, 3, 64, 64)),
}


def randomize_offsets(net, seed=123, bias_sigma=0.5):
    """Default init zeroes conv_offset (deform_conv.py:377-381): every tap would sit on the integer grid
    and the bilinear gather would never be exercised (SURVEY.md finding 5).  bias_sigma: spread of the per-channel constants -
    0.5 = sub-pixel taps (a lightly trained model), 4 / 10 = the multi-pixel per-tap displacements of a trained one (the mask
    channels keep 0.5: they go through a sigmoid)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith('conv_offset.weight'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif n.endswith('conv_offset.bias'):
                b = torch.randn(p.shape, generator=g) * 0.5
                if bias_sigma != 0.5:
                    b[:2 * p.shape[0] // 3] *= bias_sigma / 0.5  # offset channels only (same random draws)
                p.copy_(b)
    return net


def build(name, seed=10, bias_sigma=0.5):
    from edvr_amd import EDVR
    kwargs, shape = CONFIGS[name]
    torch.manual_seed(seed)
    net = randomize_offsets(EDVR(**kwargs), bias_sigma=bias_sigma).eval()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(0))
    return net, x, kwargs


def oracle_kwargs(kwargs):
    return dict(center=kwargs.get('center_frame_idx'), hr_in=kwargs.get('hr_in', False),
                with_predeblur=kwargs.get('with_predeblur', False), with_tsa=kwargs.get('with_tsa', True),
                dg=kwargs.get('deformable_groups', 8))


class DecisionRecorder:
    """Records the DISCRETE decisions of one HIP forward pass of `net` so that the fp64 oracle can take the same ones
    (oracle/edvr_oracle.py: _follow, _ACT_SIDES): the side every ReLU / LeakyReLU element landed on, the inputs of the two
    max-pools, the offsets fed to the four DCNs.  A gradient comparison is only defined away from those discontinuities; with
    the decisions shared, fp32-vs-fp64 differences are rounding only.  Usage: `with DecisionRecorder(net) as rec: out = net(x)`,
    then EO.edvr_forward(..., **rec.oracle_kwargs())."""

    def __init__(self, net):
        self.names = {id(m): n for n, m in net.named_modules()}
        self.sides, self.pool_inputs, self.oms = {}, [], []

    def __enter__(self):
        from edvr_amd import autograd as ag, functional as F_
        self._F, self._ag = F_, ag
        self._saved = (F_.conv, F_.dcn_from_packed, F_.pool_maxavg, F_.offset_mask_conv_stats, ag.ResBlockFn.forward)
        conv, dcn, pool, omc, rbf = self._saved
        rec = self

        def conv_w(m, x, **k):
            y = conv(m, x, **k)
            if k.get('act', F_.ACT_NONE) in (F_.ACT_RELU, F_.ACT_LRELU):
                v = y.detach()
                for r in (k.get('res1'), k.get('res2')):  # y = act(z) + res: the backward kernel decides on y - res1 - res2
                    if r is not None:
                        v = v - r.detach()
                rec.sides[rec.names[id(m)]] = (v > 0).cpu()
            return y

        def dcn_w(m, x, om, act=F_.ACT_NONE):
            y = dcn(m, x, om, act)
            if act in (F_.ACT_RELU, F_.ACT_LRELU):
                rec.sides[rec.names[id(m)]] = (y.detach() > 0).cpu()
            return y

        def pool_w(t):
            rec.pool_inputs.append(t.detach().cpu())
            return pool(t)

        def omc_w(m, f):
            om, sums = omc(m, f)
            rec.oms.append(om.detach().cpu())
            return om, sums

        def rbf_w(ctx, x, w1, b1, w2, b2, res_scale=1.0):  # fused residual block: the hidden ReLU output is only in ctx
            y = rbf(ctx, x, w1, b1, w2, b2, res_scale)
            rec._resblock_h.append((w1.data_ptr(), ctx.to_save[1].detach()))
            return y

        self._resblock_h = []
        F_.conv, F_.dcn_from_packed, F_.pool_maxavg, F_.offset_mask_conv_stats = conv_w, dcn_w, pool_w, omc_w
        ag.ResBlockFn.forward = staticmethod(rbf_w)
        self._net_params = None
        return self

    def bind(self, net):
        """call after the forward pass: resolves the fused residual blocks' names from their conv1 weights"""
        by_ptr = {p.data_ptr(): n for n, p in net.named_parameters()}
        for ptr, h in self._resblock_h:
            self.sides[by_ptr[ptr][:-len('.weight')]] = (h > 0).cpu()
        return self

    def __exit__(self, *exc):
        F_, ag = self._F, self._ag
        F_.conv, F_.dcn_from_packed, F_.pool_maxavg, F_.offset_mask_conv_stats = self._saved[:4]
        ag.ResBlockFn.forward = staticmethod(self._saved[4])
        return False

    def oracle_kwargs(self):
        offs = {k: om[:, :2 * om.shape[1] // 3] for k, om in zip(('l3', 'l2', 'l1', 'cas'), self.oms)}
        return dict(pool_inputs=self.pool_inputs or None, dcn_offsets=offs or None, act_sides=self.sides)


def motion_frames(batch, shape, seed=0):
    """Synthetic clips with image STRUCTURE instead of white noise: a smooth background (low-resolution noise, bilinearly enlarged) that
    drifts by (1.5, 0.75) px per frame, rectangles of constant intensity moving at their own speeds over it (object boundaries), and a
    little pixel noise (texture, +-0.06).  shape = (t, c, h, w); values in [0, 1].  The offset convs of a network see features that are smooth
    inside objects and jump at their edges - the spatial statistics of a trained model's offsets (arch_util.py:243-257), which follow
    objects, where torch.rand frames give spatially constant offsets + noise."""
    import torch.nn.functional as F
    t, c, h, w = shape
    g = torch.Generator().manual_seed(1000 + seed)
    pad = 16
    bg = torch.rand(batch, c, (h + 2 * pad) // 16 + 2, (w + 2 * pad) // 16 + 2, generator=g)
    bg = F.interpolate(bg, size=(h + 2 * pad, w + 2 * pad), mode='bilinear', align_corners=False)
    ys = torch.arange(h, dtype=torch.float32).view(1, h, 1).expand(batch, h, w)
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, w).expand(batch, h, w)
    n_obj = max(2, (h * w) // 2500)
    ox, oy = torch.rand(batch, n_obj, generator=g) * w, torch.rand(batch, n_obj, generator=g) * h
    ow, oh = 6 + torch.rand(batch, n_obj, generator=g) * w / 6, 6 + torch.rand(batch, n_obj, generator=g) * h / 4
    vx, vy = (torch.rand(batch, n_obj, generator=g) - 0.5) * 6, (torch.rand(batch, n_obj, generator=g) - 0.5) * 4
    col = torch.rand(batch, n_obj, c, generator=g)
    frames = []
    for f in range(t):
        gx = (xs + pad + 1.5 * (f - t // 2)) / (w + 2 * pad - 1) * 2 - 1
        gy = (ys + pad + 0.75 * (f - t // 2)) / (h + 2 * pad - 1) * 2 - 1
        img = F.grid_sample(bg, torch.stack([gx, gy], -1), mode='bilinear', align_corners=True)
        for k in range(n_obj):
            cx, cy = (ox[:, k] + vx[:, k] * (f - t // 2)).view(-1, 1, 1), (oy[:, k] + vy[:, k] * (f - t // 2)).view(-1, 1, 1)
            inside = ((xs - cx).abs() < ow[:, k].view(-1, 1, 1) / 2) & ((ys - cy).abs() < oh[:, k].view(-1, 1, 1) / 2)
            img = torch.where(inside.unsqueeze(1), col[:, k].view(batch, c, 1, 1).expand(-1, -1, h, w), img)
        frames.append(img)
    x = torch.stack(frames, 1) + (torch.rand(batch, t, c, h, w, generator=g) - 0.5) * 0.12
    return x.clamp_(0, 1)


def motion_like_offsets(net, x, target_rough=0.5, bias_sigma=3.0, rounds=4, seed=123, absmean_cap=6.0):
    """Give `net` (on the GPU) the offset statistics of a trained model on the clips `x`: per-tap displacements of a few pixels
    (conv_offset.bias ~ N(0, bias_sigma^2)) PLUS a spatially varying part that follows the input's structure - conv_offset.weight is
    rescaled, per DCN module, until the mean |horizontal neighbour difference| of its offsets on `x` is `target_rough` px (the
    difference is linear in the weights; the cascade of the pyramid makes a few rounds necessary) - but never so far that the mean
    |offset| of the layer exceeds `absmean_cap` px (trained EDVRs stay in single digits; the reference warns at 50,
    arch_util.py:248-253): where the cap binds the field stays smoother than the target.  Returns the per-module
    (mean |offset|, roughness) of the final state."""
    randomize_offsets(net, seed=seed, bias_sigma=bias_sigma)
    dcns = net.pcd_align.dcn_modules()

    def measure():
        with torch.no_grad():
            net(x)
        net.check_offsets()
        return [(m.last_offset_absmean, m.last_offset_rough) for m in dcns]
    for _ in range(rounds):
        stats = measure()
        with torch.no_grad():
            for m, (absmean, rough) in zip(dcns, stats):
                if rough and rough > 0:
                    n_off = 2 * m.conv_offset.weight.shape[0] // 3
                    b0 = float(m.conv_offset.bias[:n_off].abs().mean())  # the constant part's share of the mean |offset|
                    part = max(absmean * absmean - b0 * b0, 1e-8) ** 0.5   # ... and (roughly) the spatially varying part's
                    room = max(absmean_cap * absmean_cap - b0 * b0, 0.0) ** 0.5
                    f = min(max(target_rough / rough, 0.05), 200.0, max(room / part, 0.05))
                    m.conv_offset.weight[:n_off].mul_(f)
    return measure()
