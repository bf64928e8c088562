"""TEST INFRASTRUCTURE ONLY - CPU oracle of the whole EDVR forward, functional style.

A state_dict-driven restatement of the reference network in stock torch ops,
with the DCNv2 op supplied by oracle/dcn_oracle.py.  It exists because
/root/reference cannot travel to the GPU box: tests there compare the HIP path
against THIS, and tests here (tests/test_oracle_edvr.py) pin THIS against the
reference's own Python imported unchanged (oracle/ref_import.py) and against
tests/golden/.

Follows (paths relative to /root/reference/basicsr/models/archs/):
  edvr_arch.py:358-420  EDVR.forward            -> edvr_forward
  edvr_arch.py:72-117   PCDAlignment.forward    -> pcd_align
  edvr_arch.py:161-214  TSAFusion.forward       -> tsa_fusion
  edvr_arch.py:250-269  PredeblurModule.forward -> predeblur
  arch_util.py:92-95    ResidualBlockNoBN       -> resblock
  arch_util.py:243-257  DCNv2Pack.forward       -> dcn_pack

`sd` is a state_dict with the reference's key names; `taps`, if given, collects
named intermediates for parity checks beyond the final output.
"""
import torch
import torch.nn.functional as F

from . import dcn_oracle


# ---- test aid: "follow" another run of the same network at its DISCRETE decisions (see _follow below).  ReLU / LeakyReLU pick a
# side per element; the derivative jumps at 0, and a pre-activation smaller than the forward rounding error lands on different
# sides in an fp32 run and in this oracle (root cause of round 1's intermittent test failure, profiles/r2/flake_rootcause_*.log).
# One flip right below `aligned` moves every PCD gradient by ~1e-3.  With _ACT_SIDES = {layer name: bool tensor "the other run
# took the positive side"} every activation of this oracle takes the other run's side and keeps its own derivative there.
_ACT_SIDES = None
_FOLLOW_STATS = None  # list collecting, per followed decision point, how far the followed run is from this oracle's OWN values
_FRAME = None  # (frame index, frames per clip) while the per-frame PCD loop runs: the other run batches all frames of a clip


def _tag(y, name):
    y._edvr_name = name  # which layer produced this tensor (read by the activation that follows)
    return y


def _side(z, name):
    name = name or getattr(z, '_edvr_name', None)
    if _ACT_SIDES is None or name is None or name not in _ACT_SIDES:
        return None
    s = _ACT_SIDES[name]
    if s.shape[0] != z.shape[0]:  # (b * t, ...) in the other run, frame i of every clip here
        i, t = _FRAME
        s = s.reshape(z.shape[0], t, *s.shape[1:])[:, i]
    assert s.shape == z.shape, (name, tuple(s.shape), tuple(z.shape))
    if _FOLLOW_STATS is not None:  # elements where the followed run took the other side than this oracle would, and how close to
        with torch.no_grad():      # the kink this oracle's pre-activation is there (a rounding-level |z|, or a real forward bug?)
            flipped = (z > 0) != s
            nf = int(flipped.sum())
            band = float((z.abs() * flipped).max() / z.abs().max().clamp_min(1e-30)) if nf else 0.0
        _FOLLOW_STATS.append(('act', name, nf, z.numel(), band))
    return s


_CONV_IMPL = 'conv2d'  # 'unfold': im2col (F.unfold) + one GEMM per image instead of F.conv2d, see conv_unfold


def conv_unfold(x, w, b, stride, pad):
    """F.conv2d restated as im2col + GEMM in stock torch ops (F.unfold, torch.addmm -> rocBLAS on a GPU), one image at a time.
    For the GPU arm of the big full-size parity checks: MIOpen compiles a kernel per new convolution shape on a fresh box
    (minutes for the ~20 shapes of a 720x1280 -> 4K forward), unfold and the GEMM library do not."""
    co, ci, kh, kw = w.shape
    n, _, h, wd = x.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    wm = w.reshape(co, ci * kh * kw)
    out = x.new_empty(n, co, ho, wo)
    pointwise = kh == 1 and kw == 1 and stride == 1 and pad == 0
    xp = x if (pointwise or pad == 0) else F.pad(x, (pad, pad, pad, pad))
    rows = ho if pointwise else max(1, min(ho, (1 << 28) // (ci * kh * kw * wo)))  # column strips of <= 2^28 elements (1 GB)
    for i in range(n):
        o = out[i].view(co, ho * wo)
        for r0 in range(0, ho, rows):
            r1 = min(ho, r0 + rows)
            if pointwise:
                cols = x[i].reshape(ci, h * wd)
            else:
                cols = F.unfold(xp[i:i + 1, :, r0 * stride:(r1 - 1) * stride + kh], (kh, kw), stride=stride)[0]
            res = wm @ cols
            o[:, r0 * wo:r1 * wo] = res if b is None else res + b.view(co, 1)
    return out


def conv_unfold_autograd(x, w, b, stride, pad):
    """The same restatement for runs that need gradients (bench.py's training witness on the GPU): F.unfold over the whole batch +
    one batched GEMM, every op differentiable by torch autograd (fold / GEMM kernels - no MIOpen backward kernels to compile)."""
    co, ci, kh, kw = w.shape
    n, _, h, wd = x.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    cols = x.reshape(n, ci, h * wd) if (kh == 1 and kw == 1 and stride == 1 and pad == 0) else F.unfold(x, (kh, kw), padding=pad, stride=stride)
    out = torch.matmul(w.reshape(co, ci * kh * kw), cols)
    if b is not None:
        out = out + b.view(1, co, 1)
    return out.view(n, co, ho, wo)


def _conv(sd, name, x, stride=1, padding=None):
    w = sd[name + '.weight']
    pad = (w.shape[-1] // 2) if padding is None else padding
    if _CONV_IMPL == 'unfold':
        if torch.is_grad_enabled():
            return _tag(conv_unfold_autograd(x, w, sd.get(name + '.bias'), stride, pad), name)
        return _tag(conv_unfold(x, w, sd.get(name + '.bias'), stride, pad), name)
    return _tag(F.conv2d(x, w, sd.get(name + '.bias'), stride, pad), name)


def _lrelu(x, name=None):
    side = _side(x, name)
    return F.leaky_relu(x, 0.1) if side is None else torch.where(side, x, 0.1 * x)


def _relu(x, name=None):
    side = _side(x, name)
    return F.relu(x) if side is None else torch.where(side, x, torch.zeros_like(x))


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def resblock(sd, name, x):
    return x + _conv(sd, name + '.conv2', _relu(_conv(sd, name + '.conv1', x)))


def dcn_pack(sd, name, x, feat, dg, dcn, stats=None, forced_offset=None):
    """forced_offset (test aid, see _follow): the offsets ANOTHER run of the network fed to this DCN.  d/d(offset) is the
    one-sided derivative inside the cell floor() selects; where a sampling position sits within rounding error of an integer,
    an fp32 run and this oracle select different cells and the gradient jumps."""
    out = _conv(sd, name + '.conv_offset', feat)
    k3 = out.shape[1] // 3
    offset, mask = _follow(out[:, :2 * k3], forced_offset), torch.sigmoid(out[:, 2 * k3:])
    if stats is not None:
        stats.append(offset.abs().mean().item())
    return _tag(dcn(x, offset.contiguous(), mask.contiguous(), sd[name + '.weight'], sd.get(name + '.bias'), 1, 1, 1, 1, dg), name)


def pcd_align(sd, pre, nbr, ref, dg, dcn, stats=None, forced_offsets=None):
    """nbr/ref: [L1, L2, L3] feature lists, each (n, C, h, w).  forced_offsets: {'l3' | 'l2' | 'l1' | 'cas': offsets} or None."""
    fo = forced_offsets or {}
    up_off = up_feat = feat = None
    for lv in (3, 2, 1):
        L = f'l{lv}'
        off = _lrelu(_conv(sd, f'{pre}offset_conv1.{L}', torch.cat([nbr[lv - 1], ref[lv - 1]], 1)))
        if lv == 3:
            off = _lrelu(_conv(sd, f'{pre}offset_conv2.{L}', off))
        else:
            off = _lrelu(_conv(sd, f'{pre}offset_conv2.{L}', torch.cat([off, up_off], 1)))
            off = _lrelu(_conv(sd, f'{pre}offset_conv3.{L}', off))
        feat = dcn_pack(sd, f'{pre}dcn_pack.{L}', nbr[lv - 1], off, dg, dcn, stats, fo.get(L))
        if lv < 3:
            feat = _conv(sd, f'{pre}feat_conv.{L}', torch.cat([feat, up_feat], 1))
        if lv > 1:
            feat = _lrelu(feat)
            up_off = _up2(off) * 2
            up_feat = _up2(feat)
    off = torch.cat([feat, ref[0]], 1)
    off = _lrelu(_conv(sd, f'{pre}cas_offset_conv2', _lrelu(_conv(sd, f'{pre}cas_offset_conv1', off))))
    return _lrelu(dcn_pack(sd, f'{pre}cas_dcnpack', feat, off, dg, dcn, stats, fo.get('cas')))


def _follow(x, forced):
    """Value of `forced` (another run's tensor at this point), gradient of `x`: the discrete decisions taken downstream (which
    element a max-pool window routes its gradient to, which cell a deformable tap's floor() selects) then follow the other
    run, the derivative stays this run's."""
    if forced is None:
        return x
    forced = forced.to(x)
    if _FOLLOW_STATS is not None:  # the followed tensor against this oracle's own value of it: a forward parity check per stage
        with torch.no_grad():
            _FOLLOW_STATS.append(('value', None, 0, x.numel(), float((forced - x).abs().max() / x.abs().max().clamp_min(1e-30))))
    return x + (forced - x).detach()


def tsa_fusion(sd, pre, aligned, center, taps=None, pool_inputs=None):
    """pool_inputs (test aid): the two max-pool inputs of ANOTHER forward run of the same network.  MaxPool2d's gradient is
    discontinuous where two window elements tie; an fp32 run and this oracle can pick different elements when they differ by
    less than the forward rounding error, and one such flip at the 1/2- or 1/4-resolution attention maps moves every gradient
    upstream of the fusion module by ~1e-3 (profiles/r2/grad_bisect_L_T7_direct_vs_winograd.log).  With the other run's values
    substituted (straight-through: its values, this run's derivative) both take the same routing decisions."""
    b, t, c, h, w = aligned.shape
    emb_ref = _conv(sd, pre + 'temporal_attn1', aligned[:, center])
    emb = _conv(sd, pre + 'temporal_attn2', aligned.reshape(-1, c, h, w)).view(b, t, -1, h, w)
    prob = torch.sigmoid((emb * emb_ref.unsqueeze(1)).sum(2))  # (b, t, h, w)
    al = (aligned * prob.unsqueeze(2)).reshape(b, t * c, h, w)
    if taps is not None:
        taps['tsa_modulated'] = al
    feat = _lrelu(_conv(sd, pre + 'feat_fusion', al))
    attn = _follow(_lrelu(_conv(sd, pre + 'spatial_attn1', al)), pool_inputs[0] if pool_inputs else None)
    pooled = torch.cat([F.max_pool2d(attn, 3, 2, 1), F.avg_pool2d(attn, 3, 2, 1)], 1)
    attn = _lrelu(_conv(sd, pre + 'spatial_attn2', pooled))
    lvl = _follow(_lrelu(_conv(sd, pre + 'spatial_attn_l1', attn)), pool_inputs[1] if pool_inputs else None)
    pooled = torch.cat([F.max_pool2d(lvl, 3, 2, 1), F.avg_pool2d(lvl, 3, 2, 1)], 1)
    lvl = _lrelu(_conv(sd, pre + 'spatial_attn_l2', pooled))
    lvl = _up2(_lrelu(_conv(sd, pre + 'spatial_attn_l3', lvl)))
    attn = _lrelu(_conv(sd, pre + 'spatial_attn3', attn)) + lvl
    attn = _up2(_lrelu(_conv(sd, pre + 'spatial_attn4', attn)))
    attn = _conv(sd, pre + 'spatial_attn5', attn)
    attn_add = _conv(sd, pre + 'spatial_attn_add2', _lrelu(_conv(sd, pre + 'spatial_attn_add1', attn)))
    return feat * torch.sigmoid(attn) * 2 + attn_add


def predeblur(sd, pre, x, hr_in):
    f1 = _lrelu(_conv(sd, pre + 'conv_first', x))
    if hr_in:
        f1 = _lrelu(_conv(sd, pre + 'stride_conv_hr1', f1, 2))
        f1 = _lrelu(_conv(sd, pre + 'stride_conv_hr2', f1, 2))
    f2 = _lrelu(_conv(sd, pre + 'stride_conv_l2', f1, 2))
    f3 = _lrelu(_conv(sd, pre + 'stride_conv_l3', f2, 2))
    f3 = _up2(resblock(sd, pre + 'resblock_l3', f3))
    f2 = resblock(sd, pre + 'resblock_l2_1', f2) + f3
    f2 = _up2(resblock(sd, pre + 'resblock_l2_2', f2))
    for i in range(2):
        f1 = resblock(sd, f'{pre}resblock_l1.{i}', f1)
    f1 = f1 + f2
    for i in range(2, 5):
        f1 = resblock(sd, f'{pre}resblock_l1.{i}', f1)
    return f1


def _count(sd, prefix):
    idx = {int(k[len(prefix):].split('.')[0]) for k in sd if k.startswith(prefix)}
    return (max(idx) + 1) if idx else 0


def edvr_forward(sd, x, center=None, hr_in=False, with_predeblur=False, with_tsa=True, dg=8, dcn=None, taps=None,
                 stats=None, pool_inputs=None, dcn_offsets=None, act_sides=None, follow_stats=None, conv_impl='conv2d'):
    """x: (b, t, 3, h, w) -> (b, 3, 4h, 4w)  [or (b, 3, h, w) when hr_in].
    pool_inputs / dcn_offsets / act_sides: test aids, the discrete decisions of another run (see _follow, _ACT_SIDES).
    conv_impl: 'conv2d' (F.conv2d) or 'unfold' (im2col + GEMM: conv_unfold without gradients, conv_unfold_autograd with).
    follow_stats: a list; receives one record per followed decision point - ('act', layer, flipped elements, elements, largest
    |pre-activation| among the flipped ones relative to the layer's max) or ('value', None, 0, elements, max |followed - own| /
    max |own|) - so that a test can bound how far the followed run strays from this oracle's own forward pass."""
    global _ACT_SIDES, _FRAME, _FOLLOW_STATS, _CONV_IMPL
    _ACT_SIDES, _FOLLOW_STATS, _CONV_IMPL = act_sides, follow_stats, conv_impl
    try:
        return _edvr_forward(sd, x, center, hr_in, with_predeblur, with_tsa, dg, dcn, taps, stats, pool_inputs, dcn_offsets)
    finally:
        _ACT_SIDES, _FRAME, _FOLLOW_STATS, _CONV_IMPL = None, None, None, 'conv2d'


def _edvr_forward(sd, x, center, hr_in, with_predeblur, with_tsa, dg, dcn, taps, stats, pool_inputs, dcn_offsets):
    global _FRAME
    dcn = dcn or dcn_oracle.dcnv2_c
    b, t, c, h, w = x.shape
    center = t // 2 if center is None else center
    xc = x[:, center].contiguous()
    if with_predeblur:
        f1 = _conv(sd, 'conv_1x1', predeblur(sd, 'predeblur.', x.reshape(-1, c, h, w), hr_in))
        if hr_in:
            h, w = h // 4, w // 4
    else:
        f1 = _lrelu(_conv(sd, 'conv_first', x.reshape(-1, c, h, w)))
    for i in range(_count(sd, 'feature_extraction.')):
        f1 = resblock(sd, f'feature_extraction.{i}', f1)
    f2 = _lrelu(_conv(sd, 'conv_l2_2', _lrelu(_conv(sd, 'conv_l2_1', f1, 2))))
    f3 = _lrelu(_conv(sd, 'conv_l3_2', _lrelu(_conv(sd, 'conv_l3_1', f2, 2))))
    f1 = f1.view(b, t, -1, h, w)
    f2 = f2.view(b, t, -1, h // 2, w // 2)
    f3 = f3.view(b, t, -1, h // 4, w // 4)
    ref = [f1[:, center], f2[:, center], f3[:, center]]
    # dcn_offsets (test aid): {'l3' | 'l2' | 'l1' | 'cas': (b * t, 2 * dg * 9, h_l, w_l)} offsets of another run, frames batched
    # as image b_idx * t + t_idx (the layout edvr_amd's one-pass PCD alignment uses)
    def frame_offsets(i):
        return None if dcn_offsets is None else {k: v.reshape(b, t, *v.shape[1:])[:, i] for k, v in dcn_offsets.items()}
    frames = []
    for i in range(t):
        _FRAME = (i, t)
        frames.append(pcd_align(sd, 'pcd_align.', [f1[:, i], f2[:, i], f3[:, i]], ref, dg, dcn, stats, frame_offsets(i)))
    _FRAME = None
    aligned = torch.stack(frames, 1)
    if taps is not None:
        taps['aligned'] = aligned
    if with_tsa:
        feat = tsa_fusion(sd, 'fusion.', aligned, center, taps, pool_inputs)
    else:
        feat = _conv(sd, 'fusion', aligned.reshape(b, -1, h, w))
    if taps is not None:
        taps['fused'] = feat
    out = feat
    for i in range(_count(sd, 'reconstruction.')):
        out = resblock(sd, f'reconstruction.{i}', out)
    if taps is not None:
        taps['trunk'] = out
    out = _lrelu(F.pixel_shuffle(_conv(sd, 'upconv1', out), 2), 'upconv1')
    out = _lrelu(F.pixel_shuffle(_conv(sd, 'upconv2', out), 2), 'upconv2')
    out = _conv(sd, 'conv_last', _lrelu(_conv(sd, 'conv_hr', out)))
    base = xc if hr_in else F.interpolate(xc, scale_factor=4, mode='bilinear', align_corners=False)
    return out + base


def tensor2img_uint8(t):
    """basicsr/utils/img_util.py:67,93 restated: clamp [0,1] -> x255 -> round -> uint8 (layout kept CHW)."""
    return (t.detach().float().clamp(0, 1) * 255.0).round().to(torch.uint8)


def psnr_uint8(a, b):
    """basicsr/metrics/psnr_ssim.py:37-51 restated (crop_border 0, float64 MSE over all elements)."""
    mse = ((a.to(torch.float64) - b.to(torch.float64)) ** 2).mean().item()
    if mse == 0:
        return float('inf')
    import math
    return 20.0 * math.log10(255.0 / math.sqrt(mse))


def psnr(pred, gt):
    return psnr_uint8(tensor2img_uint8(pred), tensor2img_uint8(gt))


def psnr_cropped(pred, gt, crop_border=0, test_y_channel=False):
    """calculate_psnr(tensor2img(pred), tensor2img(gt), crop_border, 'HWC', test_y_channel) restated in NumPy for ONE image
    (c, h, w), RGB channel order (psnr_ssim.py:7-51, metric_util.py:34-47, matlab_functions.py:207-240)."""
    import math

    import numpy as np
    a = tensor2img_uint8(pred).numpy().astype(np.float64)
    b = tensor2img_uint8(gt).numpy().astype(np.float64)
    if crop_border:
        a, b = a[:, crop_border:-crop_border, crop_border:-crop_border], b[:, crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel and a.shape[0] == 3:
        def to_y(img):  # BGR of the reference = RGB reversed
            f = img.astype(np.float32) / 255.
            y = np.dot(np.moveaxis(f[::-1], 0, -1), [24.966, 128.553, 65.481]) + 16.0
            return (y / 255.).astype(np.float32) * 255.
        a, b = to_y(a), to_y(b)
    mse = np.mean((a - b) ** 2)
    return float('inf') if mse == 0 else float(20. * math.log10(255. / math.sqrt(mse)))


def ssim_cropped(pred, gt, crop_border=0, test_y_channel=False):
    """calculate_ssim(tensor2img(pred), tensor2img(gt), crop_border, 'HWC', test_y_channel) restated in NumPy for ONE image
    (c, h, w), RGB channel order (psnr_ssim.py:54-141): per channel, cv2.getGaussianKernel(11, 1.5) outer window correlated
    over the float64 images, valid region [5:-5, 5:-5], mean of the SSIM map; mean over channels."""
    import numpy as np
    a = tensor2img_uint8(pred).numpy().astype(np.float64)
    b = tensor2img_uint8(gt).numpy().astype(np.float64)
    if crop_border:
        a, b = a[:, crop_border:-crop_border, crop_border:-crop_border], b[:, crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel and a.shape[0] == 3:
        def to_y(img):  # to_y_channel: float32 / 255, BGR dot + 16, / 255 in float32, x 255 (metric_util.py:34-47)
            f = img.astype(np.float32) / 255.
            y = np.dot(np.moveaxis(f[::-1], 0, -1), [24.966, 128.553, 65.481]) + 16.0
            return ((y / 255.).astype(np.float32) * 255.)[None].astype(np.float64)
        a, b = to_y(a), to_y(b)
    k = np.exp(-0.5 / 1.5 ** 2 * (np.arange(11) - 5.0) ** 2)
    k = k / k.sum()
    win = np.outer(k, k)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2

    def filt(img):  # cv2.filter2D(img, -1, window)[5:-5, 5:-5] = plain correlation over the valid region
        v = np.lib.stride_tricks.sliding_window_view(img, (11, 11))
        return np.einsum('ijkl,kl->ij', v, win)

    vals = []
    for i in range(a.shape[0]):
        x, y = a[i], b[i]
        mu1, mu2 = filt(x), filt(y)
        s1, s2, s12 = filt(x * x) - mu1 ** 2, filt(y * y) - mu2 ** 2, filt(x * y) - mu1 * mu2
        vals.append((((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean())
    return float(np.array(vals).mean())


def charbonnier_sum(pred, target, eps=1e-12):
    """basicsr/models/losses/losses.py:23-25 with reduction='sum'."""
    return torch.sqrt((pred - target) ** 2 + eps).sum()
