"""Module-level functional layer: nn.Conv2d parameter holders -> fused HIP launches.

`conv(m, x, ...)` runs an nn.Conv2d's parameters through the fp32 MFMA kernel with the
surrounding elementwise ops of the reference graph fused in (activation, residual adds,
channel concat of a second input, PixelShuffle).  When gradients are required the same
launches are recorded as autograd Functions (edvr_amd/autograd.py).
"""
import torch

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, OUT_NCHW, OUT_PIXEL_SHUFFLE2  # noqa: F401


def _conv_geometry(m):
    ks, st = m.kernel_size, m.stride
    if (ks[0] != ks[1] or ks[0] not in (1, 3) or st[0] != st[1] or st[0] not in (1, 2) or (ks[0] == 1 and st[0] != 1)
            or tuple(m.padding) != (ks[0] // 2, ks[0] // 2) or tuple(m.dilation) != (1, 1) or m.groups != 1):
        raise NotImplementedError(
            f'edvr_amd conv kernel supports 3x3 (stride 1/2, pad 1) and 1x1 convolutions, dilation 1, groups 1; got {m}')
    return ks[0], st[0]


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def conv(m, x, *, x2=None, x2_map=None, act=ACT_NONE, act_from=0, res1=None, res2=None, out_mode=OUT_NCHW, y_scale=1.0,
         abs_sum_channels=0):
    """y_scale * act(conv(cat(x, x2)) + bias) + res1 + res2 with the parameters of nn.Conv2d `m`.
    abs_sum_channels > 0 (no-grad calls only): returns (y, per-image sums of |y[:, :abs_sum_channels]|), ops.conv2d."""
    ks, stride = _conv_geometry(m)
    if x.dim() != 4:
        raise ValueError(f'expected a 4-D input, got {tuple(x.shape)}')
    cin = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    if cin != m.in_channels:
        raise RuntimeError(f'expected {m.in_channels} input channels, got {cin}')
    ops.require_gpu(x, x2, m.weight)
    if _needs_grad(x, x2, m.weight, m.bias, res1, res2):
        from . import autograd as ag
        assert abs_sum_channels == 0
        return ag.conv(m, x, x2, x2_map, act, act_from, res1, res2, out_mode, ks, stride, y_scale)
    wpk = ops.pack_conv_weight(m.weight)
    # the F(4x4,3x3) Winograd weights let the C side pick that kernel where it is the fastest (2.25 instead of 4 multiplies per
    # output, rounding ~1e-6 of the output scale instead of ~2e-7; EDVR_WINOGRAD_F4=0 switches it off here,
    # EDVR_WINOGRAD_F4_TRAIN=0 in the training path above)
    f4 = ops.F4_INFERENCE and ks == 3 and stride == 1 and m.in_channels >= 32 and m.out_channels >= 48
    # ... and the split-operand form of the same algorithm (csrc/winograd_f4s.hip: fp32 operands as f16 (hi, lo) pairs on the f16
    # matrix pipe, all four cross products, fp32 accumulation - the fp32 kernel's accuracy at 1.3-1.5x its speed; the bound of the
    # input's magnitude it needs travels with the tensors, ops.set_bound / input_bound); EDVR_WINOGRAD_F4S=0 keeps the fp32 kernel
    f4s = f4 and ops.F4S_INFERENCE and x.shape[3] % 4 == 0
    wf4 = ops.pack_conv_weight(m.weight, f4=True) if (f4 and not f4s) else None
    wf4s = ops.pack_conv_weight(m.weight, f4s=True) if f4s else None
    if ops.F4S_INFERENCE and ks == 1 and stride == 1 and cin >= 320 and x.shape[1] % 8 == 0 and cin % 8 == 0:
        wf4s = ops.pack_conv_weight(m.weight, f4s=True)  # the streaming 1x1 kernel's split form (csrc/conv1x1_s.hip; the C side decides)
    bias = m.bias.detach() if m.bias is not None else None
    r = ops.conv2d(x, wpk, bias, m.out_channels, ks, x2=x2, x2_map=x2_map, stride=stride, act=act, act_from=act_from,
                   res1=res1, res2=res2, out_mode=out_mode, y_scale=y_scale, wpk_f4=wf4, abs_sum_channels=abs_sum_channels,
                   wpk_f4s=wf4s)
    y = r[0] if abs_sum_channels > 0 else r
    if ops.F4S_INFERENCE and ops.get_bound(y) is None:  # a kernel without the y_amax epilogue: the bound from the weights' norms
        ops.linear_bound(y, m.weight, m.bias, (x, x2), (res1, res2), scale=y_scale, floor=1.0 if act == ACT_SIGMOID else 0.0)
    return r


def offset_mask_conv_stats(conv_offset, feat):
    """offset_mask_conv + the per-image statistics of the offsets -> (om, stats (2, n)): row 0 = sums of |offset|, what DCNv2Pack's
    `> 50` check needs (arch_util.py:248-253), row 1 = sums of horizontal neighbour differences (the field's roughness: which DCN
    kernel suits it, `halo_hint_from_stats`).  Without gradients they come out of the conv's own epilogue (no second pass over
    the offsets) where the layer runs on the F(4x4) kernel; the training path and small layers use the separate reduction kernel."""
    co = conv_offset.out_channels
    if _needs_grad(feat, conv_offset.weight, conv_offset.bias):
        om = offset_mask_conv(conv_offset, feat)
        return om, ops.abs_stats_per_image(om.detach()[:, :2 * co // 3])
    return conv(conv_offset, feat, act=ACT_SIGMOID, act_from=2 * co // 3, abs_sum_channels=2 * co // 3)


def offset_mask_conv(conv_offset, feat):
    """conv_offset(feat) -> ONE tensor `om` (n, 3*dg*K, h, w) as in arch_util.py:244-247 / deform_conv.py:384-387:
    chunk(3) + cat(o1, o2) is channels [0, 2/3) (offset) and sigmoid(chunk 3) is channels [2/3, 1) (mask), the sigmoid
    applied in the conv epilogue.  Consumers take zero-copy channel slices."""
    co = conv_offset.out_channels
    return conv(conv_offset, feat, act=ACT_SIGMOID, act_from=2 * co // 3)


def tapwin_takes(x, m):
    """Does the tap-window kernel (csrc/dcn_tapwin.hip) take this layer?  (W % 4 == 0, W >= 32, 8 or 16 channels per deformable
    group, at most 8 groups, the EDVR signature.)"""
    c, w, dg = x.shape[1], x.shape[3], m.deformable_groups
    return (w % 4 == 0 and w >= 32 and dg <= 8 and c % dg == 0 and c // dg in (8, 16) and tuple(m.kernel_size) == (3, 3)
            and _one(m.stride) == 1 and _one(m.padding) == 1 and _one(m.dilation) == 1 and m.groups == 1)


def _one(v):
    return v if isinstance(v, int) else (v[0] if v[0] == v[1] else -1)


def halo_hint_from_stats(absmean, rough, tapwin_ok=True):
    """Kernel class of the fused DCNv2 forward from the statistics of the PREVIOUS call of the same layer (a performance hint only):
    `absmean` = mean |offset|, `rough` = mean |horizontal neighbour difference| (None = unknown).  The kernel whose staged windows
    follow every tap's displacement (csrc/dcn_tapwin.hip, EDVR_DCN_HALO_TAPWIN) is the default: on a spatially smooth field - what
    conv_offset produces, fresh or trained: per-tap displacements of any size that vary slowly across the image - its cost does not
    depend on the offsets at all (98 - 101 TF/s on the 20 x 128 x 180x320 layer from 0 to 8 px mean |offset|), and it is the fastest
    class on most white-noise fields too (profiles/r4/dcn_sigma_sweep.log: 99 TF/s at sigma 0.5, 41 at 4, 28 at 16).  Two
    exceptions, both rough fields: white noise of about a pixel, which the zero-centred R = 7 halo covers completely (83 vs 67
    TF/s at sigma 1), and white noise of tens of pixels (every tap of every lane through the fix-up pass: the column-buffer path,
    19 vs 17.5 TF/s at sigma 64).  Layers the tap-window kernel does not take (widths not a multiple of 4, other group sizes) fall
    back to the zero-centred halo inside the C entry point."""
    if not tapwin_ok:  # the C side would fall back to R = 7 whatever the offsets: pick the halo class by magnitude (R = 3 stages half as much)
        return 3 if (absmean is None or absmean < 1.2) else (7 if absmean < 3.0 else -1)
    if absmean is None:
        return ops.DCN_HALO_TAPWIN
    if rough is None:  # rows that are not 16-byte groups: the tap-window kernel does not take those anyway -> halo classes by magnitude
        return 3 if absmean < 1.2 else (7 if absmean < 3.0 else -1)
    if rough >= 0.85 and absmean < 2.0:
        return 7
    if rough >= 40.0 and absmean >= 30.0:
        return -1
    return ops.DCN_HALO_TAPWIN


def halo_hint_from_absmean(absmean):
    return halo_hint_from_stats(absmean, None)


def scatter_hint_from_stats(absmean, rough):
    """How the DCNv2 backward accumulates dx (include/edvr_amd.h EDVR_DCN_SCATTER_*), from the statistics of the layer's latest forward.
    Sub-pixel offsets (fresh or lightly trained conv_offset): the kernels that need no scatter for taps with |offset| < 1.  Anything
    larger goes through the LDS window, whose cost barely depends on the offset field (12.7 - 15.6 ms on smooth multi-pixel fields,
    18 ms on white noise, where device atomics need 30 - 120 ms: profiles/r4/dcn_sigma_sweep.log); device atomics only in the
    narrow band where most taps of a ROUGH field are still sub-pixel (a smooth field of that size keeps the no-scatter kernels)."""
    if absmean is None:
        return ops.DCN_SCATTER_LDS
    if absmean >= 4.0:  # taps several pixels out: the window with the 6 px margin (15.0 -> 11.0 ms at sigma 6 px per tap; equal at sigma 4)
        return ops.DCN_SCATTER_LDS_WIDE
    if absmean < 0.4:  # white-noise offsets of sigma 0.5 (|mean| 0.4): 9 % of the taps already leave the sub-pixel window
        return ops.DCN_SCATTER_STRIP
    if rough is not None and rough < 0.45:  # smooth: per-tap constants - most taps are still sub-pixel up to ~0.6 (11.9 vs 14.8 ms at 0.5)
        return ops.DCN_SCATTER_STRIP if absmean < 0.6 else ops.DCN_SCATTER_LDS
    return ops.DCN_SCATTER_DEVICE if absmean < 0.75 else ops.DCN_SCATTER_LDS


def scatter_hint_from_absmean(absmean):
    return scatter_hint_from_stats(absmean, None)


def dcn_from_packed(m, x, om, act=ACT_NONE):
    """Modulated deformable conv of module `m` (weight/bias/geometry) with offsets+masks packed in `om`."""
    cfg = (m.stride, m.padding, m.dilation, m.groups, m.deformable_groups)
    hint = halo_hint_from_stats(getattr(m, 'last_offset_absmean', None), getattr(m, 'last_offset_rough', None), tapwin_takes(x, m))
    if _needs_grad(x, om, m.weight, m.bias):
        from . import autograd as ag
        return ag.DcnFromPackedFn.apply(x, om, m.weight, m.bias, (*cfg, act, hint, m))
    split = 2 * om.shape[1] // 3
    bias = m.bias.detach() if m.bias is not None else None
    # masks are sigmoid outputs and bilinear taps convex combinations of x and the zero padding: a bound of |x| bounds the columns
    # (-> the split-operand tap-window kernel) and, through the weights' norms, the output
    xb = ops.input_bound(x) if (ops.F4S_INFERENCE and hint == ops.DCN_HALO_TAPWIN and tapwin_takes(x, m)) else None
    y = ops.dcnv2_forward(x, om[:, :split], om[:, split:], m.weight.detach(), bias, *cfg, act=act, halo_hint=hint, xm_bound=xb)
    if ops.F4S_INFERENCE:
        ops.linear_bound(y, m.weight, m.bias, (x,))
    return y


def upsample2x(x, scale=1.0):
    if _needs_grad(x):
        from . import autograd as ag
        return ag.Upsample2x.apply(x, scale)
    return ops.upsample2x(x, scale)


def pool_maxavg(x):
    if _needs_grad(x):
        from . import autograd as ag
        return ag.PoolMaxAvg.apply(x)
    return ops.pool_maxavg(x)


def tsa_temporal(emb, emb_ref, aligned):
    if _needs_grad(emb, emb_ref, aligned):
        from . import autograd as ag
        return ag.TsaTemporal.apply(emb, emb_ref, aligned)
    return ops.tsa_temporal(emb, emb_ref, aligned)


def tsa_combine(feat, attn, attn_add):
    if _needs_grad(feat, attn, attn_add):
        from . import autograd as ag
        return ag.TsaCombine.apply(feat, attn, attn_add)
    return ops.tsa_combine(feat, attn, attn_add)


def add(a, b):
    if _needs_grad(a, b):
        from . import autograd as ag
        return ag.Add.apply(a, b)
    return ops.add(a, b)


def upsample4x_add(y, base):
    """y + bilinear_x4(base)."""
    if _needs_grad(y, base):
        from . import autograd as ag
        return ag.Upsample4xAdd.apply(y, base)
    return ops.upsample4x_add_(y, base)
