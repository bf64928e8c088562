"""Autograd Functions for the fused HIP launches (training path of the EDVR hot path).

Every Function's backward runs hand-written HIP kernels too (dgrad = the forward MFMA kernel on
transposed/flipped packed weights, wgrad = csrc/wgrad.hip, glue gradients = csrc/backward.hip).
All are once_differentiable, like the reference's DCN Function (deform_conv.py:149).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from .ops import ACT_NONE, ACT_RELU, OUT_PIXEL_SHUFFLE2


class ConvFn(Function):
    """y = act(conv(cat(x, x2)) + bias) + res1 + res2 (see ops.conv2d)."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, res1, res2, cfg):
        x2_map, act, act_from, out_mode, ks, stride, y_scale = cfg
        wpk = ops.pack_conv_weight(weight)
        y = ops.conv2d(x, wpk, bias.detach() if bias is not None else None, weight.shape[0], ks, x2=x2, x2_map=x2_map, stride=stride,
                       act=act, act_from=act_from, res1=res1, res2=res2, out_mode=out_mode, y_scale=y_scale,
                       **(ops.f4_kwargs(weight, ks) if stride == 1 else {}))
        if ops.F4S_TRAINING and ops.get_bound(y) is None:  # (kernels without the y_amax epilogue: bound from the weights' norms)
            ops.linear_bound(y, weight, bias, (x, x2), (res1, res2), scale=y_scale, floor=1.0 if act == ops.ACT_SIGMOID else 0.0)
        if y_scale != 1.0 and act != ACT_NONE:
            raise NotImplementedError('y_scale together with a fused activation has no backward (not used by EDVR)')
        keep_y = act != ACT_NONE
        ctx.save_for_backward(x, x2, weight, y if keep_y else None, res1 if keep_y else None, res2 if keep_y else None)
        ctx.cfg = cfg
        ctx.has = (bias is not None, res1 is not None, res2 is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, x2, weight, y, res1, res2 = ctx.saved_tensors
        x2_map, act, act_from, out_mode, ks, stride, y_scale = ctx.cfg
        has_bias, has_r1, has_r2 = ctx.has
        need = ctx.needs_input_grad
        co = weight.shape[0]
        dres = dy if (has_r1 and need[4]) or (has_r2 and need[5]) else None
        # gradient w.r.t. the pre-activation conv output z (n, co, ho, wo)
        if out_mode == OUT_PIXEL_SHUFFLE2:
            dz = ops.pixel_unshuffle2_act_backward(dy, y, act)  # activation backward and un-shuffle in one pass
        else:
            dz = ops.act_backward(dy, y, act, act_from, res1, res2) if act != ACT_NONE else dy
        want_db = has_bias and need[3]
        if need[2]:  # the weight-gradient launch produces the bias gradient as well
            dw = ops.conv2d_wgrad(x, x2, x2_map, dz, co, ks, stride, want_db=want_db)
            dw, db = dw if want_db else (dw, None)
        else:
            dw, db = None, (ops.channel_sum(dz) if want_db else None)
        if y_scale != 1.0:  # y = y_scale * (W x + b) + res: the parameter gradients are linear in the scale (tiny tensors)
            dw = dw.mul_(y_scale) if dw is not None else None
            db = db.mul_(y_scale) if db is not None else None
        dx = dx2 = None
        if need[0] or (x2 is not None and need[1]):
            c1 = x.shape[1]
            z = ops.zero_stuff2(dz, x.shape[2], x.shape[3]) if stride == 2 else dz
            wt = ops.pack_conv_weight(weight, transpose_flip=True)
            dcat = ops.conv2d(z, wt, None, weight.shape[1], ks, y_scale=y_scale,  # data gradient = stride-1 conv with flipped W^T
                              **ops.f4_kwargs(weight, ks, transpose_flip=True))
            if ops.F4S_TRAINING and ops.get_bound(dcat) is None:
                ops.linear_bound(dcat, weight, None, (z,), transpose=True, scale=y_scale)
            if need[0]:
                dx = ops.carry_bound(dcat[:, :c1], dcat) if x2 is not None else dcat
            if x2 is not None and need[1]:
                d2 = ops.carry_bound(dcat[:, c1:], dcat)
                if x2_map is not None:
                    div, mul, add = x2_map
                    assert div == mul, 'image map must address one frame per clip'
                    dx2 = torch.zeros_like(x2)
                    ops.frame_reduce_add_(d2, dx2, div, add)  # every frame of a clip read the same reference frame
                else:
                    dx2 = d2
        return dx, dx2, dw, db, (dres if has_r1 else None), (dres if has_r2 else None), None


class ResBlockFn(Function):
    """y = x + conv2(relu(conv1(x))) (ResidualBlockNoBN, arch_util.py:66-95) as ONE autograd node.  Forward = the same two fused
    launches; backward fuses what separate nodes cannot: the ReLU backward is the `gate` of the data-gradient conv of conv2
    (no act_backward launch) and the identity branch's gradient is the residual of the data-gradient conv of conv1 (no
    gradient-accumulation add by the autograd engine)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res_scale=1.0):
        c = w1.shape[0]
        h = ops.conv2d(x, ops.pack_conv_weight(w1), b1.detach() if b1 is not None else None, c, 3, act=ACT_RELU, **ops.f4_kwargs(w1, 3))
        y = ops.conv2d(h, ops.pack_conv_weight(w2), b2.detach() if b2 is not None else None, c, 3, res1=x, y_scale=res_scale,
                       **ops.f4_kwargs(w2, 3))
        ctx.save_for_backward(x, h, w1, w2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.res_scale = float(res_scale)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, h, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        c = w1.shape[0]
        s = ctx.res_scale
        dy = dy.contiguous()
        dw2 = db2 = dw1 = db1 = dx = None
        if need[3] or (need[4] and ctx.has_bias[1]):
            dw2, db2 = ops.conv2d_wgrad(h, None, None, dy, c, 3, 1, want_db=True)
            if s != 1.0:  # the branch's output gradient is s * dy; dW2 / db2 are linear in it (two small tensors)
                dw2.mul_(s)
                db2.mul_(s)
        # d(pre-activation of conv1) = s * (W2^T * dy) gated by relu'(z1) = [h > 0]
        dz1 = ops.conv2d(dy, ops.pack_conv_weight(w2, transpose_flip=True), None, c, 3, gate=h, gate_slope=0.0, y_scale=s,
                         **ops.f4_kwargs(w2, 3, transpose_flip=True))
        if need[1] or (need[2] and ctx.has_bias[0]):
            dw1, db1 = ops.conv2d_wgrad(x, None, None, dz1, c, 3, 1, want_db=True)
        if need[0]:
            dx = ops.conv2d(dz1, ops.pack_conv_weight(w1, transpose_flip=True), None, c, 3, res1=dy,  # + identity branch
                            **ops.f4_kwargs(w1, 3, transpose_flip=True))
        return (dx, dw1 if need[1] else None, db1 if (need[2] and ctx.has_bias[0]) else None, dw2 if need[3] else None,
                db2 if (need[4] and ctx.has_bias[1]) else None, None)


def resblock(m, x):
    return ResBlockFn.apply(x, m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias, float(m.res_scale))


def conv(m, x, x2, x2_map, act, act_from, res1, res2, out_mode, ks, stride, y_scale=1.0):
    return ConvFn.apply(x, x2, m.weight, m.bias, res1, res2, (x2_map, act, act_from, out_mode, ks, stride, float(y_scale)))


class DcnFromPackedFn(Function):
    """DCNv2 fed by ONE conv_offset output `om` (n, 3*dg*K, h, w): offset = first 2/3 channels, mask = last third
    (already sigmoid-ed by the conv epilogue).  Backward writes d(offset) and d(mask) straight into slices of d(om)."""

    @staticmethod
    def forward(ctx, x, om, weight, bias, cfg):
        stride, padding, dilation, groups, dg, act, hint, _ = cfg
        split = 2 * om.shape[1] // 3
        xb = ops.input_bound(x) if (ops.F4S_TRAINING and hint == ops.DCN_HALO_TAPWIN) else None  # (sigmoid masks: a bound of |x| does)
        out = ops.dcnv2_forward(x, om[:, :split], om[:, split:], weight, bias, stride, padding, dilation, groups, dg, act=act,
                                halo_hint=hint, xm_bound=xb)
        if ops.F4S_TRAINING:  # (masks are sigmoid outputs, bilinear taps convex combinations of x and the zero padding)
            ops.linear_bound(out, weight, bias, (x,))
        ctx.save_for_backward(x, om, weight, out if act != ACT_NONE else None)
        ctx.cfg = cfg
        ctx.with_bias = bias is not None
        ctx.xm_bound = (xb if xb is not None else ops.get_bound(x)) if ops.F4S_TRAINING else None  # for the split dW product of the backward
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, om, weight, out = ctx.saved_tensors
        stride, padding, dilation, groups, dg, act, _, module = ctx.cfg
        from .functional import scatter_hint_from_stats
        scatter = scatter_hint_from_stats(getattr(module, 'last_offset_absmean', None), getattr(module, 'last_offset_rough', None))  # latest statistics that have ARRIVED (this forward's only under EDVR_DCN_HINT_WAIT=1)
        if act != ACT_NONE:
            dy = ops.act_backward(dy, out, act)
        split = 2 * om.shape[1] // 3
        dom = torch.empty_like(om)
        xb = ctx.xm_bound
        dx, _, _, dw, db = ops.dcnv2_backward(x, om[:, :split], om[:, split:], weight, dy, ctx.with_bias, stride, padding, dilation,
                                              groups, dg, doffset=dom[:, :split], dmask=dom[:, split:], scatter_hint=scatter,
                                              xm_bound=xb, dy_bound=ops.input_bound(dy) if xb is not None else None)
        return dx, dom, dw, db, None


class Upsample2x(Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return ops.upsample2x(x, scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.upsample2x_backward(dy, ctx.scale), None


class PoolMaxAvg(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.pool_maxavg(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.pool_maxavg_backward(x, dy)


class TsaTemporal(Function):
    @staticmethod
    def forward(ctx, emb, emb_ref, aligned):
        ctx.save_for_backward(emb, emb_ref, aligned)
        return ops.tsa_temporal(emb, emb_ref, aligned)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        emb, emb_ref, aligned = ctx.saved_tensors
        return ops.tsa_temporal_backward(emb, emb_ref, aligned, dout)


class TsaCombine(Function):
    @staticmethod
    def forward(ctx, feat, attn, attn_add):
        ctx.save_for_backward(feat, attn)
        return ops.tsa_combine(feat, attn, attn_add)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        feat, attn = ctx.saved_tensors
        dfeat, dattn = ops.tsa_combine_backward(feat, attn, dy)
        return dfeat, dattn, dy


class Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return dy, dy


class Upsample4xAdd(Function):
    """y + bilinear_x4(base); base is the network input (no gradient is propagated to it)."""

    @staticmethod
    def forward(ctx, y, base):
        if base.requires_grad:
            raise NotImplementedError('gradient w.r.t. the input frames is not provided by the x4 base-add kernel')
        ctx.mark_dirty(y)
        return ops.upsample4x_add_(y, base)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return dy, None


class CharbonnierSum(Function):
    """CharbonnierLoss(reduction='sum', eps=1e-12) (losses/losses.py:23-25); forward + gradient in one pass."""

    @staticmethod
    def forward(ctx, pred, target, eps):
        loss, dpred = ops.charbonnier(pred, target, eps, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None


def charbonnier_loss(pred, target, eps=1e-12):
    return CharbonnierSum.apply(pred, target, eps)
