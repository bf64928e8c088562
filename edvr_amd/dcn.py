"""Modulated deformable convolution (DCNv2): the reference's op/module API on HIP kernels.

Mirrors basicsr/models/ops/dcn/deform_conv.py of xinntao/EDVR:
  ModulatedDeformConvFunction (:111-181)  -> ModulatedDeformConvFunction
  modulated_deform_conv       (:185)      -> modulated_deform_conv
  ModulatedDeformConv         (:295-342)  -> ModulatedDeformConv
  ModulatedDeformConvPack     (:345-390)  -> ModulatedDeformConvPack
  DeformConvFunction / deform_conv / DeformConv / DeformConvPack (:12-108,183,188-292: DCNv1)
Same constructor arguments, attributes, parameter names/shapes/init, `_version = 2`,
and the same refusal of CPU tensors (NotImplementedError, :133-134,151-152).  The native
side is edvr_dcnv2_{fwd,bwd}_f32 of libedvr_amd.so instead of the deform_conv_ext module.
"""
import math

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from . import ops


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, act=ops.ACT_NONE):
        # `act` (extension, default = the reference op): activation fused into the GEMM epilogue.
        if not input.is_cuda:
            raise NotImplementedError
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups)
        ctx.with_bias = bias is not None
        ctx.act = act
        out = ops.dcnv2_forward(input, offset, mask, weight, bias, *ctx.cfg, act=act)
        ctx.offset_stat = None
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, out if act != ops.ACT_NONE else None)
            if offset.dtype == torch.float32:  # mean |offset| of this call, on its way to the host while the graph above runs:
                ctx.offset_stat = ops.note_abs_mean(offset)  # the backward's dX strategy (performance only, never waited for)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, out = ctx.saved_tensors
        if ctx.act != ops.ACT_NONE:
            grad_output = ops.act_backward(grad_output, out, ctx.act)
        from .functional import scatter_hint_from_stats
        st = ops.offset_stats_if_ready(ctx.offset_stat)  # (EDVR_DCN_HINT_WAIT=1: waits, so the kernel choice is deterministic)
        hint = scatter_hint_from_stats(*st) if st is not None else ops.DCN_SCATTER_AUTO
        dx, doff, dmsk, dw, db = ops.dcnv2_backward(input, offset, mask, weight, grad_output, ctx.with_bias, *ctx.cfg, scatter_hint=hint)
        return dx, doff, dmsk, dw, db, None, None, None, None, None, None


modulated_deform_conv = ModulatedDeformConvFunction.apply


class ModulatedDeformConv(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.with_bias = bias
        self.transposed, self.output_padding = False, _single(0)  # nn.Conv2d look-alike attributes
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.init_weights()

    @torch.no_grad()
    def init_weights(self):
        # in-place ops on the Parameters themselves (not on .data): they bump the version counter the packed-weight cache
        # of ops.pack_conv_weight is keyed on, so a re-initialisation after a first forward is seen by the kernels
        fan = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan)
        self.weight.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """DCNv2 that predicts its own offsets/masks with a zero-initialised `conv_offset`."""

    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        taps = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 3 * taps, kernel_size=self.kernel_size,
                                     stride=_pair(self.stride), padding=_pair(self.padding), dilation=_pair(self.dilation),
                                     bias=True)
        self.init_weights()

    @torch.no_grad()
    def init_weights(self):
        super().init_weights()
        if hasattr(self, 'conv_offset'):
            self.conv_offset.weight.zero_()
            self.conv_offset.bias.zero_()

    def forward(self, x):
        from . import functional as F_
        return F_.dcn_from_packed(self, x, F_.offset_mask_conv(self.conv_offset, x))


# ------------------------------------------------------------------------------------------------ DCNv1
class DeformConvFunction(Function):
    """deform_conv.py:12-108.  `im2col_step` only sizes the reference's column buffer; it is accepted, checked the same way
    (must divide the batch) and otherwise unused."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {input.dim()}D tensor instead.')
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)  # (h, w) pairs, :33-35
        if not input.is_cuda:
            raise NotImplementedError
        cur = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur) == 0, 'im2col step must divide batchsize'
        DeformConvFunction._output_size(input, weight, padding, dilation, stride)  # raises like the reference if too small
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups)  # rectangular pairs travel as EDVR_HW(h, w) (ops._enc_hw)
        ctx.save_for_backward(input, offset, weight)
        return ops.dcnv1_forward(input.contiguous(), offset, weight.contiguous(), *ctx.cfg)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise NotImplementedError
        dx, doff, dw = ops.dcnv1_backward(input.contiguous(), offset, weight.contiguous(), grad_output, *ctx.cfg)
        return dx, doff, dw, None, None, None, None, None, None

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        """(N, Co, Ho, Wo) of the op - same name, arguments and error as the reference's helper (deform_conv.py:94-108)."""
        spatial = tuple((int(i) + 2 * p - (d * (int(k) - 1) + 1)) // s + 1
                        for i, k, p, d, s in zip(input.shape[2:], weight.shape[2:], padding, dilation, stride))
        shape = (int(input.shape[0]), int(weight.shape[0])) + spatial
        if min(shape) <= 0:
            raise ValueError('convolution input is too small (output would be ' + 'x'.join(str(v) for v in shape) + ')')
        return shape


deform_conv = DeformConvFunction.apply


class DeformConv(nn.Module):
    """deform_conv.py:188-250."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                 bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, f'in_channels {in_channels} is not divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} is not divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed = False  # enable compatibility with nn.Conv2d
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        # An input smaller than the kernel is grown with zeros at the bottom / right up to the kernel size, the offsets alike, and the
        # output cropped back by the same amount - the behaviour of the reference's workaround (deform_conv.py:234-250)
        grow_h, grow_w = max(self.kernel_size[0] - x.shape[2], 0), max(self.kernel_size[1] - x.shape[3], 0)
        if grow_h or grow_w:
            x, offset = (torch.nn.functional.pad(t, (0, grow_w, 0, grow_h)).contiguous() for t in (x, offset))
        y = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)
        if grow_h or grow_w:
            y = y[:, :, :y.shape[2] - grow_h, :y.shape[3] - grow_w].contiguous()
        return y


class DeformConvPack(DeformConv):
    """deform_conv.py:253-292: conv_offset (zero-initialised, nn.Conv2d with the same geometry) predicts the offsets."""

    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
                                     dilation=_pair(self.dilation), bias=True)
        self.init_offset()

    @torch.no_grad()
    def init_offset(self):
        self.conv_offset.weight.zero_()
        self.conv_offset.bias.zero_()

    def forward(self, x):
        from . import functional as F_
        offset = F_.conv(self.conv_offset, x)  # the fused conv kernel: 3x3 (stride 1/2, pad 1) or 1x1 geometries
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)
