"""Modulated deformable convolution (DCNv2): the reference's op/module API on HIP kernels.

Mirrors basicsr/models/ops/dcn/deform_conv.py of xinntao/EDVR:
  ModulatedDeformConvFunction (:111-181)  -> ModulatedDeformConvFunction
  modulated_deform_conv       (:185)      -> modulated_deform_conv
  ModulatedDeformConv         (:295-342)  -> ModulatedDeformConv
  ModulatedDeformConvPack     (:345-390)  -> ModulatedDeformConvPack
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
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, out if act != ops.ACT_NONE else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, out = ctx.saved_tensors
        if ctx.act != ops.ACT_NONE:
            grad_output = ops.act_backward(grad_output, out, ctx.act)
        dx, doff, dmsk, dw, db = ops.dcnv2_backward(input, offset, mask, weight, grad_output, ctx.with_bias, *ctx.cfg)
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

    def init_weights(self):
        fan = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan)
        self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.zero_()

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

    def init_weights(self):
        super().init_weights()
        if hasattr(self, 'conv_offset'):
            self.conv_offset.weight.data.zero_()
            self.conv_offset.bias.data.zero_()

    def forward(self, x):
        from . import functional as F_
        return F_.dcn_from_packed(self, x, F_.offset_mask_conv(self.conv_offset, x))
