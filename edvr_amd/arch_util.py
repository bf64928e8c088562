"""Building blocks of the EDVR hot path on HIP kernels.

Mirrors the EDVR-relevant part of basicsr/models/archs/arch_util.py (xinntao/EDVR):
  default_init_weights (:20-48), make_layer (:51-64), ResidualBlockNoBN (:67-95), DCNv2Pack (:232-257).
Same names, constructor arguments, parameter names and initialisation (including the order in
which the RNG is consumed, so `torch.manual_seed(s)` builds bit-identical networks).
"""
import logging

import torch
from torch import nn
from torch.nn import init
from torch.nn.modules.batchnorm import _BatchNorm

from . import functional as F_
from .dcn import ModulatedDeformConvPack

OFFSET_ABSMEAN_LIMIT = 50  # arch_util.py:249


def get_root_logger():
    return logging.getLogger('basicsr')


@torch.no_grad()
def default_init_weights(module_list, scale=1, bias_fill=0, **kwargs):
    for top in (module_list if isinstance(module_list, list) else [module_list]):
        for m in top.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, **kwargs)
                m.weight.mul_(scale)  # in place on the Parameter (not .data): bumps the version the packed-weight cache checks
                if m.bias is not None:
                    m.bias.fill_(bias_fill)
            elif isinstance(m, _BatchNorm):
                init.constant_(m.weight, 1)
                if m.bias is not None:
                    m.bias.fill_(bias_fill)


def make_layer(basic_block, num_basic_block, **kwarg):
    return nn.Sequential(*(basic_block(**kwarg) for _ in range(num_basic_block)))


class ResidualBlockNoBN(nn.Module):
    """x + res_scale * conv2(relu(conv1(x))): two MFMA launches, ReLU, the scale and the identity add fused in."""

    def __init__(self, num_feat=64, res_scale=1, pytorch_init=False):
        super().__init__()
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=True)
        self.relu = nn.ReLU(inplace=True)
        if not pytorch_init:
            default_init_weights([self.conv1, self.conv2], 0.1)

    def forward(self, x):
        c = self.conv1.out_channels
        if (torch.is_grad_enabled() and (x.requires_grad or self.conv1.weight.requires_grad) and x.dim() == 4
                and self.conv1.in_channels == c and F_.ops.conv_gate_supported(x.shape[0], c, x.shape[2], x.shape[3], c)):
            from . import autograd as ag  # where the Winograd kernel applies: the fused ReLU-backward gate lives in its epilogue
            return ag.resblock(self, x)
        return F_.conv(self.conv2, F_.conv(self.conv1, x, act=F_.ACT_RELU), res1=x, y_scale=float(self.res_scale))


def warn_offset_absmean(value):
    if value > OFFSET_ABSMEAN_LIMIT:
        get_root_logger().warning(f'Offset abs mean is {value}, larger than 50.')


class DCNv2Pack(ModulatedDeformConvPack):
    """DCNv2 whose offsets/masks are predicted from a SECOND feature map (arch_util.py:232-257).

    `stats_sink`: when a list is attached (EDVR.forward does), the per-image sums of |offset| are
    appended as device tensors and the `> 50` warning is evaluated once per forward; standalone use
    keeps the reference behaviour (check right away, one host sync per call).
    """

    stats_sink = None
    last_offset_absmean = None  # mean |offset| of the previous call and
    last_offset_rough = None    # its mean |horizontal neighbour difference|: pick the fused kernel's class (perf hints only)

    def forward(self, x, feat, act=F_.ACT_NONE):
        om, stats = F_.offset_mask_conv_stats(self.conv_offset, feat)
        offset = om.detach()[:, :2 * om.shape[1] // 3]
        if self.stats_sink is not None:
            self.stats_sink.append((stats, offset[0].numel(), self))
        else:
            self.last_offset_absmean, self.last_offset_rough = F_.ops.offset_stats(stats.cpu(), offset.numel())
            warn_offset_absmean(self.last_offset_absmean)
        return F_.dcn_from_packed(self, x, om, act)
