"""edvr_amd - the EDVR hot path (DCNv2 / PCD alignment / TSA fusion / conv trunks) as
hand-written HIP for AMD MI355X (gfx950), behind the module API of xinntao/EDVR.

    from edvr_amd import EDVR, PCDAlignment, TSAFusion, PredeblurModule
    from edvr_amd import DCNv2Pack, ModulatedDeformConv, ModulatedDeformConvPack, modulated_deform_conv

The kernels live in edvr_amd/lib/libedvr_amd.so (C ABI: include/edvr_amd.h; build with
`python -m edvr_amd.build`).  There is no CPU or stock-PyTorch fallback: CPU tensors raise
NotImplementedError exactly like the reference op, and a missing library raises ExtensionMissing.
"""
from ._lib import ExtensionMissing  # noqa: F401
from .arch_util import DCNv2Pack, ResidualBlockNoBN, default_init_weights, make_layer  # noqa: F401
from .dcn import (DeformConv, DeformConvFunction, DeformConvPack, ModulatedDeformConv,  # noqa: F401
                  ModulatedDeformConvFunction, ModulatedDeformConvPack, deform_conv, modulated_deform_conv)
from .edvr_arch import EDVR, PCDAlignment, PredeblurModule, TSAFusion  # noqa: F401

__version__ = '0.1.0'
