"""TEST INFRASTRUCTURE ONLY - import the reference's own Python EDVR, unchanged, on CPU.

Works only where /root/reference exists (the authoring container), never on the GPU box.
Stubs the modules the reference imports but this image lacks (cv2, lmdb, torchvision and
the un-buildable CUDA extension basicsr.models.ops.dcn.deform_conv_ext) and replaces exactly
ONE symbol: basicsr.models.archs.arch_util.modulated_deform_conv, which DCNv2Pack.forward
looks up at call time (arch_util.py:255), by the oracle DCNv2 op - the reference's native op
is CUDA-only (deform_conv.py:133-134).
"""
import os
import sys
import types

REF_ROOT = '/root/reference'


def available():
    return os.path.exists(os.path.join(REF_ROOT, 'basicsr', 'models', 'archs', 'edvr_arch.py'))


def load(dcn=None):
    """Return (edvr_arch module, arch_util module) of the reference."""
    if not available():
        raise FileNotFoundError(REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for n in ['cv2', 'lmdb', 'torchvision', 'torchvision.utils', 'torchvision.models', 'torchvision.models.vgg',
              'basicsr.models.ops.dcn.deform_conv_ext']:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    tv = sys.modules['torchvision']
    tv.__version__ = 'stub'
    tv.utils = sys.modules['torchvision.utils']
    tv.utils.make_grid = lambda *a, **k: None
    tv.models = sys.modules['torchvision.models']
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import basicsr.models.archs.arch_util as au
        import basicsr.models.archs.edvr_arch as ea
    if dcn is None:
        from . import dcn_oracle
        dcn = dcn_oracle.dcnv2_c
    au.modulated_deform_conv = lambda x, o, m, w, b, s, p, d, g, dg: dcn(x, o.contiguous(), m.contiguous(), w, b, s, p, d, g, dg)
    return ea, au
