"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/edvr_amd.h declares."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'edvr_amd.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(edvr_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ('edvr_dcnv2_fwd_f32', 'edvr_dcnv2_bwd_f32', 'edvr_conv2d_f32', 'edvr_tsa_temporal_f32'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from edvr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from edvr_amd import build
        build.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(handle, s)]
    assert not missing, missing
    assert sorted(_lib.PROTOTYPES) == declared_symbols(), 'python binding and header disagree'
    lib = _lib.lib()
    assert b'gfx950' in lib.edvr_version() and b'variant:' not in lib.edvr_version()  # (scripts/build_variant.sh builds say so)
    # pure host helpers, no GPU needed: direct layout [16][9][64] + Winograd-transformed weights [16][16][64]; 1x1: [32][1][32]
    assert lib.edvr_conv2d_packed_weight_elems(64, 3, 3) == 16 * 9 * 64 + 16 * 16 * 64
    assert lib.edvr_conv2d_packed_weight_elems(20, 33, 1) == 64 * 32


def test_conv_desc_layout_matches_the_c_struct(tmp_path):
    """sizeof / offsetof of edvr_conv2d_desc as gcc sees them == the ctypes mirror."""
    from edvr_amd._lib import ConvDesc
    fields = [f[0] for f in ConvDesc._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "edvr_amd.h"\nint main(){printf("%zu", sizeof(edvr_conv2d_desc));' + \
        ''.join(f'printf(" %zu", offsetof(edvr_conv2d_desc, {f}));' for f in fields) + 'return 0;}'
    src = tmp_path / 't.c'
    src.write_text(prog)
    exe = tmp_path / 't'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    vals = list(map(int, subprocess.check_output([str(exe)]).split()))
    assert vals[0] == ctypes.sizeof(ConvDesc)
    assert vals[1:] == [getattr(ConvDesc, f).offset for f in fields]


def test_missing_library_fails_loudly(monkeypatch):
    from edvr_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libedvr_amd.so')
    with pytest.raises(_lib.ExtensionMissing):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """The product path must not route through test infrastructure."""
    pkg = os.path.join(ROOT, 'edvr_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
                assert 'liboracle' not in txt and 'libdcn_ref' not in txt, f
