"""Ablation / experiment builds of the split-operand F(4x4) kernel WITHOUT switches in the product source: a named list of textual
patches is applied to a copy of edvr_amd/csrc/winograd_f4s.hip, the copy is compiled, and linked with the product's other objects into
edvr_amd/lib/variants/libedvr_amd_<name>.so (edvr_version() says "variant:<name>"; tests/conftest.py refuses such a library).
Most ablations compute WRONG results on purpose: they exist to be timed (scripts/bench_f4s.py, EDVR_AMD_LIB=...).
    python scripts/exp/f4s_variant.py NAME [NAME ...]        (names may be joined with '+': nouload+nodma)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from edvr_amd import build  # noqa: E402

PATCHES = {
    # multiplying waves: the weights are fetched once per item instead of once per chunk
    'nouload': [('          load_a(c, soff_nxt);  // the same position of the next chunk', '          // (ablation: no reload)  // the same position of the next chunk')],
    # staging waves: no input fetch at all
    'nodma': [('    auto dma_issue = [&]() {\n', '    auto dma_issue = [&]() {\n      return;\n')],
    # staging waves only keep the barrier count in the chunk loop
    'noprod': [('        read_patch();\n        advance();\n        load_begin(l_k * CK);\n        dma_issue();\n#pragma unroll\n        for (int cp = 0; cp < 3; ++cp) transform_cols(cp);\n#pragma unroll\n        for (int r = 0; r < 6; ++r) commit_row(Vd, r);\n        F4S_BARRIER();\n        par ^= 1;',
                '        (void)Vd;\n        F4S_BARRIER();\n        par ^= 1;')],
    # no output transform / stores (both sides skip the eight phases)
    'noepi': [('        for (int p = 0; p < 8; ++p) {\n          F4S_BARRIER();  // T of this phase', '        for (int p = 0; p < 0; ++p) {\n          F4S_BARRIER();  // T of this phase'),
              ('      for (int p = 0; p < 8; ++p) {\n        float *Xb = Xs + (p & 1) * (XSZ / 2);', '      for (int c = 0; c < 6; ++c) asm volatile("" ::"v"(acc[c]));\n#pragma unroll\n      for (int p = 0; p < 0; ++p) {\n        float *Xb = Xs + (p & 1) * (XSZ / 2);')],
    # one MFMA per position instead of two (no rotation)
    'nomfma2': [('          asm volatile("v_alignbit_b32 %0, %0, %0, 16\\n\\tv_alignbit_b32 %1, %1, %1, 16\\n\\tv_alignbit_b32 %2, %2, %2, 16\\n\\tv_alignbit_b32 %3, %3, %3, 16"\n                       : "+v"(A[c][0]), "+v"(A[c][1]), "+v"(A[c][2]), "+v"(A[c][3]));\n          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c]), B, acc[c], 0, 0, 0);\n', '')],
    # no MFMA at all (operands fetched and kept alive)
    'nomfma': [('          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c]), B, acc[c], 0, 0, 0);\n          // (hi, lo) -> (lo, hi) IN PLACE',
                '          asm volatile("" ::"v"(A[c]), "v"(B));\n          // (hi, lo) -> (lo, hi) IN PLACE'),
               ('                       : "+v"(A[c][0]), "+v"(A[c][1]), "+v"(A[c][2]), "+v"(A[c][3]));\n          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c]), B, acc[c], 0, 0, 0);',
                '                       : "+v"(A[c][0]), "+v"(A[c][1]), "+v"(A[c][2]), "+v"(A[c][3]));')],
    # multiplying waves do not read V from LDS
    'nobread': [('          asm volatile("ds_read_b32 %0, %4 offset:%5\\n\\tds_read_b32 %1, %4 offset:%6\\n\\tds_read_b32 %2, %4 offset:%7\\n\\tds_read_b32 %3, %4 offset:%8\\n\\t"\n                       "s_waitcnt lgkmcnt(0)"',
                 '          asm volatile("v_mov_b32 %0, %4\\n\\tv_mov_b32 %1, %4\\n\\tv_mov_b32 %2, %4\\n\\tv_mov_b32 %3, %4\\n\\t; %5 %6 %7 %8"')],
    # staging waves: no split / no V writes (transform kept alive)
    'novw': [('      unsigned pk[6];\n      split6_f16x2(t, s_v, pk);\n#pragma unroll\n      for (int c = 0; c < 6; ++c) dst[c * 32] = pk[c];', '      asm volatile("" ::"v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(dst));')],
    # cache-policy experiments: nt on the input DMA / on the output stores / on the weight loads
    'dmant': [('(lvoid *)(Rw + 1 + i * 256), 16, dma_off[i], 0, 0, 0);', '(lvoid *)(Rw + 1 + i * 256), 16, dma_off[i], 0, 0, 2);')],
    'stnt': [('                if (i < rows_in) *reinterpret_cast<f32x4 *>(q + i * d.w) = Y[i];', '                if (i < rows_in) __builtin_nontemporal_store(Y[i], reinterpret_cast<f32x4 *>(q + i * d.w));')],
    'unt': [('__builtin_amdgcn_raw_buffer_load_b128(u_rsrc, voff, soff + c * 1024, 0)', '__builtin_amdgcn_raw_buffer_load_b128(u_rsrc, voff, soff + c * 1024, 2)')],
    # every XCD works on ONE 64-channel block of the weights (its L2 holds half / a quarter of U); the input is then read by two / four L2s
    'xcdco': [('  const int span = (a.items + n_xcd - 1) / n_xcd;\n  const int item_end = min(a.items, (xcd + 1) * span);\n  const int item_first = xcd * span + xcd_rank;',
               '  const bool cbfix = n_xcd == 8 && (co_blocks == 2 || co_blocks == 4 || co_blocks == 8);\n  const int cb_fixed = cbfix ? xcd % co_blocks : -1;\n  const int n_grp = cbfix ? 8 / co_blocks : n_xcd, grp = cbfix ? xcd / co_blocks : xcd;\n  const int v_items = cbfix ? a.items / co_blocks : a.items;\n  const int span = (v_items + n_grp - 1) / n_grp;\n  const int item_end = min(v_items, (grp + 1) * span);\n  const int item_first = grp * span + xcd_rank;'),
              ('    co_blk = __builtin_amdgcn_readfirstlane((item % co_blocks) * 64);\n    const int tile_blk = __builtin_amdgcn_readfirstlane((item / co_blocks) % (a.tiles_x * a.tiles_y));\n    img = __builtin_amdgcn_readfirstlane(item / (co_blocks * a.tiles_x * a.tiles_y));',
               '    const int cbd = cb_fixed >= 0 ? 1 : co_blocks;\n    co_blk = __builtin_amdgcn_readfirstlane((cb_fixed >= 0 ? cb_fixed : item % co_blocks) * 64);\n    const int tile_blk = __builtin_amdgcn_readfirstlane((item / cbd) % (a.tiles_x * a.tiles_y));\n    img = __builtin_amdgcn_readfirstlane(item / (cbd * a.tiles_x * a.tiles_y));')],
    'nosplit': [('      split6_f16x2(t, s_v, pk);', '      for (int c = 0; c < 6; ++c) pk[c] = __builtin_bit_cast(unsigned, t[c]);')],
}


def make(name):
    src = open(os.path.join(build.CSRC, 'winograd_f4s.hip')).read()
    for part in name.split('+'):
        if part == 'base':
            continue
        for old, new in PATCHES[part]:
            assert src.count(old) == 1, (part, src.count(old), old[:60])
            src = src.replace(old, new)
    vdir = os.path.join(build.OBJDIR, 'f4sv_' + name.replace('+', '_'))
    os.makedirs(vdir, exist_ok=True)
    path = os.path.join(vdir, 'winograd_f4s.hip')
    open(path, 'w').write(src)
    hipcc = build._hipcc()
    tag = name.replace('+', '_')
    flags = build.FLAGS + ['-I', build.CSRC, f'-DEDVR_VARIANT={tag}']
    subprocess.check_call([hipcc] + flags + ['-c', path, '-o', os.path.join(vdir, 'winograd_f4s.o')])
    subprocess.check_call([hipcc] + flags + ['-c', os.path.join(build.CSRC, 'api.hip'), '-o', os.path.join(vdir, 'api.o')])
    build.build()
    objs = [os.path.join(vdir, o) if o in ('winograd_f4s.o', 'api.o') else os.path.join(build.OBJDIR, o) for o in (s.replace('.hip', '.o') for s in build.SOURCES)]
    out = os.path.join(build.LIBDIR, 'variants', f'libedvr_amd_{tag}.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


if __name__ == '__main__':
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        for o in ex.map(make, sys.argv[1:]):
            print('built', o)
