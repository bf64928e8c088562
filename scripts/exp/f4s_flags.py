"""Experiment builds of ONE translation unit with extra -D flags: the unit (default winograd_f4s.hip) and api.hip are compiled with the
flags + -DEDVR_VARIANT=<name>, linked with the product's other objects into edvr_amd/lib/variants/libedvr_amd_<name>.so.
    python scripts/exp/f4s_flags.py NAME[:unit.hip] "-DX=1 -DY=2" [NAME "flags" ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from edvr_amd import build  # noqa: E402


def make(name, flags):
    name, _, unit = name.partition(':')
    unit = unit or 'winograd_f4s.hip'
    vdir = os.path.join(build.OBJDIR, 'flagv_' + name)
    os.makedirs(vdir, exist_ok=True)
    hipcc = build._hipcc()
    fl = build.FLAGS + [f'-DEDVR_VARIANT={name}'] + flags.split()
    procs = [subprocess.Popen([hipcc] + fl + ['-c', os.path.join(build.CSRC, u), '-o', os.path.join(vdir, u.replace('.hip', '.o'))]) for u in (unit, 'api.hip')]
    assert all(p.wait() == 0 for p in procs), name
    mine = (unit.replace('.hip', '.o'), 'api.o')
    objs = [os.path.join(vdir, o) if o in mine else os.path.join(build.OBJDIR, o) for o in (s.replace('.hip', '.o') for s in build.SOURCES)]
    out = os.path.join(build.LIBDIR, 'variants', f'libedvr_amd_{name}.so')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return out


if __name__ == '__main__':
    build.build()
    from concurrent.futures import ThreadPoolExecutor
    args = sys.argv[1:]
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(lambda p: make(*p), zip(args[0::2], args[1::2])):
            print('built', o)
