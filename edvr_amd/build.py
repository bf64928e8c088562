"""Build libedvr_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m edvr_amd.build [--force]

One object per translation unit (compiled in parallel), linked into
edvr_amd/lib/libedvr_amd.so.  hipcc cross-compiles gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libedvr_amd.so')
OBJDIR = os.path.join(HERE, 'build')
SOURCES = ['api.hip', 'pack.hip', 'conv2d.hip', 'dcn.hip', 'dcn_any.hip', 'elementwise.hip', 'wgrad.hip', 'backward.hip', 'dcn_fused.hip', 'dcn_tapwin.hip', 'dcn_tapwin_s.hip', 'dcn_bwd_fused.hip', 'winograd.hip', 'winograd_f4.hip', 'winograd_f4s.hip', 'winograd_wgrad.hip', 'winograd_wgrad_s.hip', 'wgrad_direct_s.hip', 'gemm_nt_s.hip', 'optim.hip', 'metrics.hip', 'conv_small.hip', 'conv1x1.hip', 'conv1x1_s.hip', 'data.hip']
LINK = []  # no library dependencies beyond the HIP runtime: every kernel of the path is in csrc/
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-ffp-contract=fast']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: the HIP extension cannot be built')
    return exe


def _deps():
    return [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'pack.h'), os.path.join(CSRC, 'dcn_tap.h'), os.path.join(HERE, '..', 'include', 'edvr_amd.h')]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def source_hash():
    """sha256[:16] over the kernel sources (csrc/*, include/edvr_amd.h): recorded in every profiles/*.json so that a committed
    measurement can be tied to - or flagged as older than - the kernels in the tree (bench.py `traffic_detail.stale`)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h', '.inc')))
    for path in files + [os.path.join(HERE, '..', 'include', 'edvr_amd.h')]:
        h.update(os.path.basename(path).encode() + b'\0')
        h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + _deps()):
            jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + LINK)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
