"""Barrier timeline of workgroup 0 of the F(4x4) kernel (variant built with -DF4_EXP_TRACE): per barrier, each wave's arrival and
release in core cycles relative to the first stamp.  EDVR_AMD_LIB=.../libedvr_amd_trace.so python scripts/f4_trace.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops, _lib
dev = torch.device('cuda')
x = torch.randn(20, 128, 180, 320, device=dev); w = torch.randn(128, 128, 3, 3, device=dev) * 0.05; b = torch.randn(128, device=dev)
wpk, wf4 = ops.pack_conv_weight(w), ops.pack_conv_weight(w, f4=True)
for _ in range(3): ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4=wf4, algo=ops.CONV_WINOGRAD_F4)
torch.cuda.synchronize()
N = 16 * 2 * 512
buf = (ctypes.c_longlong * N)()
L = _lib.lib()
L.edvr_f4_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', L.edvr_f4_trace_read(buf, N))
NP = 4 * 4 * 512
bufp = (ctypes.c_longlong * NP)()
L.edvr_f4_trace_read_p.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.edvr_f4_trace_read_p(bufp, NP)
t0 = min(v for v in buf if v > 0)
nslots = 60
for slot in range(nslots):
    arr = [buf[(slot * 16 + wv) * 2 + 0] - t0 for wv in range(16)]
    rel = [buf[(slot * 16 + wv) * 2 + 1] - t0 for wv in range(16)]
    print(f'barrier {slot:3d}: release {min(rel):8d}  | arrival - release: prod ' + ' '.join(f'{a - min(rel):6d}' for a in arr[:4]) +
          ' | cons ' + ' '.join(f'{a - min(rel):6d}' for a in arr[4:]))
    if slot >= 1:
        prev_rel = min(buf[((slot - 1) * 16 + wv) * 2 + 1] - t0 for wv in range(16))
        print('      producer stamps after previous release (patches-in-regs, dma-issued, transformed+written, dma-landed, arrival): ' +
              ' | '.join(' '.join(f'{bufp[(slot * 4 + wv) * 4 + i] - t0 - prev_rel:5d}' for i in range(4)) + f' {arr[wv] - prev_rel:5d}' for wv in range(4)))
