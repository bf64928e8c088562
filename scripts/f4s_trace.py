"""Barrier timeline of workgroup 0 of the split-operand F(4x4) kernel (variant built with -DF4S_TRACE): per barrier, each wave's arrival
relative to the release, in core cycles.  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_f4strace.so python scripts/f4s_trace.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops, _lib
dev = torch.device('cuda')
x = torch.randn(20, 128, 180, 320, device=dev); w = torch.randn(128, 128, 3, 3, device=dev) * 0.05; b = torch.randn(128, device=dev)
wpk, wf4s = ops.pack_conv_weight(w), ops.pack_conv_weight(w, f4s=True)
bound = ops.amax(x)
for _ in range(3): ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, x_amax=bound, algo=ops.CONV_WINOGRAD_F4S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv2d(x, wpk, b, 128, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, x_amax=bound, algo=ops.CONV_WINOGRAD_F4S)
e1.record(); torch.cuda.synchronize()
print(f'n = 20 layer with the stamps in: {e0.elapsed_time(e1) / 10:.3f} ms per launch')
L = _lib.lib()
SL = L.edvr_f4s_trace_slots()
N = 16 * 2 * SL + 16 * 8
buf = (ctypes.c_uint * N)()
L.edvr_f4s_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', L.edvr_f4s_trace_read(buf, N))
t0 = buf[1]  # release of barrier 0, wave 0
u = lambda v: (v - t0) & 0xffffffff
prev = None
for slot in range(SL):
    arr = [u(buf[(slot * 16 + wv) * 2 + 0]) for wv in range(16)]
    rel = min(u(buf[(slot * 16 + wv) * 2 + 1]) for wv in range(16))
    step = '' if prev is None else f' step {rel - prev:6d}'
    print(f'barrier {slot:3d}: release {rel:8d}{step} | arrival - release: staging ' + ' '.join(f'{a - rel:6d}' for a in arr[:4]) +
          ' | multiplying ' + ' '.join(f'{a - rel:6d}' for a in arr[4:]))
    prev = rel
rel5 = min(u(buf[(5 * 16 + wv) * 2 + 1]) for wv in range(16))
print('positions of the chunk step after barrier 5 (cycles after its release), per multiplying wave:')
for wv in range(4, 16):
    print(f'  wave {wv:2d} (SIMD {wv % 4}): ' + ' '.join(f'{u(buf[16 * 2 * SL + wv * 8 + c]) - rel5:6d}' for c in range(6)))
print('the same chunk step in the staging waves: patch in registers | next DMA issued | column transform | rows 0-2 committed | rows 3-5 committed')
for wv in range(4):
    print(f'  wave {wv:2d} (SIMD {wv % 4}): ' + ' '.join(f'{u(buf[16 * 2 * SL + wv * 8 + c]) - rel5:6d}' for c in range(5)))
