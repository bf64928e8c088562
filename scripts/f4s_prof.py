"""Where the waves of the split-operand F(4x4) kernel spend their cycles (variant built with -DF4S_PROF=1|2, scripts/exp/f4s_flags.py):
per wave class, s_memtime cycles waiting at the chunk barriers / epilogue barriers, in the chunk loops / epilogues, waiting for the LDS-DMA.
    EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_prof.so python scripts/f4s_prof.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edvr_amd import ops, _lib
dev = torch.device('cuda')
L = _lib.lib()
L.edvr_f4s_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for n, c, co in [(50, 128, 128), (20, 64, 64)]:
    x = torch.randn(n, c, 180, 320, device=dev); w = torch.randn(co, c, 3, 3, device=dev) * 0.05; b = torch.randn(co, device=dev)
    wpk, wf4s = ops.pack_conv_weight(w), ops.pack_conv_weight(w, f4s=True)
    bound = ops.amax(x)
    run = lambda: ops.conv2d(x, wpk, b, co, 3, act=ops.ACT_LRELU, wpk_f4s=wf4s, x_amax=bound, algo=ops.CONV_WINOGRAD_F4S)
    for _ in range(3): run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    L.edvr_f4s_prof_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 10
    e0.record()
    for _ in range(R): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    L.edvr_f4s_prof_read(buf, 1)
    wgs = 256
    v = [[buf[wv * 8 + i] / (R * wgs) for i in range(8)] for wv in range(16)]
    print(f'{n}x{c}x180x320 -> {co}: {ms:.3f} ms per launch with the counters in; cycles per workgroup and launch (mean over 256 workgroups)')
    print('  wave     chunk loops  of which barrier wait   epilogues  of which barrier wait   DMA wait   B wait   A wait')
    for wv in range(16):
        r = v[wv]
        print(f'  {wv:2d} {"stage" if wv < 4 else "mult ":5s} {r[2]:11.0f} {r[0]:11.0f} ({r[0] / max(r[2], 1):4.0%}) {r[3]:11.0f} {r[1]:11.0f} ({r[1] / max(r[3], 1):4.0%}) {r[4]:10.0f} {r[5]:9.0f} {r[6]:8.0f}')
    tot = v[0][2] + v[0][3]
    print(f'  total per workgroup {tot:.0f} cycles = {tot / (ms * 1e3):.0f} cycles/us')
