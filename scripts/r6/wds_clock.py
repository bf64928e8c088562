import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from edvr_amd import ops, _lib
dev = torch.device('cuda')
L = _lib.lib()
L.edvr_wds_clock_read.argtypes = [ctypes.c_void_p]
for (n, ci, h, w, co) in [(160, 128, 64, 64, 128), (20, 128, 180, 320, 128)]:
    x = torch.randn(n, ci, h, w, device=dev); dz = torch.randn(n, co, h, w, device=dev) * 1e-3
    ops.input_bound(x); ops.input_bound(dz); ops.set_f4s(training=True)
    for _ in range(5): ops.conv2d_wgrad(x, None, None, dz, co, 3, 1)
    torch.cuda.synchronize()
    v = ctypes.c_ulonglong(0); L.edvr_wds_clock_read(ctypes.byref(v))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for _ in range(R): ops.conv2d_wgrad(x, None, None, dz, co, 3, 1)
    e1.record(); torch.cuda.synchronize()
    L.edvr_wds_clock_read(ctypes.byref(v))
    ms = e0.elapsed_time(e1) / R
    print(f'{n}x{ci}x{h}x{w}: {ms:.3f} ms per call (kernel + reduction); longest workgroup {v.value} cycles of s_memtime = {v.value / (ms * 1e3):.0f} cycles/us if the kernel were the whole call')
