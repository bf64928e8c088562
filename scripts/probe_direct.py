"""Direct-kernel probe on the EDVR-L (128-channel) shapes of the small test configs: every (ks, ci, co, h, w) against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from edvr_amd import ops
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
for (ks, ci, co, n, h, w) in [(1, 896, 128, 1, 32, 48), (1, 128, 896, 1, 32, 48), (1, 256, 128, 1, 16, 24), (1, 128, 256, 1, 16, 24), (1, 128, 128, 1, 8, 12),
                              (3, 128, 128, 7, 32, 48), (3, 128, 128, 7, 16, 24), (3, 128, 128, 7, 8, 12), (3, 256, 128, 7, 32, 48),
                              (3, 128, 256, 7, 32, 48), (3, 128, 256, 7, 8, 12), (3, 128, 216, 7, 32, 48), (3, 216, 128, 7, 32, 48), (3, 216, 128, 7, 8, 12),
                              (3, 128, 512, 1, 32, 48), (3, 512, 128, 1, 32, 48), (3, 64, 128, 1, 64, 96), (3, 128, 64, 1, 64, 96)]:
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, ks, ks, generator=g) * 0.05
    ref = F.conv2d(x.double(), wt.double(), None, 1, ks // 2)
    for tf in (False, True):
        if tf:  # data-gradient packing of the transposed weight: conv with w'[c][o] = flip(w[o][c]) -> compare against conv_transpose
            wt2 = torch.randn(ci, co, ks, ks, generator=g) * 0.05  # forward weight (ci_fwd = co here)
            refd = F.conv_transpose2d(x.double(), wt2.double().transpose(0, 1).transpose(0, 1), None, 1, ks // 2) if False else None
            wpk = ops.pack_conv_weight(wt2.to(dev), transpose_flip=True)  # dgrad of a conv with weight (ci, co): maps ci -> co channels
            refd = F.conv_transpose2d(x.double(), wt2.double(), None, 1, ks // 2)
            for name, algo in (('direct', ops.CONV_DIRECT), ('auto', ops.CONV_AUTO)):
                y = ops.conv2d(x.to(dev), wpk, None, co, ks, algo=algo)
                e = ((y.double().cpu() - refd).abs().max() / refd.abs().max()).item()
                print(f'dgrad ks{ks} {ci}->{co} n{n} {h}x{w} {name}: {e:.2e}' + ('   <<<<' if e > 2e-5 else ''), flush=True)
        else:
            wpk = ops.pack_conv_weight(wt.to(dev))
            for name, algo in (('direct', ops.CONV_DIRECT), ('auto', ops.CONV_AUTO)):
                y = ops.conv2d(x.to(dev), wpk, None, co, ks, algo=algo)
                e = ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item()
                print(f'fwd   ks{ks} {ci}->{co} n{n} {h}x{w} {name}: {e:.2e}' + ('   <<<<' if e > 2e-5 else ''), flush=True)
