"""Offset statistics and forward time of the motion-like field on the headline shape and on the training shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from edvr_amd import EDVR
from util_edvr import motion_frames, motion_like_offsets
dev = torch.device('cuda')
for shape, b in [((5, 3, 180, 320), 10), ((5, 3, 64, 64), 32)]:
    for cap in (4.0, 6.0, 10.0):
        torch.manual_seed(10)
        net = EDVR(num_feat=128, num_frame=5, num_reconstruct_block=40).eval().to(dev)
        x = motion_frames(b, shape, seed=0).to(dev)
        st = motion_like_offsets(net, x, target_rough=0.5, bias_sigma=3.0, absmean_cap=cap)
        with torch.no_grad():
            for _ in range(2): net(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): net(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(shape, 'cap', cap, [(round(a, 2), round(r, 3)) for a, r in st], f'{dt * 1e3:.1f} ms/forward', flush=True)
