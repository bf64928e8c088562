"""Forward time of the motion-like field with the split kernels off (fp32 tap-window DCN forward): the fix-up pass of dcn_tapwin.hip."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from edvr_amd import EDVR, ops
from util_edvr import motion_frames, motion_like_offsets
dev = torch.device('cuda')
torch.manual_seed(10)
net = EDVR(num_feat=128, num_frame=5, num_reconstruct_block=40).eval().to(dev)
x = motion_frames(10, (5, 3, 180, 320), seed=0).to(dev)
st = motion_like_offsets(net, x, target_rough=0.5, bias_sigma=3.0)
for split in (True, False):
    ops.set_f4s(split, split)
    with torch.no_grad():
        for _ in range(2): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): net(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print('split kernels' if split else 'fp32 kernels ', f'{dt * 1e3:.1f} ms/forward', [(round(a, 2), round(r, 3)) for a, r in st], flush=True)
