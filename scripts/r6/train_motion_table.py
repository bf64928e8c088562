"""Per-kernel table of the EDVR-L training step on the sub-pixel field and on the motion-like field (where do the extra ms go?)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench as B
from util_edvr import motion_frames, motion_like_offsets
dev = torch.device('cuda')
from edvr_amd import ops
ops.HINT_WAIT = True
cfg = B.WORKLOADS['edvr_l_train_t5_64x64']
tabs = {}
for name in ('sub-pixel', 'motion'):
    if name == 'motion':
        net = B.build_net(cfg, dev, offset_bias_sigma=3.0)
        x = motion_frames(cfg['batch'], cfg['shape'], seed=0).to(dev)
        motion_like_offsets(net, x, target_rough=0.5, bias_sigma=3.0)
        step = B.make_train_step(net, cfg, cfg['batch'], dev, 0, 'fused', x=x, lr=1e-6)
    else:
        net = B.build_net(cfg, dev)
        step = B.make_train_step(net, cfg, cfg['batch'], dev, 0, 'fused')
    el = B.timed(step, 8, 3, None, dev)
    per = B.instrumented_pass(step, 2)
    tab = B.kernel_table(per, 2, el / 8)
    tabs[name] = (el / 8, tab)
    del net, step
    torch.cuda.empty_cache()
a, b = tabs['sub-pixel'], tabs['motion']
print(f'ms per iteration: sub-pixel {a[0] * 1e3:.1f}, motion {b[0] * 1e3:.1f}')
keys = sorted(set(a[1]) | set(b[1]), key=lambda k: -(b[1].get(k, {}).get('ms_per_step', 0) - a[1].get(k, {}).get('ms_per_step', 0)))
for k in keys[:14]:
    ma, mb = a[1].get(k, {}).get('ms_per_step', 0.0), b[1].get(k, {}).get('ms_per_step', 0.0)
    print(f'  {k[:64]:64s} {ma:8.2f} -> {mb:8.2f} ms  ({mb - ma:+.2f})')
