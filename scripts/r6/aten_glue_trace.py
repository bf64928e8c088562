"""Which Python lines launch the ATen kernels (fan-in adds, copies, fills) of one EDVR-L training iteration."""
import os, sys, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
cfg = B.WORKLOADS['edvr_l_train_t5_64x64']
net = B.build_net(cfg, dev)
step = B.make_train_step(net, cfg, cfg['batch'], dev, 0, 'fused')
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0, set()])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.device_time_total > 0 and ev.name in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::mul', 'aten::cat', 'aten::sum', 'aten::maximum', 'aten::max', 'aten::abs'):
        st = [s for s in ev.stack if 'edvr_amd' in s or 'bench.py' in s]
        key = (ev.name, str(ev.input_shapes)[:70], st[0][-70:] if st else 'autograd engine / torch')
        r = rows[key]
        r[0] += 1; r[1] += ev.device_time_total
tot = 0.0
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{v[1] / 1e3:7.3f} ms {v[0]:4d}x  {k[0]:16s} {k[1]:72s} {k[2]}')
    tot += v[1]
print(f'listed: {tot / 1e3:.2f} ms of device time')
