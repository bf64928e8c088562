import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from edvr_amd import ops
from util_edvr import build
ops.BOUND_CHECK = True
for cfg in ['M_T5', 'L_T7']:
    net, x, _ = build(cfg)
    net = net.cuda(); x = x.cuda()
    try:
        with torch.no_grad():
            y = net(x)
        print(cfg, 'forward ok, finite:', torch.isfinite(y).all().item(), 'loosest bounds:', sorted(ops.BOUND_CHECK_LOG, key=lambda r: -r[1])[:5])
    except AssertionError as e:
        import traceback; traceback.print_exc()
    ops.BOUND_CHECK_LOG.clear()
    net.train()
    try:
        net(x).sum().backward()
        print(cfg, 'train ok; loosest:', sorted(ops.BOUND_CHECK_LOG, key=lambda r: -r[1])[:5])
    except AssertionError as e:
        import traceback; traceback.print_exc()
    ops.BOUND_CHECK_LOG.clear()
