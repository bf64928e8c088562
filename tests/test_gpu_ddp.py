"""-m gpu: clip-level data parallelism THROUGH THE PRODUCT - 2 ranks share the one GPU of the test box, gradients are
all-reduced by DistributedDataParallel over gloo (the RCCL path needs one GPU per rank; the driver's 8-GPU run covers it).

What the reference does (basicsr/models/base_model.py:63-69, options/train/EDVR/train_EDVR_L_x4_SR_REDS.yml:120
`find_unused_parameters: true`, edvr_model.py:55-69 TSA warm-up): wrap the net in DDP with find_unused_parameters, freeze
everything but the TSA fusion module for the first iterations, unfreeze later.  Checked here with edvr_amd's EDVR, whose
backward runs in custom autograd Functions: the DDP-averaged gradients of both phases equal the mean of the per-rank gradients
computed by one process, frozen parameters receive none.
"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
NET = dict(num_feat=32, num_frame=3, num_reconstruct_block=2, center_frame_idx=1, deformable_groups=4)
SHAPE = (2, 3, 3, 32, 32)


def _build():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from edvr_amd import EDVR
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    return randomize_offsets(EDVR(**NET)).train()


def _data(rank):
    x = torch.rand(*SHAPE, generator=torch.Generator().manual_seed(100 + rank))
    gt = torch.rand(SHAPE[0], 3, 4 * SHAPE[3], 4 * SHAPE[4], generator=torch.Generator().manual_seed(200 + rank))
    return x, gt


def _grads(net):
    return {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in net.named_parameters()}


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from edvr_amd import dist as D
    from edvr_amd.autograd import charbonnier_loss
    from edvr_amd.optim import tsa_freeze_schedule
    torch.cuda.set_device(0)
    dist.init_process_group(backend='gloo')  # CUDA tensors over gloo: both ranks live on cuda:0
    dev = torch.device('cuda:0')
    net = _build().to(dev)
    ddp = D.wrap_ddp(net, find_unused_parameters=True)  # wrapped with every parameter trainable, as base_model.py:63-69
    assert type(ddp).__name__ == 'DistributedDataParallel'
    tsa_freeze_schedule(ddp, 1, tsa_iter=3)  # iteration 1: only `fusion.*` trains (edvr_model.py:57-62)
    x, gt = _data(rank)
    x, gt = x.to(dev), gt.to(dev)
    out = {}
    for it, phase in ((1, 'tsa_only'), (3, 'all')):
        if it == 3:
            assert tsa_freeze_schedule(ddp, 3, tsa_iter=3)
            ddp = D.rewrap_ddp(ddp, find_unused_parameters=False)  # (the reference flips the flag in place: breaks on PyTorch 2.x)
        ddp.zero_grad(set_to_none=True)
        charbonnier_loss(ddp(x), gt).backward()
        out[phase] = {k: (None if v is None else v.numpy()) for k, v in _grads(net).items()}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_one_gpu_tsa_warmup(gpu):
    world, port = 2, 29611 + os.getpid() % 1000
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    import time
    got, deadline = {}, time.time() + 420
    while len(got) < world:
        assert time.time() < deadline, 'DDP workers did not report in time'  # fail fast if a worker died (a blocking get would sit out its whole timeout on the GPU box)
        try:
            rank, out = q.get(timeout=5)
            got[rank] = out
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f'DDP worker exited with {dead}'
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single-process reference: mean of the two ranks' gradients
    from edvr_amd.autograd import charbonnier_loss
    from edvr_amd.optim import tsa_freeze_schedule
    net = _build().to(gpu)
    for it, phase in ((1, 'tsa_only'), (3, 'all')):
        tsa_freeze_schedule(net, it, tsa_iter=3)
        acc = None
        for rank in range(world):
            net.zero_grad(set_to_none=True)
            x, gt = _data(rank)
            charbonnier_loss(net(x.to(gpu)), gt.to(gpu)).backward()
            g = _grads(net)
            acc = g if acc is None else {k: (None if v is None else acc[k] + v) for k, v in g.items()}
        n_train = 0
        for k, v in acc.items():
            r0, r1 = got[0][phase][k], got[1][phase][k]
            if v is None:
                assert r0 is None and r1 is None, (phase, k)
                continue
            n_train += 1
            mean = (v / world).numpy()
            assert (r0 == r1).all(), (phase, k)  # every rank holds the same averaged gradient
            scale = max(abs(mean).max(), 1e-30)
            # dx of the DCNs accumulates by fp32 atomics (summation order differs run to run), hence a tolerance
            assert abs(r0 - mean).max() / scale < 2e-3, (phase, k, abs(r0 - mean).max() / scale)
        if phase == 'tsa_only':
            assert 0 < n_train < len(acc) and all(('fusion' in k) == (v is not None) for k, v in acc.items())
        else:
            assert n_train == len(acc)
