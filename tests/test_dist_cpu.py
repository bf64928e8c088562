"""CPU, world_size 2, gloo: the multi-GPU host logic (clip sharding, gradient all-reduce through DDP)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from edvr_amd import dist as D
    r, w = D.init_dist(backend='gloo')
    assert (r, w) == (rank, world) and D.get_dist_info() == (rank, world)
    mine = D.shard_indices(11)
    per = D.shard_batch(7)
    # gradient averaging: DDP over gloo == mean of per-rank gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, 1, 1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(4, 3, 3, 1, 1))
    ddp = D.wrap_ddp(net)
    x = torch.rand(2, 3, 8, 8, generator=torch.Generator().manual_seed(100 + rank))
    ddp(x).square().sum().backward()
    grads = [p.grad.flatten().tolist() for p in net.parameters()]  # plain lists: tensors cannot outlive the worker
    m = D.reduce_scalar(float(rank + 1), torch.device('cpu'))
    q.put((rank, mine, per, grads, m))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gradient_allreduce():
    world, port = 2, 29533 + os.getpid() % 1000
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, idx0, per0, g0, m0), (r1, idx1, per1, g1, m1) = got
    assert sorted(idx0 + idx1) == list(range(11)) and not set(idx0) & set(idx1)  # disjoint, complete
    assert idx0 == list(range(0, 11, 2)) and idx1 == list(range(1, 11, 2))
    assert per0 + per1 == 7 and per0 == 4
    assert m0 == m1 == 1.5
    # single-process reference: mean over the two ranks' gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, 1, 1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(4, 3, 3, 1, 1))
    acc = None
    for rank in range(world):
        net.zero_grad()
        x = torch.rand(2, 3, 8, 8, generator=torch.Generator().manual_seed(100 + rank))
        net(x).square().sum().backward()
        g = [p.grad.clone() for p in net.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for a, b0, b1 in zip(acc, g0, g1):
        assert torch.allclose((a / world).flatten(), torch.tensor(b0), atol=1e-6) and b0 == b1


def test_single_process_defaults():
    from edvr_amd import dist as D
    assert D.get_dist_info() == (0, 1)
    assert D.shard_indices(5) == [0, 1, 2, 3, 4] and D.shard_batch(9) == 9
    net = torch.nn.Linear(2, 2)
    assert D.wrap_ddp(net) is net


# ---------------------------------------------------------------------------------------------- bench.py's multi-GPU contract
def _bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(root, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_gpus_8_builds_the_torchrun_launch():
    """`python bench.py --gpus 8 --steps 20 --warmup 5` outside torchrun re-launches itself as the driver would launch it:
    one node, 8 ranks, rendezvous on 127.0.0.1, its own flags passed through, dmabuf IPC mode in the environment."""
    import sys
    b = _bench()
    cmd = b.spawn_command(8, 29512, ['--gpus', '8', '--steps', '20', '--warmup', '5'])
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29512'
    script = cmd.index(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
    assert b.spawn_env({'A': '1'}) == {'A': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'}


def test_bench_leg_plan_never_lets_one_rank_skip_a_collective_leg():
    """The legs beside the timed region (bench.leg_plan): at N > 1 the legs that hold barriers run on EVERY rank, the legs that can fail on
    one rank alone (`configs`: out of memory) or use the host's cores (CPU oracle, stock-ops arm) do not run at all, and nothing is
    'rank0'-only unless it is collective-free - so no rank can be left waiting in a collective."""
    import sys
    from unittest import mock
    b = _bench()
    with mock.patch.object(sys, 'argv', ['bench.py']):
        args = b.parse()
    args.workload = 'edvr_l_x4_t5_180x320'
    one, eight = b.leg_plan(args, 1, 10), b.leg_plan(args, 8, 10)
    assert all(one[k] for k in one), one  # the default run at N = 1: every leg
    assert one['cpu_baseline'] == one['stock_rocm_baseline'] == one['roofline'] == 'rank0'
    for leg in ('fp32_mfma', 'batch4', 'target_4k', 'trained_like', 'train'):  # barriers / DDP collectives inside: all ranks or none
        assert one[leg] == eight[leg] == 'all', (leg, one[leg], eight[leg])
    assert eight['configs'] is None and eight['cpu_baseline'] is None and eight['stock_rocm_baseline'] is None
    assert eight['roofline'] == 'rank0'  # inference: an instrumented forward has no collective
    args.mode, args.workload = 'train', 'edvr_l_train_t5_64x64'
    t8 = b.leg_plan(args, 8, 32)
    assert t8['roofline'] is None  # an extra DDP step on rank 0 alone would wait for its peers
    assert t8['fp32_mfma'] == 'all' and t8['train'] is None and t8['configs'] is None and t8['cpu_baseline'] is None
    # the compact line of an N = 8 run carries the world size and the backend
    line = __import__('json').loads(b.compact_line({'metric': 'm', 'value': 1.0, 'unit': 'clips/s', 'n_gpus': 8, 'steps': 1, 'warmup': 0, 'ms_per_step': 1.0,
                                                    'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': b.DTYPE, 'data': 'synthetic',
                                                    'config': {'workload': 'w', 'world_size': 8, 'backend': 'RCCL (torch.distributed nccl)'}}))
    assert line['n_gpus'] == 8 and line['config']['world_size'] == 8 and 'RCCL' in line['config']['backend']


def _bench_worker(rank, world, port, q):
    import contextlib
    import io
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group(backend='gloo')
    b = _bench()
    calls = []

    def step():  # rank 1 is the slow rank: the reported time must be ITS time on every rank
        calls.append(1)
        time.sleep(0.02 * (1 + 2 * rank))
        return torch.ones(1)
    elapsed = b.timed(step, 5, 2, dist, torch.device('cpu'))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        b.emit({'value': 1.0, 'n_gpus': world}, rank, os.path.join(os.environ.get('TMPDIR', '/tmp'), f'bench_full_test_{os.getpid()}.json'))
    q.put((rank, elapsed, len(calls), buf.getvalue()))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timing_is_max_over_ranks_and_only_rank0_prints():
    import json
    world, port = 2, 30533 + os.getpid() % 1000
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, e0, n0, out0), (_, e1, n1, out1) = got
    assert n0 == n1 == 7                      # 2 warm-up + exactly 5 timed steps
    assert e0 == e1 and e0 >= 5 * 0.06 * 0.9  # both ranks report the slow rank's time (5 x 60 ms)
    line = json.loads(out0)                   # the compact contract line: one line, rank 0 only
    assert line['value'] == 1.0 and line['n_gpus'] == 2 and out0.count('\n') == 1
    assert out1 == ''                         # rank > 0 prints nothing
