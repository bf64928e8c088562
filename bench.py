#!/usr/bin/env python
"""bench.py - EDVR hot-path throughput on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode infer|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path (EDVR forward, or forward+backward+Adam with --mode train)
over one batch of synthetic REDS-shaped clips per GPU, inputs resident in HBM before timing.
Clips are independent, so ranks shard them with no data-path collective (inference) or with
the DDP gradient all-reduce over RCCL (training).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     - dominant kernel (fp32 MFMA 3x3 conv): algorithmic FLOPs of every launch of that
                 kernel / its HIP-event time, measured live on the launch stream in an instrumented
                 pass run right after the timed region (same launches, same shapes).
  cpu_baseline - the CPU oracle (oracle/: reference network restated in torch CPU ops + C DCNv2)
                 timed on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32

WORKLOADS = {
    # BASELINE.json metric: "EDVR-L x4 5-frame 720p clips/sec" -> EDVR-L, T=5, 180x320 LR -> 720x1280
    'edvr_l_x4_t5_180x320': dict(net=dict(num_feat=128, num_frame=5, num_reconstruct_block=40, center_frame_idx=None),
                                 shape=(5, 3, 180, 320), batch=4,
                                 desc='EDVR-L x4, 5 frames, 180x320 LR -> 720x1280, batch 4/GPU, inference'),
    # BASELINE.json configs[1]
    'edvr_m_x4_t5_180x320': dict(net=dict(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2),
                                 shape=(5, 3, 180, 320), batch=4,
                                 desc='EDVR-M x4, 5 frames, 180x320 LR -> 720x1280, batch 4/GPU, inference'),
    # BASELINE.json configs[3]: EDVR-L training, 5 frames, 64x64 LR crops, 32 clips per GPU (global 256 on 8)
    'edvr_l_train_t5_64x64': dict(net=dict(num_feat=128, num_frame=5, num_reconstruct_block=40, center_frame_idx=None),
                                  shape=(5, 3, 64, 64), batch=32,
                                  desc='EDVR-L x4 training, 5 frames, 64x64 LR crops (256x256 GT), 32 clips/GPU, '
                                       'Charbonnier(sum) + Adam(4e-4, betas 0.9/0.99), DDP'),
    # BASELINE.json configs[2]: EDVR-L, 7 frames, 180x320, batch 8, TSA on; with --mode train it is the fwd+bwd case
    # (the saved activations of 8 x 7 frames at 128 channels need ~150 GB: sized for the 288 GB of one MI355X)
    'edvr_l_x4_t7_180x320': dict(net=dict(num_feat=128, num_frame=7, num_reconstruct_block=40, center_frame_idx=None),
                                 shape=(7, 3, 180, 320), batch=8,
                                 desc='EDVR-L x4, 7 frames, 180x320 LR -> 720x1280, batch 8/GPU, TSA on'),
    # BASELINE.json configs[4]: EDVR-L deblur (hr_in + predeblur, no upscale), 5 frames of 1280x720, 32 clips over 8 GPUs
    'edvr_l_deblur_t5_720x1280': dict(net=dict(num_feat=128, num_frame=5, num_reconstruct_block=40, center_frame_idx=None,
                                               hr_in=True, with_predeblur=True),
                                      shape=(5, 3, 720, 1280), batch=4, scale=1,
                                      desc='EDVR-L deblur (hr_in, predeblur), 5 frames, 1280x720 -> 1280x720, batch 4/GPU'),
    # BASELINE.json configs[0] (plumbing-sized)
    'edvr_m_x4_t5_64x64': dict(net=dict(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2),
                               shape=(5, 3, 64, 64), batch=1, desc='EDVR-M x4, 5 frames, 64x64 LR crop, batch 1'),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default=None, choices=list(WORKLOADS),
                    help='default: edvr_l_x4_t5_180x320 (infer) / edvr_l_train_t5_64x64 (train)')
    ap.add_argument('--batch', type=int, default=0, help='clips per GPU (default: the workload\'s)')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help='infer: forward clips/s (default).  train: fwd + Charbonnier + bwd + grad all-reduce + Adam')
    ap.add_argument('--optimizer', default='fused', choices=['fused', 'torch'], help='train mode: edvr_amd FusedAdam or torch.optim.Adam')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    return ap.parse_args()


def build_net(cfg, device):
    from edvr_amd import EDVR
    from util_edvr import randomize_offsets
    torch.manual_seed(10)  # options/train/EDVR/*.yml manual_seed: 10
    return randomize_offsets(EDVR(**cfg['net'])).eval().to(device)


def instrumented_pass(net, x, steps):
    """Re-run the step with every conv launch bracketed by events on the launch stream."""
    from edvr_amd import ops
    records = []

    def hook(name, flops, launch, nbytes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        records.append((name, flops, e0, e1, nbytes))

    ops.LAUNCH_HOOK = hook
    try:
        with torch.no_grad():
            for _ in range(steps):
                net(x)
        torch.cuda.synchronize()
    finally:
        ops.LAUNCH_HOOK = None
    per = {}
    for name, flops, e0, e1, nbytes in records:
        d = per.setdefault(name, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += nbytes
    return per


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (scripts/prof_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, calibrated on known-size copies as
    MI355X_MICROARCH.md's HBM section prescribes).  PMC collection serialises kernels, so it is not redone inside the timed run."""
    path = os.path.join(ROOT, 'profiles', 'r1', f'traffic_{workload}.json')
    if not os.path.exists(path):
        return None, None
    rep = json.load(open(path))
    base = kernel.split('<')[0]
    for k, v in rep['kernels'].items():
        if k.split('<')[0].endswith(base) and v.get('hbm_bytes_per_launch'):
            return v, os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(cfg):
    """Oracle forward of ONE clip of the same workload on the host cores (bounded: ~10-30 s)."""
    from oracle import dcn_oracle, edvr_oracle as EO
    from edvr_amd import EDVR
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    sd = randomize_offsets(EDVR(**cfg['net'])).state_dict()
    x = torch.rand(1, *cfg['shape'], generator=torch.Generator().manual_seed(0))
    kw = dict(center=cfg['net'].get('center_frame_idx'))
    with torch.no_grad():
        t0 = time.time()
        EO.edvr_forward(sd, x, dcn=dcn_oracle.dcnv2_c, **kw)
        dt = time.time() - t0
    return dict(value=round(1.0 / dt, 5), unit='clips/s', cores=torch.get_num_threads(), kind='port',
                sample='1 clip (one forward, batch 1) of the same workload: reference network restated in torch CPU ops '
                       '(fp32, oneDNN) + C/OpenMP DCNv2 oracle; the reference itself has no CPU DCN path',
                seconds=round(dt, 2), host_cpus=os.cpu_count())


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (edvr_amd has no CPU path)'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=device)  # nccl == RCCL on ROCm
    from edvr_amd import _lib
    assert _lib.lib().edvr_check_device() == 0, _lib.lib().edvr_last_error().decode()

    if args.workload is None:
        args.workload = 'edvr_l_train_t5_64x64' if args.mode == 'train' else 'edvr_l_x4_t5_180x320'
    cfg = WORKLOADS[args.workload]
    batch = args.batch or cfg['batch']
    net = build_net(cfg, device)
    # per-rank clips (seed + rank, like train.py:53): generated on the CPU, resident in HBM before timing
    x = torch.rand(batch, *cfg['shape'], generator=torch.Generator().manual_seed(rank)).to(device)

    if args.mode == 'train':
        from edvr_amd import dist as D
        from edvr_amd.autograd import charbonnier_loss
        net.train()
        sc = cfg.get('scale', 4)
        gt = torch.rand(batch, 3, sc * cfg['shape'][2], sc * cfg['shape'][3], generator=torch.Generator().manual_seed(1000 + rank)).to(device)
        model = D.wrap_ddp(net)  # RCCL gradient all-reduce, bucketed and overlapped with backward
        from edvr_amd.optim import FusedAdam
        dcn = [p for n, p in net.named_parameters() if 'dcn' in n]  # edvr_model.py:21-53 (two groups as with dcn_lr_mul != 1)
        rest = [p for n, p in net.named_parameters() if 'dcn' not in n]
        groups = [{'params': rest, 'lr': 4e-4}, {'params': dcn, 'lr': 4e-4 * 1}]
        opt = (torch.optim.Adam if args.optimizer == 'torch' else FusedAdam)(groups, lr=4e-4, betas=(0.9, 0.99))

        def step():
            opt.zero_grad(set_to_none=True)
            out = model(x)
            loss = charbonnier_loss(out, gt)
            loss.backward()
            opt.step()
            return loss.detach()
    else:
        def step():
            with torch.no_grad():
                return net(x)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    result = None
    if rank == 0:
        clips = batch * world * args.steps
        if args.workload == 'edvr_l_x4_t5_180x320' and args.mode == 'infer':
            metric = 'EDVR-L x4 5-frame 720p clips/sec'  # BASELINE.json's metric
        elif args.mode == 'train':
            metric = ('EDVR-L x4 training clips/sec (= iters/sec x global batch)' if args.workload == 'edvr_l_train_t5_64x64'
                      else f'{args.workload} fwd+bwd+Adam clips/sec')
        else:
            metric = f'{args.workload} inference clips/sec'
        result = {
            'metric': metric,
            'value': round(clips / elapsed, 4), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic (uniform [0,1) REDS-shaped clips; random-init weights, '
            'manual_seed 10, conv_offset ~ N(0,0.02)/N(0,0.5) so taps are non-integer)',
            'config': {'workload': cfg['desc'], 'clips_per_gpu': batch, 'global_clips': batch * world,
                       'parallelism': (f'DDP x{world}: RCCL gradient all-reduce (82.5 MB fp32)' if args.mode == 'train'
                                       else f'clip-sharded x{world}, no data-path collective')},
        }
        if args.mode == 'train':
            result['iters_per_sec'] = round(args.steps / elapsed, 4)
            result['optimizer'] = ('edvr_amd.optim.FusedAdam (one HIP launch for all tensors; arithmetic of torch.optim.Adam)'
                                   if args.optimizer == 'fused' else 'torch.optim.Adam')
        if not args.no_roofline and args.mode == 'infer':
            per = instrumented_pass(net, x, max(1, min(args.steps, 3)))
            name = max(per, key=lambda k: per[k][2])
            n, flops, secs, nbytes = per[name]
            total_conv_s = sum(v[2] for v in per.values())
            tr, tr_src = measured_traffic(args.workload, name) if batch == cfg['batch'] else (None, None)
            wino = 'winograd' in name
            result['roofline'] = {
                'bound': 'mfma', 'kernel': name, 'achieved': round(flops / secs / 1e12, 2), 'peak': PEAK_F32_MFMA_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(flops / secs / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                'traffic': round(tr['hbm_bytes_per_launch']) if tr else None,
                'traffic_detail': ({'unit': 'bytes per launch (average over the launches of this kernel in one step)',
                                    'fetch': round(tr['fetch_bytes_per_launch']), 'write': round(tr['write_bytes_per_launch']),
                                    'algorithmic': round(nbytes / n), 'source': tr_src} if tr else None),
                # `achieved` counts ALGORITHMIC flops (2*9*Ci*Co per output pixel).  Winograd F(2x2,3x3) issues
                # 16 instead of 36 multiplies per 2x2 tile and channel pair, so the matrix cores execute achieved/2.25:
                'algorithm': 'winograd F(2x2,3x3), fp32' if wino else 'direct implicit GEMM, fp32',
                'mfma_executed_tflops': round(flops / secs / 1e12 / (2.25 if wino else 1.0), 2),
                'mfma_executed_frac_of_peak': round(flops / secs / 1e12 / (2.25 if wino else 1.0) / PEAK_F32_MFMA_TFLOPS, 4),
                'launches': n, 'avg_launch_us': round(secs / n * 1e6, 2), 'gflop_per_launch': round(flops / n / 1e9, 3),
                'all_conv_kernels': {k: {'launches': v[0], 'tflops': round(v[1] / v[2] / 1e12, 2), 'ms': round(v[2] * 1e3, 3)}
                                     for k, v in sorted(per.items(), key=lambda kv: -kv[1][2])},
                'conv_time_share_of_step': round(total_conv_s / max(1, min(args.steps, 3)) / (elapsed / args.steps), 3),
            }
        if not args.no_cpu_baseline and args.mode == 'infer' and world == 1:  # the CPU leg runs at N = 1 only (rank 0's host cores)
            result['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(result), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
