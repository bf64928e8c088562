#!/usr/bin/env python
"""bench.py - EDVR hot-path throughput on MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode infer|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one pass of the hot path (EDVR forward, or forward+backward+Adam with --mode train) over one batch of synthetic
REDS-shaped clips per GPU, inputs resident in HBM before timing.  Clips are independent, so ranks shard them with no data-path
collective (inference) or with the DDP gradient all-reduce over RCCL (training).  `--gpus N` without a torchrun environment
re-launches this script as N ranks through torch.distributed.run.  Rank 0 prints ONE JSON line - the COMPACT line (`compact_line`: the
contract's fields, `roofline` with the measured `traffic`, `cpu_baseline`, `parity` and the headline number of every leg; < 6 KB,
strict JSON, the last line of stdout) - and writes the full report described below to `bench_full.json` beside this file.
`leg_plan()` says which legs run on which ranks (a leg with barriers runs on every rank or on none).

Objects of the full report besides the contract's fields (everything below runs OUTSIDE the timed region):
  roofline            dominant kernel of the timed step, from an instrumented pass (every launch bracketed by HIP events on the
                      launch stream): `achieved` = flops the matrix cores EXECUTE (MFMA issues x 4096, tile and channel padding
                      included - edvr_conv2d_executed_flops, the quantity SQ_INSTS_VALU_MFMA_MOPS_F32 counts) / time, `frac` =
                      achieved / 157.3 TF/s (<= 1); the algorithmic figure (2*9*Ci*Co per pixel, what SURVEY 8(d) counts) and the
                      padding-free executed figure (algorithmic / 4 for F(4x4)) are kept next to it.
  kernels             per-kernel table of that pass: launches, ms, TF/s (MFMA-bound) or GB/s (HBM-bound), share of the step.
  target_4k           BASELINE.json north_star's target workload - EDVR-L x4, 5 frames, 720x1280 -> 2880x5120, 1 clip per GPU -
                      timed the same way (barrier + synchronize, max over ranks), with its F(4x4) roofline fraction and parity
                      (max rel err, dPSNR) against the stock-PyTorch-ROCm arm on the same clip.
  train               BASELINE.json's second headline (training iters/sec) on the cfg4 per-GPU shape, run on ALL ranks (DDP), with
                      `trained_like` (the same step with multi-pixel per-tap offsets: what the DCN backward costs there) and
                      `parity`: the same 2-clip batch and weights through oracle/edvr_oracle.py in stock fp32 torch ops + torch
                      autograd on this GPU - loss and every parameter gradient compared (the bounds of tests/test_gpu_train.py).
  trained_like        the headline workload with the offsets a TRAINED EDVR predicts: conv_offset.bias ~ N(0, 4^2) / N(0, 10^2) per
                      channel (every (group, tap) has its own multi-pixel displacement, mean |offset| ~ 3.2 / 8 px) on top of the
                      same N(0, 0.02) weights (spatially coherent field): clips/s, the DCN forward's roofline fraction, parity
                      against the stock-ops arm.  The reference's gather costs the same at any offset (.cu:570-633).
                      `motion`: the same with offsets that also VARY IN SPACE as a trained model's do (structured clips - smooth
                      drifting background, moving rectangles, texture - and offset convs rescaled to ~0.5 px of neighbour
                      difference per DCN layer under a 6 px cap on the mean offset: tests/util_edvr.py); `train.motion` likewise.
  configs             one-line results for the other BASELINE.json configs that fit one GPU: [1] EDVR-M batch 4, [2] EDVR-L T7
                      batch 8 forward and forward+backward+Adam, [4] deblur 720p batch 4.
  parity              the headline workload's output on ONE clip vs the CPU oracle's output on the same clip.
  cpu_baseline        the CPU oracle timed on that clip (median of 3) on this box's host cores.
  stock_rocm_baseline SURVEY 8(d)'s second arm: the same network in stock PyTorch-ROCm ops (MIOpen / rocBLAS) with a pure-torch
                      DCNv2 on this GPU - "what a user gets today" - and a third parity witness.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_HBM_GBPS = 8000.0        # same guide: HBM3E ~8 TB/s
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_f16, dense (16x the fp32 rate)
PROFILE_ROUND = 'r6'

L5 = dict(num_feat=128, num_frame=5, num_reconstruct_block=40, center_frame_idx=None)
WORKLOADS = {
    # BASELINE.json metric: "EDVR-L x4 5-frame 720p clips/sec" -> EDVR-L, T=5, 180x320 LR -> 720x1280
    # 10 clips per GPU: a trunk launch then has 4600 Winograd items = 17.97 rounds of the 256 persistent workgroups (4 clips: 1840 =
    # 7.19 rounds executed as 8, a 10 % tail in 81 % of the step).  Batch 4 (round 1's setting, BASELINE configs[1]'s batch) is
    # measured next to it (`batch4` in the JSON line).
    'edvr_l_x4_t5_180x320': dict(net=L5, shape=(5, 3, 180, 320), batch=10,
                                 desc='EDVR-L x4, 5 frames, 180x320 LR -> 720x1280, batch 10/GPU, inference'),
    # BASELINE.json north_star "Target": x4 720p -> 4K, 5 frames, EDVR-L (67.7 TFLOP per clip)
    'edvr_l_x4_t5_720x1280': dict(net=L5, shape=(5, 3, 720, 1280), batch=1,
                                  desc='EDVR-L x4, 5 frames, 720x1280 LR -> 2880x5120 (4K), batch 1/GPU, inference'),
    # BASELINE.json configs[1]
    'edvr_m_x4_t5_180x320': dict(net=dict(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2),
                                 shape=(5, 3, 180, 320), batch=4,
                                 desc='EDVR-M x4, 5 frames, 180x320 LR -> 720x1280, batch 4/GPU, inference'),
    # BASELINE.json configs[3]: EDVR-L training, 5 frames, 64x64 LR crops, 32 clips per GPU (global 256 on 8)
    'edvr_l_train_t5_64x64': dict(net=L5, shape=(5, 3, 64, 64), batch=32,
                                  desc='EDVR-L x4 training, 5 frames, 64x64 LR crops (256x256 GT), 32 clips/GPU, '
                                       'Charbonnier(sum) + Adam(4e-4, betas 0.9/0.99), DDP'),
    # BASELINE.json configs[2]: EDVR-L, 7 frames, 180x320, batch 8, TSA on; with --mode train it is the fwd+bwd case
    # (the saved activations of 8 x 7 frames at 128 channels need ~150 GB: sized for the 288 GB of one MI355X)
    'edvr_l_x4_t7_180x320': dict(net=dict(num_feat=128, num_frame=7, num_reconstruct_block=40, center_frame_idx=None),
                                 shape=(7, 3, 180, 320), batch=8,
                                 desc='EDVR-L x4, 7 frames, 180x320 LR -> 720x1280, batch 8/GPU, TSA on'),
    # BASELINE.json configs[4]: EDVR-L deblur (hr_in + predeblur, no upscale), 5 frames of 1280x720, 32 clips over 8 GPUs
    'edvr_l_deblur_t5_720x1280': dict(net=dict(L5, hr_in=True, with_predeblur=True),
                                      shape=(5, 3, 720, 1280), batch=4, scale=1,
                                      desc='EDVR-L deblur (hr_in, predeblur), 5 frames, 1280x720 -> 1280x720, batch 4/GPU'),
    # BASELINE.json configs[0] (plumbing-sized)
    'edvr_m_x4_t5_64x64': dict(net=dict(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2),
                               shape=(5, 3, 64, 64), batch=1, desc='EDVR-M x4, 5 frames, 64x64 LR crop, batch 1'),
}
MFMA_KERNELS = ('conv3x3_winograd_f4_kernel', 'conv3x3_winograd_kernel', 'conv3x3_winograd_wgrad_kernel', 'conv2d_mfma_kernel', 'conv2d_wgrad_kernel',
                'conv1x1_stream_kernel', 'gemm_nt_kernel', 'dcnv2_fwd', 'dcnv2_bwd')
WINOGRAD = ('conv3x3_winograd_kernel', 'conv3x3_winograd_wgrad_kernel')  # execute 16 instead of 36 multiplies per 2x2 tile
WINOGRAD_F4 = ('conv3x3_winograd_f4_kernel',)  # F(4x4,3x3): 36 instead of 144 multiplies per 4x4 tile
# the split-operand forms (fp32 operands as f16 (hi, lo) pairs on the f16 matrix pipe, 4 cross products, fp32 accumulate): the matrix
# pipe is no longer what bounds them - the table carries their algorithmic HBM rate AND their share of the f16 matrix peak
SPLIT_KERNELS = ('conv3x3_winograd_f4s_kernel', 'conv3x3_winograd_wgrad_split_kernel', 'conv3x3_wgrad_direct_split_kernel', 'dcnv2_fwd[dcn_tapwin_split_fwd_kernel]', 'gemm_nt_split_kernel',
                 'conv1x1_split_kernel')
# multiplies the algorithm saves against the direct one (F(4x4): 4, F(2x2): 2.25, the DCN GEMM: none); each remaining fp32 product = 4 f16 ones
SPLIT_SAVING = {'conv3x3_winograd_f4s_kernel': 4.0, 'conv3x3_winograd_wgrad_split_kernel': 2.25, 'conv3x3_wgrad_direct_split_kernel': 1.0, 'dcnv2_fwd[dcn_tapwin_split_fwd_kernel]': 1.0,
                'gemm_nt_split_kernel': 1.0, 'conv1x1_split_kernel': 1.0}
DTYPE = ('f32 (3x3 / stride-1 convs and their weight gradients, the tap-window DCN forward, the dW product of the DCN backward and the 1x1 convs '
         'from 320 input channels up multiply SPLIT fp32 operands - f16 (hi, lo) pairs, all four cross products - on '
         'the f16 matrix pipe with fp32 accumulation: the fp32 result to within fp32 rounding; everything else fp32 MFMA / VALU).  '
         '`fp32_mfma` beside it = the same run with those kernels on the fp32 matrix pipe')


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
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip cpu_baseline and parity (the CPU oracle legs)')
    ap.add_argument('--no-stock-baseline', action='store_true', help='skip the stock PyTorch-ROCm arm')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-fp32-leg', action='store_true', help='skip the `fp32_mfma` leg (the same step with the split-operand kernels off)')
    ap.add_argument('--no-train-leg', action='store_true', help='infer mode: skip the training leg (the `train` object)')
    ap.add_argument('--no-batch4', action='store_true', help='headline workload: skip the extra 4-clips-per-GPU measurement')
    ap.add_argument('--no-target-4k', action='store_true', help='headline workload: skip the 720p -> 4K target leg (`target_4k`)')
    ap.add_argument('--train-steps', type=int, default=20)
    ap.add_argument('--no-trained-like', action='store_true', help='headline workload: skip the trained-like-offsets leg')
    ap.add_argument('--no-configs', action='store_true', help='headline workload: skip the other BASELINE configs (`configs` object)')
    return ap.parse_args()


def build_net(cfg, device, offset_bias_sigma=0.5):
    from edvr_amd import EDVR
    from util_edvr import randomize_offsets
    torch.manual_seed(10)  # options/train/EDVR/*.yml manual_seed: 10
    return randomize_offsets(EDVR(**cfg['net']), bias_sigma=offset_bias_sigma).eval().to(device)


# ------------------------------------------------------------------------------------------------ instrumented pass
def instrumented_pass(step, steps):
    """Re-run `step` with every launch of edvr_amd.ops bracketed by events on the launch stream.
    Returns {kernel: [launches, algorithmic flops, seconds, algorithmic bytes]} summed over `steps` steps."""
    from edvr_amd import ops
    records = []

    def hook(name, flops, launch, nbytes, executed=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        records.append((name, flops, e0, e1, nbytes, _executed(name, flops) if executed is None else executed))

    ops.LAUNCH_HOOK = hook
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        ops.LAUNCH_HOOK = None
    per = {}
    for name, flops, e0, e1, nbytes, executed in records:
        d = per.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += e0.elapsed_time(e1) * 1e-3
        d[3] += nbytes
        d[4] += executed  # flops the matrix cores issue for the launch, padding included (ops.conv2d asks the C side)
    return per


def _is_mfma(name):
    return name.startswith(MFMA_KERNELS)


def _is_split(name):
    return name.startswith(SPLIT_KERNELS)


def _executed(name, flops):
    """Padding-free executed flops from the algorithmic count: what the algorithm needs on exactly-fitting tiles."""
    for k, saving in SPLIT_SAVING.items():
        if name.startswith(k):
            return flops / saving * 4.0  # f16 flops: every fp32 product the algorithm needs = four f16 cross products
    if name.startswith(WINOGRAD_F4):
        return flops / 4.0
    return flops / 2.25 if name.startswith(WINOGRAD) else flops


def kernel_table(per, steps, step_seconds):
    """Per-kernel roofline view of one step: MFMA-bound kernels in executed TF/s (fraction of 157.3), HBM-bound ones in
    algorithmic GB/s (fraction of 8 TB/s)."""
    rows = {}
    for name, rec in sorted(per.items(), key=lambda kv: -kv[1][2]):
        n, flops, secs, nbytes = rec[:4]
        executed = rec[4] if len(rec) > 4 else _executed(name, flops)
        row = {'launches_per_step': round(n / steps, 1), 'ms_per_step': round(secs / steps * 1e3, 3),
               'share_of_step': round(secs / steps / step_seconds, 4), 'avg_launch_us': round(secs / n * 1e6, 2)}
        if _is_split(name) and flops > 0:
            gbps = nbytes / secs / 1e9
            row.update(bound='hbm', hbm_gbps=round(gbps, 1), frac_of_hbm_peak=round(gbps / PEAK_HBM_GBPS, 4),
                       f16_mfma_tflops_issued=round(executed / secs / 1e12, 2), frac_of_f16_mfma_peak=round(executed / secs / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                       tflops_algorithmic=round(flops / secs / 1e12, 2),
                       equivalent_fp32_mfma_frac=round(flops / [v for k, v in SPLIT_SAVING.items() if name.startswith(k)][0] / secs / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
        elif _is_mfma(name) and flops > 0:
            ex, useful = executed / secs / 1e12, _executed(name, flops) / secs / 1e12
            # frac_of_mfma_peak counts the USEFUL matrix-core flops (no padded tiles / channels); the issued ones are beside it
            row.update(bound='mfma', tflops_executed=round(useful, 2), frac_of_mfma_peak=round(useful / PEAK_F32_MFMA_TFLOPS, 4),
                       tflops_incl_padding=round(ex, 2), frac_incl_padding=round(ex / PEAK_F32_MFMA_TFLOPS, 4))
            if name.startswith(WINOGRAD + WINOGRAD_F4):
                row['tflops_algorithmic'] = round(flops / secs / 1e12, 2)
        else:
            gbps = nbytes / secs / 1e9
            row.update(bound='hbm', hbm_gbps=round(gbps, 1), frac_of_hbm_peak=round(gbps / PEAK_HBM_GBPS, 4))
        rows[name] = row
    return rows


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (scripts/prof_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, calibrated on known-size copies as
    MI355X_MICROARCH.md's HBM section prescribes).  PMC collection serialises kernels, so it is not redone inside the timed run;
    the summary records the hash of the kernel sources it was measured on (`csrc_sha16`), compared with the tree's below."""
    for rnd in (PROFILE_ROUND, 'r5', 'r4', 'r3', 'r2', 'r1'):
        path = os.path.join(ROOT, 'profiles', rnd, f'traffic_{workload}.json')
        if os.path.exists(path):
            rep = json.load(open(path))
            base = kernel.split('<')[0]
            # every template instantiation of the kernel (e.g. the two block shapes of the F(4x4) kernel), weighted by launches
            hits = [v for k, v in rep['kernels'].items() if k.split('<')[0].endswith(base) and v.get('hbm_bytes_per_launch')]
            if hits:
                n = sum(v['launches'] for v in hits)
                agg = {f: sum(v[f] * v['launches'] for v in hits) / n
                       for f in ('fetch_bytes_per_launch', 'write_bytes_per_launch', 'hbm_bytes_per_launch')}
                agg['launches'] = n
                agg['csrc_sha16'] = rep.get('csrc_sha16')
                return agg, os.path.relpath(path, ROOT)
    return None, None


def split_roofline_object(per, name, steps, step_seconds, workload, default_batch):
    """The dominant kernel is a split-operand one: its ceilings are HBM (algorithmic bytes) and the f16 matrix pipe, and it sits
    far below both - what bounds it is the rate at which ONE CU takes in operands from L2 (profiles/r5/README.md)."""
    n, flops, secs, nbytes = per[name][:4]
    executed = per[name][4]
    gbps = nbytes / secs / 1e9
    tr, tr_src = measured_traffic(workload, name) if default_batch else (None, None)
    from edvr_amd.build import source_hash
    return {
        'bound': 'hbm', 'kernel': name, 'achieved': round(gbps, 1), 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': round(gbps / PEAK_HBM_GBPS, 4),
        'definition': 'achieved = ALGORITHMIC bytes of the launches (every input / residual / output element once + the weights, SURVEY 8(d)) / '
                      'HIP-event time of the kernel; frac = achieved / 8 TB/s',
        'f16_mfma': {'tflops_issued': round(executed / secs / 1e12, 2), 'peak': PEAK_F16_MFMA_TFLOPS, 'frac': round(executed / secs / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                     'what': 'v_mfma_f32_32x32x16_f16 issues x 32768 flops (padding included): four f16 cross products per fp32 product of '
                             'F(4x4) - the fp32 kernel needs 4x this pipe time on a pipe 16x slower'},
        'equivalent_fp32_mfma_frac': round(flops / [v for k, v in SPLIT_SAVING.items() if name.startswith(k)][0] / secs / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        'limiter': 'neither roofline: per 8-channel chunk a CU takes in 74 KB of weights (no reuse inside a CU: 36 accumulators per output '
                   'fill the register file at 64 channels x 32 tiles) + 25 KB of input rows from L2; measured 22-25 B/clk/CU with the texture '
                   'addresser busy 63 % of the time and the staging / multiplying waves stalled on it (profiles/r5)',
        'algorithm': 'winograd F(4x4,3x3); fp32 operands as f16 (hi, lo) pairs, 4 cross products on v_mfma_f32_32x32x16_f16, fp32 accumulate',
        'algorithmic_tflops': round(flops / secs / 1e12, 2),
        'launches_per_step': round(n / steps, 1), 'avg_launch_us': round(secs / n * 1e6, 2),
        'gflop_per_launch_algorithmic': round(flops / n / 1e9, 3), 'share_of_step': round(secs / steps / step_seconds, 4),
        'algorithmic_bytes_per_launch': round(nbytes / n),
        'traffic': round(tr['hbm_bytes_per_launch']) if tr else None,
        'traffic_detail': ({'unit': 'bytes per launch (average over the launches of this kernel in one step)',
                            'fetch': round(tr['fetch_bytes_per_launch']), 'write': round(tr['write_bytes_per_launch']),
                            'algorithmic': round(nbytes / n), 'source': tr_src, 'measured_on_csrc_sha16': tr.get('csrc_sha16'),
                            'csrc_sha16': source_hash(), 'stale': tr.get('csrc_sha16') != source_hash()} if tr else None),
    }


def roofline_object(per, steps, step_seconds, workload, default_batch):
    mf = {k: v for k, v in per.items() if (_is_mfma(k) or _is_split(k)) and v[1] > 0}
    name = max(mf, key=lambda k: mf[k][2])
    if _is_split(name):
        return split_roofline_object(per, name, steps, step_seconds, workload, default_batch)
    n, flops, secs, nbytes = per[name][:4]
    executed = per[name][4] if len(per[name]) > 4 else _executed(name, flops)
    ex = executed / secs / 1e12
    useful = _executed(name, flops) / secs / 1e12
    tr, tr_src = measured_traffic(workload, name) if default_batch else (None, None)
    wino = name.startswith(WINOGRAD)
    f4 = name.startswith(WINOGRAD_F4)
    from edvr_amd.build import source_hash
    return {
        'bound': 'mfma', 'kernel': name, 'achieved': round(useful, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(useful / PEAK_F32_MFMA_TFLOPS, 4),
        'definition': 'achieved = USEFUL flops the fp32 matrix cores execute (algorithmic flops / 4 for F(4x4), / 2.25 for F(2x2): '
                      'what the algorithm needs on exactly fitting tiles) / HIP-event time of the kernel; frac = achieved / peak.  '
                      'The MFMAs issued on padded tiles and channels are NOT counted as achievement: incl_padding_* carries them '
                      '(v_mfma_f32_32x32x2_f32 issues x 4096 = edvr_conv2d_executed_flops = what rocprofv3 '
                      'SQ_INSTS_VALU_MFMA_MOPS_F32 counts, profiles/*/winograd_f4_micro_pmc.json)',
        'executed_without_padding_tflops': round(useful, 2),
        'incl_padding_tflops': round(ex, 2), 'incl_padding_frac': round(ex / PEAK_F32_MFMA_TFLOPS, 4),
        'padding_overhead': round(executed / max(_executed(name, flops), 1.0) - 1.0, 4),
        # SURVEY 8(d)'s algorithmic count against the matrix-core peak scaled by the algorithm's multiply saving (4 for F(4x4)):
        'algorithmic_frac_winograd': round(flops / secs / 1e12 / ((4.0 if f4 else (2.25 if wino else 1.0)) * PEAK_F32_MFMA_TFLOPS), 4),
        'algorithm': 'winograd F(4x4,3x3), fp32: 36 instead of 144 multiplies per 4x4 tile and channel pair' if f4
                     else ('winograd F(2x2,3x3), fp32: 16 instead of 36 multiplies per 2x2 tile and channel pair' if wino
                           else 'direct implicit GEMM, fp32'),
        # SURVEY 8(d) counts ALGORITHMIC flops (2*9*Ci*Co per output pixel); Winograd executes 1/2.25 of them, so this figure can
        # exceed the MFMA peak - it is the direct-algorithm-equivalent rate, not a roofline fraction:
        'algorithmic_tflops': round(flops / secs / 1e12, 2),
        'algorithmic_over_peak': round(flops / secs / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        'launches_per_step': round(n / steps, 1), 'avg_launch_us': round(secs / n * 1e6, 2),
        'gflop_per_launch_algorithmic': round(flops / n / 1e9, 3), 'share_of_step': round(secs / steps / step_seconds, 4),
        'algorithmic_bytes_per_launch': round(nbytes / n),
        'traffic': round(tr['hbm_bytes_per_launch']) if tr else None,
        'traffic_detail': ({'unit': 'bytes per launch (average over the launches of this kernel in one step)',
                            'fetch': round(tr['fetch_bytes_per_launch']), 'write': round(tr['write_bytes_per_launch']),
                            'algorithmic': round(nbytes / n), 'source': tr_src, 'measured_on_csrc_sha16': tr.get('csrc_sha16'),
                            'csrc_sha16': source_hash(),
                            # the committed PMC summary was taken on other kernel sources than the ones in the tree
                            'stale': tr.get('csrc_sha16') != source_hash()} if tr else None),
    }


# ------------------------------------------------------------------------------------------------ baselines / parity
def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def parity_inputs(cfg):
    """One clip + weights + a synthetic ground truth, all from CPU generators (lq seed 0, gt seed 1)."""
    from edvr_amd import EDVR
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(**cfg['net'])).eval()
    x = torch.rand(1, *cfg['shape'], generator=torch.Generator().manual_seed(0))
    sc = cfg.get('scale', 4)
    gt = torch.rand(1, 3, sc * cfg['shape'][2], sc * cfg['shape'][3], generator=torch.Generator().manual_seed(1))
    return net, x, gt


def oracle_kw(cfg):
    n = cfg['net']
    return dict(center=n.get('center_frame_idx'), hr_in=n.get('hr_in', False), with_predeblur=n.get('with_predeblur', False))


def cpu_baseline_and_parity(cfg, device, repeats=2):
    """Oracle forward of ONE clip of the workload on the host cores (median of `repeats`), and the SAME clip through the HIP
    path: max relative error and PSNR difference against a synthetic ground truth (north_star: within 1e-3 dB)."""
    from oracle import dcn_oracle, edvr_oracle as EO
    net, x, gt = parity_inputs(cfg)
    sd = net.state_dict()
    times, ref = [], None
    with torch.no_grad():
        for _ in range(repeats):
            t0 = time.time()
            ref = EO.edvr_forward(sd, x, dcn=dcn_oracle.dcnv2_c, **oracle_kw(cfg))
            times.append(time.time() - t0)
        ours = net.to(device)(x.to(device)).cpu()
    dt = statistics.median(times)
    base = dict(value=round(1.0 / dt, 5), unit='clips/s', cores=torch.get_num_threads(), kind='port',
                sample=f'1 clip (one forward, batch 1) of the same workload, median of {repeats} runs: reference network restated '
                       'in torch CPU ops (fp32, oneDNN) + C/OpenMP DCNv2 oracle; the reference itself has no CPU DCN path',
                seconds=round(dt, 2), all_seconds=[round(t, 2) for t in times], host_cpus=os.cpu_count(), cpu_model=cpu_model())
    p_ours, p_ref = EO.psnr(ours, gt), EO.psnr(ref, gt)
    par = dict(clip='lq seed 0 / gt seed 1 (uniform), weights manual_seed 10 + randomised conv_offset',
               against='CPU oracle (fp32 torch ops + C DCNv2), same clip, same weights',
               max_rel_err=float(((ours - ref).abs().max() / ref.abs().max()).item()),
               psnr_ours=round(p_ours, 6), psnr_oracle=round(p_ref, 6), d_psnr=round(abs(p_ours - p_ref), 8),
               tolerance={'max_rel_err': 2e-4, 'd_psnr_db': 1e-3},
               ok=bool(((ours - ref).abs().max() / ref.abs().max()).item() < 2e-4 and abs(p_ours - p_ref) <= 1e-3))
    return base, par, ours


def stock_rocm_baseline(cfg, device, ours_cpu, repeats=3):
    """SURVEY 8(d) second arm: the reference network in stock PyTorch-ROCm ops (F.conv2d -> MIOpen, einsum -> rocBLAS) with the
    pure-torch floor/gather DCNv2 (oracle/dcn_oracle.py::dcnv2_torch) on THIS GPU, one clip.  Baseline + parity witness only."""
    from oracle import dcn_oracle, edvr_oracle as EO
    net, x, _ = parity_inputs(cfg)
    sd = {k: v.to(device) for k, v in net.state_dict().items()}
    xd = x.to(device)
    if ours_cpu is None:  # (the CPU legs were skipped: this arm is then the parity witness of the workload)
        with torch.no_grad():
            ours_cpu = net.to(device)(xd).cpu()
        net = None
        torch.cuda.empty_cache()
    times = []
    with torch.no_grad():
        ref = EO.edvr_forward(sd, xd, dcn=dcn_oracle.dcnv2_torch, **oracle_kw(cfg))  # warm-up (MIOpen find)
        torch.cuda.synchronize()
        for _ in range(repeats):
            t0 = time.perf_counter()
            ref = EO.edvr_forward(sd, xd, dcn=dcn_oracle.dcnv2_torch, **oracle_kw(cfg))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    ref = ref.cpu()
    out = dict(value=round(1.0 / dt, 3), unit='clips/s', ms_per_clip=round(dt * 1e3, 2),
               what='same network in stock torch ops (MIOpen / rocBLAS, fp32, cudnn.benchmark off) + pure-torch DCNv2, batch 1, '
                    f'median of {repeats} after one warm-up, on this GPU')
    out['max_rel_err_vs_ours'] = float(((ours_cpu - ref).abs().max() / ref.abs().max()).item())
    return out


# ------------------------------------------------------------------------------------------------ steps
def make_train_step(net, cfg, batch, device, rank, optimizer, x=None, lr=4e-4):
    from edvr_amd import dist as D
    from edvr_amd.autograd import charbonnier_loss
    from edvr_amd.optim import FusedAdam
    net.train()
    sc = cfg.get('scale', 4)
    if x is None:
        x = torch.rand(batch, *cfg['shape'], generator=torch.Generator().manual_seed(rank)).to(device)
    gt = torch.rand(batch, 3, sc * cfg['shape'][2], sc * cfg['shape'][3], generator=torch.Generator().manual_seed(1000 + rank)).to(device)
    model = D.wrap_ddp(net)  # RCCL gradient all-reduce, bucketed and overlapped with backward
    dcn = [p for n, p in net.named_parameters() if 'dcn' in n]  # edvr_model.py:21-53 (two groups as with dcn_lr_mul != 1)
    rest = [p for n, p in net.named_parameters() if 'dcn' not in n]
    groups = [{'params': rest, 'lr': lr}, {'params': dcn, 'lr': lr * 1}]
    opt = (torch.optim.Adam if optimizer == 'torch' else FusedAdam)(groups, lr=lr, betas=(0.9, 0.99))

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        loss = charbonnier_loss(out, gt)
        loss.backward()
        opt.step()
        return loss.detach()
    return step


def timed(step, steps, warmup, dist, device):
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; returns the MAX over ranks."""
    sync = torch.cuda.synchronize if device.type == 'cuda' else (lambda: None)  # (cpu: the world-size-2 gloo test of this function)
    for _ in range(warmup):
        step()
    sync()
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    if dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()
    return elapsed


def _rel_err(a, ref):
    return float(((a.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item())


def train_parity(cfg, device, clips=2, offset_bias_sigma=0.5):
    """Witness of the training half of the metric (sr_model.py:88-112): `clips` clips, same weights, ONE forward + Charbonnier(sum)
    + backward through this path and through oracle/edvr_oracle.py in stock fp32 torch ops (convs = F.unfold + GEMM, pure-torch
    DCNv2, torch autograd) on this GPU.  Nothing is shared between the two runs (activation sides, pooling routes, DCN cells are each
    run's own), so single tensors may differ by a finite jump where a value sits within rounding distance of a kink
    (tests/test_gpu_train.py): bounded are the loss, the output, the MEDIAN over the parameter tensors and the maximum."""
    from edvr_amd.autograd import charbonnier_loss
    from oracle import dcn_oracle, edvr_oracle as EO
    net = build_net(cfg, device, offset_bias_sigma=offset_bias_sigma).train()
    sc = cfg.get('scale', 4)
    x = torch.rand(clips, *cfg['shape'], generator=torch.Generator().manual_seed(0)).to(device)
    gt = torch.rand(clips, 3, sc * cfg['shape'][2], sc * cfg['shape'][3], generator=torch.Generator().manual_seed(1)).to(device)
    with torch.no_grad():
        net(x)  # (settles the per-layer offset statistics the backward's dX strategy is picked from)
    from edvr_amd import ops as _ops
    names = []
    _ops.LAUNCH_HOOK = lambda name, flops, launch, nbytes, executed=None: (names.append(name), launch())
    try:
        out = net(x)
        loss = charbonnier_loss(out, gt)
        loss.backward()
    finally:
        _ops.LAUNCH_HOOK = None
    ours = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    t0 = time.perf_counter()
    ref_out = EO.edvr_forward(sd, x, dcn=dcn_oracle.dcnv2_torch, conv_impl='unfold', **oracle_kw(cfg))
    ref_loss = EO.charbonnier_sum(ref_out, gt)
    ref_loss.backward()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    errs = sorted((_rel_err(ours[k], sd[k].grad), k) for k in ours if sd[k].grad is not None and sd[k].grad.abs().max() > 0)
    vals = [e for e, _ in errs]
    par = dict(against='oracle/edvr_oracle.py in stock PyTorch-ROCm fp32 ops (F.unfold + rocBLAS GEMM convs, pure-torch DCNv2, torch autograd) on '
                       f'this GPU: {clips} clips of the training shape, same weights, one forward + Charbonnier(sum) + backward, nothing shared',
               loss_ours=float(loss.item()), loss_stock=float(ref_loss.item()),
               loss_rel_err=abs(float(loss.item()) - float(ref_loss.item())) / abs(float(ref_loss.item())),
               output_max_rel_err=_rel_err(out.detach(), ref_out.detach()),
               grad_tensors=len(vals), grad_rel_err_median=vals[len(vals) // 2], grad_rel_err_p90=vals[int(len(vals) * 0.9)],
               grad_rel_err_max=vals[-1], worst_tensor=errs[-1][1],
               tolerance={'loss_rel_err': 1e-5, 'output_max_rel_err': 2e-4, 'grad_rel_err_median': 1e-3, 'grad_rel_err_p90': 5e-3, 'grad_rel_err_max': 2e-2},
               dcn_kernels=sorted({k for k in names if k.startswith('dcnv2')}), conv_kernels=sorted({k.split('<')[0] for k in names if k.startswith('conv3x3')}),
               witness_seconds=round(dt, 2))
    par['ok'] = bool(par['loss_rel_err'] < 1e-5 and par['output_max_rel_err'] < 2e-4 and par['grad_rel_err_median'] < 1e-3 and
                     par['grad_rel_err_p90'] < 5e-3 and par['grad_rel_err_max'] < 2e-2)
    return par


def trained_like_leg(cfg, batch, args, device, rank, world, dist, sigmas=(4.0, 10.0)):
    """The headline workload with the offsets of a trained model.  bias_sigma_S: conv_offset.bias ~ N(0, S^2) per channel on white-noise
    frames (every tap its own multi-pixel displacement, constant in space).  motion: structured clips (smooth drifting background, moving
    rectangles, a little texture: tests/util_edvr.py motion_frames) and offset convs rescaled until the offsets vary IN SPACE like a
    trained model's - per-tap displacements of sigma 3 px plus a field that is smooth inside objects and jumps at their edges, mean
    |horizontal neighbour difference| ~0.5 px per DCN layer (arch_util.py:243-257)."""
    from util_edvr import motion_frames, motion_like_offsets
    out = {}
    for sigma in tuple(sigmas) + ('motion',):
        if sigma == 'motion':
            net = build_net(cfg, device, offset_bias_sigma=3.0)
            x = motion_frames(batch, cfg['shape'], seed=rank).to(device)
            motion_like_offsets(net, x, target_rough=0.5, bias_sigma=3.0)
        else:
            net = build_net(cfg, device, offset_bias_sigma=sigma)
            x = torch.rand(batch, *cfg['shape'], generator=torch.Generator().manual_seed(rank)).to(device)

        def step():
            with torch.no_grad():
                return net(x)
        steps = 5
        elapsed = timed(step, steps, 3, dist, device)  # (the first forwards also settle the per-layer kernel hints)
        rec = None
        if rank == 0:
            net.check_offsets()
            dcns = net.pcd_align.dcn_modules()
            rough = [m.last_offset_rough for m in dcns]
            rec = {'value': round(batch * world * steps / elapsed, 4), 'unit': 'clips/s', 'ms_per_step': round(elapsed / steps * 1e3, 3), 'steps': steps,
                   'clips_per_gpu': batch, 'offset_bias_sigma': 3.0 if sigma == 'motion' else sigma,
                   'frames': 'structured (tests/util_edvr.py motion_frames)' if sigma == 'motion' else 'torch.rand',
                   'mean_abs_offset_px': [round(m.last_offset_absmean, 3) for m in dcns],
                   'offset_roughness_px': [None if r is None else round(r, 3) for r in rough],
                   'offset_roughness_px_mean': round(sum(r or 0.0 for r in rough) / len(rough), 3)}
            if not args.no_roofline:
                per = instrumented_pass(step, 1)
                tab = kernel_table(per, 1, elapsed / steps)
                for k, v in tab.items():
                    if k.startswith('dcnv2_fwd'):
                        rec[k] = v
                rec['dcn_kernel_classes'] = sorted(k for k in tab if k.startswith('dcnv2'))
            if world == 1 and not args.no_stock_baseline:
                rec['parity'] = one_clip_parity(net, cfg, x, device)
            out['motion' if sigma == 'motion' else f'bias_sigma_{sigma:g}'] = rec
        del net, x, step
        torch.cuda.empty_cache()
    return out if rank == 0 else None


def one_clip_parity(net, cfg, x, device):
    """One clip of the leg's own input through this path and through oracle/edvr_oracle.py in stock PyTorch-ROCm fp32 ops, same weights."""
    from oracle import dcn_oracle, edvr_oracle as EO
    try:
        x1 = x[:1].contiguous()
        sc = cfg.get('scale', 4)
        gt = torch.rand(1, 3, sc * cfg['shape'][2], sc * cfg['shape'][3], generator=torch.Generator().manual_seed(1)).to(device)
        with torch.no_grad():
            ours = net(x1)
            ref = EO.edvr_forward(net.state_dict(), x1, dcn=dcn_oracle.dcnv2_torch, conv_impl='unfold', **oracle_kw(cfg))
        err = _rel_err(ours, ref)
        p_ours, p_ref = EO.psnr(ours, gt), EO.psnr(ref, gt)
        return dict(against='stock PyTorch-ROCm fp32 ops (F.unfold + GEMM convs, pure-torch DCNv2) on this GPU, one clip, same weights',
                    max_rel_err=err, d_psnr=round(abs(p_ours - p_ref), 8), tolerance={'max_rel_err': 2e-4, 'd_psnr_db': 1e-3},
                    ok=bool(err < 2e-4 and abs(p_ours - p_ref) <= 1e-3))
    except Exception as e:  # (a witness arm must never take the measurement down: e.g. the stock arm's memory at 720p)
        return {'error': f'{type(e).__name__}: {str(e)[:200]}'}


def configs_leg(args, device, rank, world, dist):
    """The BASELINE.json configs the headline does not cover, one line each, timed like the headline (barrier + synchronize, max
    over ranks); full-size parity of each: tests/test_gpu_fullsize_parity.py."""
    out = {}
    plan = [('configs[1] EDVR-M x4 T5 180x320 batch 4, inference', 'edvr_m_x4_t5_180x320', 'infer', 5),
            ('configs[2] EDVR-L x4 T7 180x320 batch 8, forward', 'edvr_l_x4_t7_180x320', 'infer', 3),
            ('configs[2] EDVR-L x4 T7 180x320 batch 8, forward + backward + Adam', 'edvr_l_x4_t7_180x320', 'train', 3),
            ('configs[4] EDVR-L deblur T5 1280x720 batch 4, inference', 'edvr_l_deblur_t5_720x1280', 'infer', 3)]
    for label, wl, mode, steps in plan:
        cfg = WORKLOADS[wl]
        try:
            net = build_net(cfg, device)
            if mode == 'train':
                step = make_train_step(net, cfg, cfg['batch'], device, rank, 'fused')
            else:
                x = torch.rand(cfg['batch'], *cfg['shape'], generator=torch.Generator().manual_seed(rank)).to(device)

                def step(net=net, x=x):
                    with torch.no_grad():
                        return net(x)
            elapsed = timed(step, steps, 2, dist, device)
            if rank == 0:
                out[label] = {'clips_per_sec': round(cfg['batch'] * world * steps / elapsed, 3), 'ms_per_step': round(elapsed / steps * 1e3, 2),
                              'steps': steps, 'clips_per_gpu': cfg['batch'], 'workload': wl}
                if mode == 'infer' and not args.no_roofline:
                    out[label]['kernels'] = {k: v for k, v in list(kernel_table(instrumented_pass(step, 1), 1, elapsed / steps).items())[:6]}
                if mode == 'infer' and world == 1 and not args.no_stock_baseline:
                    out[label]['parity'] = one_clip_parity(net, cfg, x, device)
        except Exception as e:  # e.g. out of memory on a smaller part: the headline line must survive
            if rank == 0:
                out[label] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
        net = step = x = None
        torch.cuda.empty_cache()
    return out if rank == 0 else None


def train_leg(args, device, rank, world, dist):
    """BASELINE.json's second headline: EDVR-L training iterations/sec on the cfg4 per-GPU shape (32 clips of 5 x 64x64 per GPU)."""
    cfg = WORKLOADS['edvr_l_train_t5_64x64']
    net = build_net(cfg, device)
    step = make_train_step(net, cfg, cfg['batch'], device, rank, 'fused')
    elapsed = timed(step, args.train_steps, 2, dist, device)
    out = {'workload': cfg['desc'], 'iters_per_sec': round(args.train_steps / elapsed, 4), 'ms_per_iter': round(elapsed / args.train_steps * 1e3, 2),
           'clips_per_sec': round(cfg['batch'] * world * args.train_steps / elapsed, 2), 'steps': args.train_steps, 'warmup': 2,
           'n_gpus': world, 'global_batch': cfg['batch'] * world,
           'optimizer': 'edvr_amd.optim.FusedAdam', 'what': 'forward + Charbonnier(sum) + backward + DDP all-reduce + Adam step'}
    if rank == 0 and not args.no_roofline and world == 1:  # (an extra DDP step on one rank alone would wait for its peers)
        per = instrumented_pass(step, 1)
        tab = kernel_table(per, 1, elapsed / args.train_steps)
        out['dominant_kernels'] = {k: v for k, v in list(tab.items())[:8]}
        for key, label in (('conv3x3_winograd_f4_kernel', 'fwd_dgrad_f4'), ('conv3x3_winograd_kernel', 'fwd_dgrad_f2'), ('conv3x3_winograd_wgrad_kernel', 'wgrad')):
            if key in tab:
                out[f'{label}_mfma_frac'] = tab[key]['frac_of_mfma_peak']
    if not args.no_fp32_leg:
        from edvr_amd import ops as _ops
        prev = _ops.set_f4s(False, False)
        try:
            e32 = timed(step, 6, 2, dist, device)  # all ranks
        finally:
            _ops.set_f4s(*prev)
        out['fp32_mfma'] = {'iters_per_sec': round(6 / e32, 4), 'ms_per_iter': round(e32 / 6 * 1e3, 2), 'steps': 6,
                            'what': 'the same step with EVERY split-operand kernel off (ops.set_f4s(False, False) = EDVR_WINOGRAD_F4S=0: convs, weight gradients, DCN forward, dW products, 1x1 convs): fp32 matrix pipe'}
    if not args.no_trained_like:
        # the same training step with the offsets of a TRAINED model (conv_offset.bias ~ N(0, 4^2): every tap its own multi-pixel
        # displacement): the forward's cost does not change (tap-window kernel), the backward's dX leaves the no-scatter kernels
        del step, net
        torch.cuda.empty_cache()
        net = build_net(cfg, device, offset_bias_sigma=4.0)
        step = make_train_step(net, cfg, cfg['batch'], device, rank, 'fused')
        tsteps = 10
        e2 = timed(step, tsteps, 3, dist, device)  # all ranks (DDP collectives)
        tl = {'offset_bias_sigma': 4.0, 'iters_per_sec': round(tsteps / e2, 4), 'ms_per_iter': round(e2 / tsteps * 1e3, 2), 'steps': tsteps,
              'vs_sub_pixel_offsets': round((tsteps / e2) / (args.train_steps / elapsed), 4)}
        if rank == 0 and not args.no_roofline and world == 1:
            tab = kernel_table(instrumented_pass(step, 1), 1, e2 / tsteps)
            tl['dcn_kernels'] = {k: v for k, v in tab.items() if k.startswith('dcnv2')}
            tl['mean_abs_offset_px'] = [round(m.last_offset_absmean, 3) for m in net.pcd_align.dcn_modules()]
        if rank == 0 and world == 1 and not args.no_stock_baseline:
            try:  # its own witness: the LDS-window dX strategy is a different kernel set from the sub-pixel step's
                tl['parity'] = train_parity(cfg, device, offset_bias_sigma=4.0)
            except Exception as e:
                tl['parity'] = {'error': f'{type(e).__name__}: {str(e)[:300]}'}
        out['trained_like'] = tl
        # ... and with offsets that vary in SPACE as a trained model's do (structured crops, offset convs rescaled to ~0.5 px of
        # neighbour difference per DCN layer: trained_like_leg's `motion`), which is what decides the backward's dX strategy per layer
        del step, net
        torch.cuda.empty_cache()
        from util_edvr import motion_frames, motion_like_offsets
        net = build_net(cfg, device, offset_bias_sigma=3.0)
        xm = motion_frames(cfg['batch'], cfg['shape'], seed=rank).to(device)
        motion_like_offsets(net, xm, target_rough=0.5, bias_sigma=3.0)
        # (lr 1e-6: Adam's first steps move every weight by +-lr whatever the gradient - at 4e-4 the rescaled offset convs would drift by
        # pixels per iteration and the field measured would not be the field set up)
        step = make_train_step(net, cfg, cfg['batch'], device, rank, 'fused', x=xm, lr=1e-6)
        e3 = timed(step, tsteps, 3, dist, device)  # all ranks
        mo = {'offset_bias_sigma': 3.0, 'frames': 'structured (tests/util_edvr.py motion_frames)', 'iters_per_sec': round(tsteps / e3, 4),
              'ms_per_iter': round(e3 / tsteps * 1e3, 2), 'steps': tsteps, 'vs_sub_pixel_offsets': round((tsteps / e3) / (args.train_steps / elapsed), 4)}
        if rank == 0 and not args.no_roofline and world == 1:
            tab = kernel_table(instrumented_pass(step, 1), 1, e3 / tsteps)
            mo['dcn_kernels'] = {k: v for k, v in tab.items() if k.startswith('dcnv2')}
            dcns = net.pcd_align.dcn_modules()
            mo['mean_abs_offset_px'] = [round(m.last_offset_absmean, 3) for m in dcns]
            mo['offset_roughness_px'] = [None if m.last_offset_rough is None else round(m.last_offset_rough, 3) for m in dcns]
        out['motion'] = mo
    if rank == 0 and world == 1 and not args.no_stock_baseline:
        del step, net
        step = None
        torch.cuda.empty_cache()
        try:
            out['parity'] = train_parity(cfg, device)
        except Exception as e:  # (a witness arm must never take the measurement down)
            out['parity'] = {'error': f'{type(e).__name__}: {str(e)[:300]}'}
    return out, step


def target_4k_leg(net, args, device, rank, world, dist):
    """BASELINE.json north_star "Target": x4 720p -> 4K, 5 frames, EDVR-L - one clip per GPU (67.7 algorithmic TFLOP), the
    network (and weights) of the headline workload.  Timed like the headline: barrier + synchronize on both sides, max over ranks."""
    cfg = WORKLOADS['edvr_l_x4_t5_720x1280']
    x = torch.rand(1, *cfg['shape'], generator=torch.Generator().manual_seed(rank)).to(device)

    def step():
        with torch.no_grad():
            return net(x)
    steps = 5
    elapsed = timed(step, steps, 2, dist, device)
    out = None
    if rank == 0:
        out = {'workload': cfg['desc'], 'value': round(world * steps / elapsed, 4), 'unit': 'clips/s', 'ms_per_clip': round(elapsed / steps * 1e3, 2),
               'steps': steps, 'warmup': 2, 'n_gpus': world, 'clips_per_gpu': 1,
               'output_megapixels_per_sec': round(world * steps / elapsed * 2880 * 5120 / 1e6, 1)}
        if not args.no_roofline:
            per = instrumented_pass(step, 1)
            tab = kernel_table(per, 1, elapsed / steps)
            out['kernels'] = {k: v for k, v in list(tab.items())[:6]}
            f4 = next((v for k, v in tab.items() if k.startswith(WINOGRAD_F4)), None)
            if f4:
                out['f4_frac_of_mfma_peak'] = f4['frac_of_mfma_peak']
            out['fallbacks'] = {'conv1x1_stream_kernel_used': any(k.startswith('conv1x1_stream_kernel') for k in tab)}
    if rank == 0 and world == 1 and not args.no_stock_baseline:
        # parity witness at this size: the stock-PyTorch-ROCm arm on the same clip (the CPU oracle would need ~4 minutes)
        from oracle import dcn_oracle, edvr_oracle as EO
        try:
            gt = torch.rand(1, 3, 2880, 5120, generator=torch.Generator().manual_seed(1)).to(device)
            with torch.no_grad():
                ours = net(x)
                sd = net.state_dict()
                # convs as F.unfold + rocBLAS GEMM: on a fresh box MIOpen compiles a kernel per new conv shape (~4 minutes for
                # this workload's ~20 shapes), im2col + GEMM needs none; the MIOpen-based arm is timed on the headline workload
                t0 = time.perf_counter()
                ref = EO.edvr_forward(sd, x, dcn=dcn_oracle.dcnv2_torch, conv_impl='unfold', **oracle_kw(cfg))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            err = float(((ours - ref).abs().max() / ref.abs().max()).item())
            p_ours, p_ref = EO.psnr(ours, gt), EO.psnr(ref, gt)
            out['parity'] = dict(against='oracle/edvr_oracle.py in stock PyTorch-ROCm fp32 ops (convs = F.unfold + rocBLAS GEMM, pure-torch DCNv2) on this GPU, same clip, '
                                         'same weights; intermediates at this size: tests/test_gpu_fullsize_parity.py',
                                 max_rel_err=err, psnr_ours=round(p_ours, 6), psnr_stock=round(p_ref, 6), d_psnr=round(abs(p_ours - p_ref), 8),
                                 tolerance={'max_rel_err': 2e-4, 'd_psnr_db': 1e-3}, ok=bool(err < 2e-4 and abs(p_ours - p_ref) <= 1e-3))
            out['parity']['witness_seconds'] = round(dt, 2)
        except Exception as e:  # (a baseline arm must never take the measurement down)
            out['parity'] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
    return out


def spawn_command(gpus, port, argv):
    """The launch `python bench.py --gpus N` performs outside a torchrun environment: N ranks of this script on this node, one
    per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve), the script's own arguments passed through."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def spawn_env(environ):
    return dict(environ, HSA_ENABLE_IPC_MODE_LEGACY='0')  # the host driver only supports dmabuf IPC (RCCL / tensor sharing across processes)


def self_spawn(args):
    """`python bench.py --gpus N` outside torchrun: run N ranks of this script on this node (RCCL over xGMI)."""
    n = torch.cuda.device_count()
    assert n >= args.gpus, f'--gpus {args.gpus} but only {n} GPU(s) visible'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    sys.exit(subprocess.call(spawn_command(args.gpus, port, sys.argv[1:]), env=spawn_env(os.environ)))


DTYPE_SHORT = 'fp32 emulated as 2xf16 (hi+lo, 22-bit operands, 4 cross products), fp32 accumulate'
FULL_REPORT = 'bench_full.json'
COMPACT_LIMIT = 6000  # bytes: the driver reads ONE line; round 5's 24 KB line did not parse


def _num(v, nd=4):
    """A finite float rounded for the compact line; None for anything json.loads could choke on (nan / inf) or that is missing."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, (int, float)):
        f = float(v)
        if f != f or f in (float('inf'), float('-inf')):
            return None
        return v if isinstance(v, int) else (round(f, nd) if abs(f) >= 1e-3 or f == 0 else float(f'{f:.3e}'))
    return v


def _pick(src, *keys):
    return {k: _num(src[k]) for k in keys if isinstance(src, dict) and k in src}


def _sanitize(o):
    """The full report with every non-finite float replaced by None (strict JSON)."""
    if isinstance(o, dict):
        return {str(k): _sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sanitize(v) for v in o]
    if isinstance(o, float) and (o != o or o in (float('inf'), float('-inf'))):
        return None
    return o


def compact_line(full):
    """The ONE line the driver parses: the contract's fields + `roofline` and `cpu_baseline` objects in numbers only, one short
    record per optional leg.  Kernel tables, definitions, per-config records and witnesses stay in FULL_REPORT."""
    c = {k: _num(full.get(k)) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                        'scaling', 'vs_baseline')}
    c['dtype'] = DTYPE_SHORT if full.get('dtype') == DTYPE else str(full.get('dtype'))[:100]
    c['data'] = 'synthetic'
    cfg = full.get('config', {})
    c['config'] = {'workload': str(cfg.get('workload', ''))[:120], 'clips_per_gpu': cfg.get('clips_per_gpu'),
                   'global_clips': cfg.get('global_clips'), 'parallelism': str(cfg.get('parallelism', ''))[:60],
                   'world_size': cfg.get('world_size'), 'backend': str(cfg.get('backend', ''))[:40]}
    if 'iters_per_sec' in full:
        c['iters_per_sec'] = _num(full['iters_per_sec'])
    r = full.get('roofline')
    if isinstance(r, dict):
        c['roofline'] = _pick(r, 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'launches_per_step', 'avg_launch_us',
                              'share_of_step', 'algorithmic_bytes_per_launch', 'algorithmic_tflops')
        c['roofline'].setdefault('traffic', None)
        if isinstance(r.get('f16_mfma'), dict):
            c['roofline']['f16_mfma_frac'] = _num(r['f16_mfma'].get('frac'))
        if isinstance(r.get('traffic_detail'), dict):
            c['roofline']['traffic_stale'] = r['traffic_detail'].get('stale')
    b = full.get('cpu_baseline')
    if isinstance(b, dict):
        c['cpu_baseline'] = _pick(b, 'value', 'unit', 'cores', 'kind', 'seconds')
        c['cpu_baseline']['sample'] = str(b.get('sample', ''))[:110]
    if isinstance(full.get('parity'), dict):
        c['parity'] = _pick(full['parity'], 'max_rel_err', 'd_psnr', 'ok', 'error')
        c['parity']['against'] = 'CPU oracle, same clip and weights'
    if isinstance(full.get('fp32_mfma'), dict):
        c['fp32_mfma'] = _pick(full['fp32_mfma'], 'value', 'ms_per_step', 'iters_per_sec')
    if isinstance(full.get('stock_rocm_baseline'), dict):
        c['stock_rocm'] = _pick(full['stock_rocm_baseline'], 'value', 'max_rel_err_vs_ours', 'error')
    t = full.get('train')
    if isinstance(t, dict):
        c['train'] = _pick(t, 'iters_per_sec', 'ms_per_iter', 'clips_per_sec', 'global_batch', 'steps')
        if isinstance(t.get('fp32_mfma'), dict):
            c['train']['fp32_mfma'] = _pick(t['fp32_mfma'], 'iters_per_sec')
        if isinstance(t.get('parity'), dict):
            c['train']['parity'] = _pick(t['parity'], 'ok', 'grad_rel_err_max', 'loss_rel_err', 'error')
        for leg in ('trained_like', 'motion'):
            if isinstance(t.get(leg), dict):
                c['train'][leg] = _pick(t[leg], 'iters_per_sec', 'vs_sub_pixel_offsets')
                if isinstance(t[leg].get('parity'), dict):
                    c['train'][leg]['parity_ok'] = t[leg]['parity'].get('ok')
    t4 = full.get('target_4k')
    if isinstance(t4, dict):
        c['target_4k'] = _pick(t4, 'value', 'ms_per_clip')
        if isinstance(t4.get('parity'), dict):
            c['target_4k']['parity_ok'] = t4['parity'].get('ok')
    tl = full.get('trained_like')
    if isinstance(tl, dict):
        c['trained_like'] = {}
        for k, rec in tl.items():
            if isinstance(rec, dict):
                c['trained_like'][k] = _pick(rec, 'value', 'vs_headline', 'offset_roughness_px_mean')
                if isinstance(rec.get('parity'), dict):
                    c['trained_like'][k]['parity_ok'] = rec['parity'].get('ok')
    if isinstance(full.get('batch4'), dict):
        c['batch4'] = _pick(full['batch4'], 'value')
    cf = full.get('configs')
    if isinstance(cf, dict):
        c['configs'] = {}
        for k, rec in cf.items():
            short = k.split(',')[0].replace('configs', 'c')[:40] + (' train' if 'backward' in k else '')
            if isinstance(rec, dict):
                c['configs'][short] = _num(rec.get('clips_per_sec')) if 'clips_per_sec' in rec else str(rec.get('error', ''))[:60]
    k = full.get('kernels')
    if isinstance(k, dict):  # the five largest rows only: share of the step and fraction of the roofline that bounds each
        top = sorted(k.items(), key=lambda kv: -kv[1].get('ms_per_step', 0.0))[:5]
        c['top_kernels'] = {n[:48]: [_num(v.get('share_of_step')), v.get('bound'),
                                     _num(v.get('frac_of_hbm_peak', v.get('frac_of_mfma_peak')))] for n, v in top}
    for key in ('csrc_sha16', 'library'):
        if key in full:
            c[key] = full[key]
    c['full_report'] = FULL_REPORT
    line = json.dumps(c, allow_nan=False, separators=(',', ':'))
    if len(line) > COMPACT_LIMIT:  # never let an optional object endanger the parse: drop them, least important first
        for key in ('top_kernels', 'configs', 'batch4', 'trained_like', 'stock_rocm', 'target_4k'):
            c.pop(key, None)
            line = json.dumps(c, allow_nan=False, separators=(',', ':'))
            if len(line) <= COMPACT_LIMIT:
                break
    return line


def emit(result, rank, path=None):
    """Rank 0 writes the full report to FULL_REPORT (beside this script; stderr says where) and prints the ONE compact JSON line of
    the run as the LAST line of stdout; every other rank prints nothing."""
    if rank != 0:
        return
    path = path or os.path.join(ROOT, FULL_REPORT)
    try:
        with open(path, 'w') as f:
            json.dump(_sanitize(result), f, indent=1, allow_nan=False)
        print(f'[bench] full report: {path}', file=sys.stderr, flush=True)
    except OSError as e:  # (a read-only tree must not take the measurement down)
        print(f'[bench] could not write {path}: {e}', file=sys.stderr, flush=True)
    sys.stderr.flush()
    print(compact_line(result), flush=True)


def leg_plan(args, world, batch):
    """Which legs of the report run where: 'all' = every rank calls it (it holds barriers / collectives, so NO rank may skip it), 'rank0' =
    rank 0 alone and collective-free, None = not at this N.  Legs that can fail on one rank alone (out of memory in `configs`) or
    that use the host's cores (the CPU oracle, the stock-ops arm) run at N = 1 only: a rank that stopped there would leave the others
    waiting in the next collective (tests/test_dist_cpu.py)."""
    infer = args.mode == 'infer'
    headline = infer and args.workload == 'edvr_l_x4_t5_180x320'
    single = world == 1
    return {
        'fp32_mfma': None if args.no_fp32_leg else 'all',
        'batch4': 'all' if (headline and batch != 4 and not args.no_batch4) else None,
        'roofline': 'rank0' if (not args.no_roofline and (infer or single)) else None,
        'target_4k': 'all' if (headline and not args.no_target_4k) else None,
        'trained_like': 'all' if (headline and not args.no_trained_like) else None,
        'configs': 'all' if (headline and not args.no_configs and single) else None,
        'train': 'all' if (infer and not args.no_train_leg) else None,
        'cpu_baseline': 'rank0' if (infer and single and not args.no_cpu_baseline) else None,
        'stock_rocm_baseline': 'rank0' if (infer and single and not args.no_stock_baseline) else None,
    }


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_spawn(args)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (edvr_amd has no CPU path)'
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=device)  # nccl == RCCL on ROCm
        assert dist.get_world_size() == world
    from edvr_amd import _lib
    assert _lib.lib().edvr_check_device() == 0, _lib.lib().edvr_last_error().decode()
    from edvr_amd import ops as _ops0
    _ops0.HINT_WAIT = True  # the DCN backward picks its dX strategy from THIS iteration's offset statistics (one host sync per iteration):
    #                         two runs of this line then time the same kernels (EDVR_DCN_HINT_WAIT=1)

    if args.workload is None:
        args.workload = 'edvr_l_train_t5_64x64' if args.mode == 'train' else 'edvr_l_x4_t5_180x320'
    cfg = WORKLOADS[args.workload]
    batch = args.batch or cfg['batch']
    net = build_net(cfg, device)
    # per-rank clips (seed + rank, like train.py:53): generated on the CPU, resident in HBM before timing
    x = torch.rand(batch, *cfg['shape'], generator=torch.Generator().manual_seed(rank)).to(device)

    if args.mode == 'train':
        step = make_train_step(net, cfg, batch, device, rank, args.optimizer, x)
    else:
        def step():
            with torch.no_grad():
                return net(x)

    elapsed = timed(step, args.steps, args.warmup, dist, device)

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
            'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic (uniform [0,1) REDS-shaped clips; random-init weights, '
            'manual_seed 10, conv_offset ~ N(0,0.02)/N(0,0.5) so taps are non-integer)',
            'config': {'workload': cfg['desc'], 'clips_per_gpu': batch, 'global_clips': batch * world,
                       'parallelism': (f'DDP x{world}: RCCL gradient all-reduce (82.5 MB fp32)' if args.mode == 'train'
                                       else f'clip-sharded x{world}, no data-path collective'),
                       'world_size': world, 'backend': 'RCCL (torch.distributed nccl)' if world > 1 else 'single process'},
        }
        from edvr_amd.build import source_hash
        result['csrc_sha16'] = source_hash()  # the kernel sources this line was measured on (ties profiles/*.json to the tree)
        result['library'] = _lib.lib().edvr_version().decode()  # (an experiment build says "variant:NAME" here)
        result['dcn_hint_wait'] = True
        if args.mode == 'train':
            result['iters_per_sec'] = round(args.steps / elapsed, 4)
            result['optimizer'] = ('edvr_amd.optim.FusedAdam (one HIP launch for all tensors; arithmetic of torch.optim.Adam)'
                                   if args.optimizer == 'fused' else 'torch.optim.Adam')
    # ---- everything below is outside the timed region
    plan = leg_plan(args, world, batch)
    if plan['fp32_mfma']:
        # the same step with the 3x3 convs (and, in training, their weight gradients) on the fp32 matrix pipe - the exact-fp32-product
        # kernels of rounds 2-4 - timed the same way (all ranks: barriers inside)
        from edvr_amd import ops as _ops
        prev = _ops.set_f4s(False, False)
        try:
            e32 = timed(step, 5, 2, dist, device)
        finally:
            _ops.set_f4s(*prev)
        if rank == 0:
            result['fp32_mfma'] = {'value': round(batch * world * 5 / e32, 4), 'unit': 'clips/s', 'ms_per_step': round(e32 / 5 * 1e3, 3), 'steps': 5, 'warmup': 2,
                                   'dtype': 'f32 (every product on v_mfma_f32_32x32x2_f32 / the vector ALUs)',
                                   'what': 'ops.set_f4s(False, False) = EDVR_WINOGRAD_F4S=0: conv3x3_winograd_f4_kernel, dcn_tapwin_fwd_kernel, conv1x1_stream_kernel (training: + conv3x3_winograd_wgrad_kernel, gemm_nt_kernel) instead of their split-operand forms',
                                   'speedup_of_the_default_path': round(e32 / 5 / (elapsed / args.steps), 4)}
            if args.mode == 'train':
                result['fp32_mfma']['iters_per_sec'] = round(5 / e32, 4)
    if plan['batch4']:
        x4 = x[:4].contiguous()

        def step4():
            with torch.no_grad():
                return net(x4)
        e4 = timed(step4, 5, 2, dist, device)  # all ranks (barriers inside)
        if rank == 0:
            result['batch4'] = {'clips_per_gpu': 4, 'value': round(4 * world * 5 / e4, 4), 'ms_per_step': round(e4 / 5 * 1e3, 3), 'steps': 5,
                                'note': 'same network and clips at 4 clips per GPU (round-1 setting): 7.19 rounds of trunk items run as 8'}
        del x4
    if rank == 0 and plan['roofline']:
        isteps = 1 if args.mode == 'train' else max(1, min(args.steps, 3))
        per = instrumented_pass(step, isteps)
        result['roofline'] = roofline_object(per, isteps, elapsed / args.steps, args.workload, batch == cfg['batch'])
        result['kernels'] = kernel_table(per, isteps, elapsed / args.steps)
    del step
    if plan['target_4k']:
        del x
        torch.cuda.empty_cache()
        x = None
        t4k = target_4k_leg(net, args, device, rank, world, dist)  # all ranks (barriers inside)
        torch.cuda.empty_cache()
        if rank == 0:
            result['target_4k'] = t4k
    if plan['trained_like']:
        tl = trained_like_leg(cfg, batch, args, device, rank, world, dist)  # all ranks (barriers inside)
        if rank == 0:
            result['trained_like'] = tl
            for rec in tl.values():
                rec['vs_headline'] = round(rec['value'] / result['value'], 4)
    if plan['configs']:
        # (N = 1 only: a configuration that fails on ONE rank - out of memory - must not leave the others waiting in a collective)
        net = None  # (the headline network is not needed any more: the remaining legs build their own)
        torch.cuda.empty_cache()
        cf = configs_leg(args, device, rank, world, dist)  # all ranks
        if rank == 0:
            result['configs'] = cf
    if plan['train']:
        del net, x
        torch.cuda.empty_cache()
        tr, tstep = train_leg(args, device, rank, world, dist)  # all ranks: DDP collectives
        del tstep
        torch.cuda.empty_cache()
        if rank == 0:
            result['train'] = tr
    if rank == 0:  # the CPU / stock legs run at N = 1 only (rank 0's host cores)
        ours = None
        if plan['cpu_baseline']:
            result['cpu_baseline'], result['parity'], ours = cpu_baseline_and_parity(cfg, device)
        if plan['stock_rocm_baseline']:
            try:
                result['stock_rocm_baseline'] = stock_rocm_baseline(cfg, device, ours)
            except Exception as e:  # a baseline arm must never take the measurement down (e.g. MIOpen workspace failure)
                result['stock_rocm_baseline'] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
    emit(result, rank)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
