"""The reference's training loop for EDVR (basicsr/train.py:110-220 + EDVRModel, models/edvr_model.py / sr_model.py) on the
pieces of this package, end to end on the GPU: REDS PNG folders -> REDSDeviceLoader -> EDVR -> Charbonnier -> FusedAdam ->
CosineAnnealingRestartLR, TSA-only warm-up, periodic validation on VideoTestClips with device PSNR, checkpoints in the
reference's formats.  One process per GPU (torchrun / torch.distributed.run sets RANK / WORLD_SIZE), clips sharded by
EnlargedSampler, gradients all-reduced over RCCL by DDP.

    python scripts/train_reds.py --gt datasets/REDS/train_sharp --lq datasets/REDS/train_sharp_bicubic/X4 \
        --meta meta_info_REDS_GT.txt --iters 600000 [--val-gt ... --val-lq ...]

It is a caller of the hot path, kept small on purpose: option parsing, logging and the experiment directory layout of
basicsr/train.py are out of scope (control plane).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def train(args, log=print):
    from edvr_amd import EDVR, dist as D, metrics
    from edvr_amd.autograd import charbonnier_loss
    from edvr_amd.data import REDSDeviceLoader, VideoTestClips
    from edvr_amd.optim import (CosineAnnealingRestartLR, load_network, make_optimizer, resume_training, save_network,
                                save_training_state, tsa_freeze_schedule)
    if getattr(args, "no_f4", False):  # forward / data-gradient convs on F(2x2) / direct kernels only (~2e-7 instead of ~1e-6 relative rounding, slower)
        from edvr_amd import ops
        ops.set_f4(training=False)
    rank, world = D.get_dist_info() if torch.distributed.is_initialized() else D.init_dist()
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed + rank)
    net = EDVR(num_in_ch=3, num_out_ch=3, num_feat=args.num_feat, num_frame=args.num_frame, deformable_groups=8,
               num_extract_block=5, num_reconstruct_block=args.num_reconstruct_block, center_frame_idx=None, hr_in=False,
               with_predeblur=False, with_tsa=True).to(device)
    if args.pretrain:  # path.pretrain_network_g of the reference's option files; on resume: net_g_<iter>.pth next to <iter>.state
        load_network(net, args.pretrain)
    it, epoch = 0, 0
    state = None
    if args.resume:
        state = torch.load(args.resume, map_location='cpu')  # optimizer 'step' counters stay host-side (no per-parameter sync in step())
        it, epoch = state['iter'], state['epoch']
    # phase of the TSA schedule the run (re)starts in: during the warm-up (1 <= it < tsa_iter) only the fusion module trains and the
    # DDP reducer must skip the frozen parameters; past it every parameter trains and no unused-parameter search is needed.  (The
    # reference re-applies neither on resume - edvr_model.py:55-69 acts at it == 1 and it == tsa_iter only - and so trains the
    # frozen parameters after a resume inside the warm-up.)
    warmup = bool(args.tsa_iter) and it < args.tsa_iter
    if warmup and it >= 1:
        for name, p in net.named_parameters():
            if 'fusion' not in name:
                p.requires_grad = False
    model = D.wrap_ddp(net, find_unused_parameters=warmup) if world > 1 else net
    opt = make_optimizer(net, lr=args.lr, dcn_lr_mul=args.dcn_lr_mul, betas=(0.9, 0.99))
    sched = CosineAnnealingRestartLR(opt, periods=args.periods, restart_weights=args.restart_weights, eta_min=1e-7)
    data_opt = dict(dataroot_gt=args.gt, dataroot_lq=args.lq, dataroot_flow=None, meta_info_file=args.meta, val_partition=args.val_partition,
                    io_backend=dict(type='disk'), num_frame=args.num_frame, gt_size=args.gt_size, interval_list=[1], random_reverse=False,
                    use_flip=True, use_rot=True, scale=4)
    loader = REDSDeviceLoader(data_opt, args.batch, device=device, rank=rank, world_size=world, ratio=args.enlarge_ratio,
                              seed=args.seed, num_threads=args.threads)
    val = None
    if args.val_lq:
        val = VideoTestClips(dict(name='REDS4', dataroot_gt=args.val_gt, dataroot_lq=args.val_lq, io_backend=dict(type='disk'),
                                  cache_data=True, num_frame=args.num_frame, padding='reflection_circle'), device=device)
    if state is not None:
        resume_training(state, [opt], [sched])
        loader.reset(epoch)
    losses = []
    while it < args.iters:
        batch = loader.next()
        if batch is None:  # end of the (enlarged) epoch
            epoch += 1
            loader.reset(epoch)
            continue
        it += 1
        if tsa_freeze_schedule(model, it, args.tsa_iter):
            log(f'iter {it}: trainable parameter set changed (TSA schedule)')
            if it == args.tsa_iter and world > 1:
                model = D.rewrap_ddp(model, find_unused_parameters=False)  # all parameters train from here on
        opt.zero_grad(set_to_none=True)
        out = model(batch['lq'])
        loss = charbonnier_loss(out, batch['gt'])  # CharbonnierLoss, loss_weight 1.0, reduction: sum (options/train/EDVR/*.yml pixel_opt)
        loss.backward()
        opt.step()
        sched.step()
        if it % args.print_freq == 0 or it == args.iters:
            v = D.reduce_scalar(loss.detach(), device) if world > 1 else float(loss.detach())
            losses.append(float(v))
            log(f'epoch {epoch} iter {it} lr {opt.param_groups[0]["lr"]:.3e} l_pix {float(v):.4e}')
        if args.save_dir and rank == 0 and (it % args.save_freq == 0 or it == args.iters):
            os.makedirs(args.save_dir, exist_ok=True)
            save_network(net, os.path.join(args.save_dir, f'net_g_{it}.pth'))
            save_training_state(os.path.join(args.save_dir, f'{it}.state'), epoch, it, [opt], [sched])
        if val is not None and (it % args.val_freq == 0 or it == args.iters):
            net.eval()
            folders = val.folders[rank::world]  # clips sharded over ranks, as dist_validation does (video_base_model.py:31-35)
            scores = {}
            for folder in folders:
                lq, gt = val.clip(folder)
                _, psnr = metrics.validate_clip(net, lq, gt, num_frame=args.num_frame, padding='reflection_circle', batch=args.val_batch)
                scores[folder] = sum(psnr) / len(psnr)
            net.train()
            log(f'iter {it} validation PSNR ' + ', '.join(f'{k}: {v:.3f}' for k, v in scores.items()))
    loader.close()
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gt', required=True)
    ap.add_argument('--lq', required=True)
    ap.add_argument('--meta', required=True)
    ap.add_argument('--val-gt')
    ap.add_argument('--val-lq')
    ap.add_argument('--val-partition', default='REDS4')
    ap.add_argument('--no-f4', action='store_true', help='keep the F(4x4,3x3) Winograd kernel out of the training path (edvr_amd.ops.set_f4)')
    ap.add_argument('--num-feat', type=int, default=128)              # EDVR-L (options/train/EDVR/train_EDVR_L_x4_SR_REDS_*.yml)
    ap.add_argument('--num-reconstruct-block', type=int, default=40)
    ap.add_argument('--num-frame', type=int, default=5)
    ap.add_argument('--gt-size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=4)                   # batch_size_per_gpu of the reference configs; 32 fits easily here
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--enlarge-ratio', type=int, default=200)
    ap.add_argument('--iters', type=int, default=600000)
    ap.add_argument('--lr', type=float, default=4e-4)
    ap.add_argument('--dcn-lr-mul', type=float, default=1)
    ap.add_argument('--periods', type=int, nargs='+', default=[50000, 100000, 150000, 150000, 150000])
    ap.add_argument('--restart-weights', type=float, nargs='+', default=[1, 0.5, 0.5, 0.5, 0.5])
    ap.add_argument('--tsa-iter', type=int, default=50000)
    ap.add_argument('--print-freq', type=int, default=100)
    ap.add_argument('--save-freq', type=int, default=5000)
    ap.add_argument('--val-freq', type=int, default=5000)
    ap.add_argument('--val-batch', type=int, default=4)
    ap.add_argument('--save-dir', default=None)
    ap.add_argument('--resume', default=None)
    ap.add_argument('--pretrain', default=None)
    ap.add_argument('--seed', type=int, default=10)
    train(ap.parse_args())


if __name__ == '__main__':
    main()
