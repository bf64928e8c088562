"""The reference's test entry point for EDVR (basicsr/test.py + VideoBaseModel.dist_validation, models/video_base_model.py:19-119)
on the pieces of this package: a folder tree of PNG clips (REDS4 / Vid4 layout) or the Vimeo90K-Test list -> EDVR -> PSNR per
clip and on average, everything after the PNG decode on the GPU.  One process per GPU; clips are sharded over the ranks as
dist_validation does (folder i goes to rank i % world) and the per-clip results are gathered on rank 0.

    python scripts/test_reds.py --lq datasets/REDS4/sharp_bicubic --gt datasets/REDS4/GT --weights EDVR_L_x4_SR_REDS_official.pth
    python scripts/test_reds.py --vimeo-meta meta_info_Vimeo90K_test_GT.txt --lq .../LRx4/sequences --gt .../sequences --num-frame 7 ...

Options mirror options/test/EDVR/*.yml (network_g keys, padding, crop_border, test_y_channel).  Saving images
(`val.save_img`) is left to the caller: `validate_clip` returns the restored frames.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def evaluate(args, log=print):
    from edvr_amd import EDVR, dist as D, metrics
    from edvr_amd.data import VideoTestClips, VideoTestVimeo90KClips
    from edvr_amd.optim import load_network
    rank, world = D.get_dist_info() if torch.distributed.is_initialized() else D.init_dist()
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(device)
    net = EDVR(num_in_ch=3, num_out_ch=3, num_feat=args.num_feat, num_frame=args.num_frame, deformable_groups=8,
               num_extract_block=5, num_reconstruct_block=args.num_reconstruct_block, center_frame_idx=None, hr_in=args.hr_in,
               with_predeblur=args.with_predeblur, with_tsa=not args.no_tsa).to(device).eval()
    if args.weights:
        load_network(net, args.weights, strict=True)
    base = dict(dataroot_gt=args.gt, dataroot_lq=args.lq, io_backend=dict(type='disk'), num_frame=args.num_frame, padding=args.padding)
    results = {}
    if args.vimeo_meta:
        ds = VideoTestVimeo90KClips(dict(base, name='Vimeo90K-Test', meta_info_file=args.vimeo_meta, cache_data=False), device=device)
        scores = []
        with torch.no_grad():
            for i in range(rank, len(ds), world):  # one 7-frame window per item
                item = ds[i]
                out = net(item['lq'][None])
                scores += metrics.calculate_psnr(out, item['gt'][None], args.crop_border, args.test_y_channel)
        results['vimeo90k'] = (sum(scores), len(scores))
    else:
        ds = VideoTestClips(dict(base, name=args.name, cache_data=True), device=device)
        for folder in ds.folders[rank::world]:
            lq, gt = ds.clip(folder)
            _, psnr = metrics.validate_clip(net, lq, gt, num_frame=args.num_frame, padding=args.padding, batch=args.batch,
                                            crop_border=args.crop_border, test_y_channel=args.test_y_channel)
            results[folder] = (sum(psnr), len(psnr))
            ds._cache.pop(folder, None)  # one clip resident at a time
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, results)
        results = {}
        for part in gathered:
            for k, (s, n) in part.items():
                s0, n0 = results.get(k, (0.0, 0))
                results[k] = (s0 + s, n0 + n)
    summary = {k: s / max(n, 1) for k, (s, n) in sorted(results.items())}
    if rank == 0:
        for k, v in summary.items():
            log(f'{k}: PSNR {v:.4f} dB')
        if summary:
            # VideoBaseModel averages the per-folder averages (video_base_model.py:146-155)
            log(f'average over {len(summary)} folder(s): {sum(summary.values()) / len(summary):.4f} dB')
    return summary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lq', required=True)
    ap.add_argument('--gt', required=True)
    ap.add_argument('--weights', default=None)
    ap.add_argument('--name', default='REDS4', help='REDS4 | Vid4 | REDSofficial (folder layout of VideoTestDataset)')
    ap.add_argument('--vimeo-meta', default=None, help='meta_info_Vimeo90K_test_GT.txt: evaluate the Vimeo90K-Test septuplets instead')
    ap.add_argument('--num-feat', type=int, default=128)
    ap.add_argument('--num-reconstruct-block', type=int, default=40)
    ap.add_argument('--num-frame', type=int, default=5)
    ap.add_argument('--hr-in', action='store_true')
    ap.add_argument('--with-predeblur', action='store_true')
    ap.add_argument('--no-tsa', action='store_true')
    ap.add_argument('--padding', default='reflection_circle')
    ap.add_argument('--crop-border', type=int, default=0)
    ap.add_argument('--test-y-channel', action='store_true')
    ap.add_argument('--batch', type=int, default=4)
    evaluate(ap.parse_args())


if __name__ == '__main__':
    main()
