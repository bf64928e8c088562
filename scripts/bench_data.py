"""Input-pipeline throughput on REDS-sized frames (180x320 LQ, 720x1280 GT PNGs): clips/s of plan + decode + byte crop into
staging for a number of decode threads (--host-only, runs without a GPU), or of the whole REDSDeviceLoader incl. the H2D copy
and edvr_frames_u8_to_f32 on the device.  The consumer to keep fed is the training step: 32 clips / 182 ms = 176 clips/s/GPU."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def make_dataset(root, clips, frames):
    from PIL import Image
    rs = np.random.RandomState(0)
    for kind, (h, w) in (('lq', (180, 320)), ('gt', (720, 1280))):
        yy, xx = np.mgrid[0:h, 0:w]
        for c in clips:
            d = os.path.join(root, kind, c)
            os.makedirs(d, exist_ok=True)
            for f in range(frames):
                if kind == 'gt' and f >= 8:  # 8 distinct 720p frames, hard-linked for the rest (encode time; decode cost is the same)
                    os.link(os.path.join(d, f'{f % 8:08d}.png'), os.path.join(d, f'{f:08d}.png'))
                    continue
                # smooth content + mild noise: PNG sizes (and decode cost) in the range of natural frames, not of white noise
                img = np.stack([(yy * 255 // h + f) % 256, (xx * 255 // w + 2 * f) % 256, ((yy + xx) // 4 + f) % 256], -1)
                img = (img + rs.randint(0, 6, img.shape)).clip(0, 255).astype(np.uint8)
                Image.fromarray(img).save(os.path.join(d, f'{f:08d}.png'), compress_level=3)
    meta = os.path.join(root, 'meta_info.txt')
    with open(meta, 'w') as fh:
        fh.writelines(f'{c} {frames} (720,1280,3)\n' for c in clips)
    return meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--host-only', action='store_true')
    ap.add_argument('--threads', type=int, nargs='+', default=[1, 4, 8, 16, 32])
    ap.add_argument('--batches', type=int, default=6)
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    from edvr_amd import data as D
    import random
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as root:
        t0 = time.time()
        meta = make_dataset(root, ['001'], 100)
        gt_bytes = os.path.getsize(os.path.join(root, 'gt', '001', '00000000.png'))
        print(f'dataset written in {time.time() - t0:.1f}s; GT PNG {gt_bytes / 1e6:.2f} MB', flush=True)
        opt = dict(dataroot_gt=os.path.join(root, 'gt'), dataroot_lq=os.path.join(root, 'lq'), dataroot_flow=None, meta_info_file=meta,
                   io_backend=dict(type='disk'), gt_size=256, scale=4, num_frame=5, interval_list=[1], random_reverse=False,
                   use_flip=True, use_rot=True, val_partition='REDS4')
        for th in a.threads:
            if a.host_only:
                pl = D.REDSClipPlanner(opt)
                rng = random.Random(0)
                lq = np.empty((a.batch, 5, 64, 64, 3), np.uint8)
                gt = np.empty((a.batch, 1, 256, 256, 3), np.uint8)
                with ThreadPoolExecutor(th) as pool:
                    t0 = time.time()
                    for b in range(a.batches):
                        plans = [pl.plan(rng.randrange(len(pl)), rng) for _ in range(a.batch)]
                        list(pool.map(lambda q: pl.load(q[1], lq[q[0]], gt[q[0], 0]), enumerate(plans)))
                    dt = time.time() - t0
            else:
                import torch
                loader = D.REDSDeviceLoader(opt, a.batch, ratio=100, seed=0, num_threads=th)
                loader.next()
                torch.cuda.synchronize()
                t0 = time.time()
                for b in range(a.batches):
                    batch = loader.next()
                    assert batch['lq'].shape == (a.batch, 5, 3, 64, 64)
                torch.cuda.synchronize()
                dt = time.time() - t0
                loader.close()
            print(f'threads {th:3d}: {a.batches * a.batch / dt:8.1f} clips/s ({"host only" if a.host_only else "device loader"})', flush=True)


if __name__ == '__main__':
    main()
