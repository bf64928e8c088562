"""CPU oracle of the input pipeline - TEST INFRASTRUCTURE, never imported by the product (edvr_amd/).

numpy restatement of what the reference does between the decoded frame and the tensor the network sees:
  imfrombytes(float32=True)   basicsr/utils/img_util.py:101-123     uint8 BGR -> float32 / 255.
  REDSDataset.__getitem__     basicsr/data/reds_dataset.py:106-234  frame selection, temporal augmentation, crop, flip/rot
  paired_random_crop          basicsr/data/transforms.py:25-81
  augment                     basicsr/data/transforms.py:84-151
  img2tensor                  basicsr/utils/img_util.py:9-33        BGR->RGB, HWC->CHW
  read_img_seq                basicsr/data/data_util.py:11-33       (validation: no crop / augmentation)
  EnlargedSampler             basicsr/data/data_sampler.py:6-49
Like the reference it works on float arrays in cv2's BGR order and draws from Python's `random` in the reference's order.

Pinned: tests/golden/data_pipeline.pt holds outputs of the reference's own REDSDataset / transforms / img_util / EnlargedSampler
sources executed in this container (oracle/make_golden.py::data_case; cv2 - absent here - replaced by a 3-function numpy stand-in
for flip / cvtColor / imdecode, storage by an in-memory client serving `synthetic_frame`); tests/test_data_cpu.py checks this
restatement against them bit for bit, including the state of the random stream after every sample.
"""
import math
import zlib

import numpy as np
import torch


def synthetic_frame(kind, clip, frame, h, w):
    """Deterministic uint8 BGR test frame (h, w, 3) for (kind, clip, frame): what the in-memory storage of the golden generator
    serves and what the tests write to PNG files.  Smooth ramp + noise, so that crops / flips / transposes are all visible."""
    seed = zlib.crc32(f'{kind}/{clip}/{frame}'.encode())
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(yy * 3 + xx) % 256, (yy + xx * 5) % 256, (yy * 7 + xx * 2 + 31) % 256], -1)
    return ((base + rs.randint(0, 64, (h, w, 3))) % 256).astype(np.uint8)


def imfrombytes_float(bgr_u8):
    """img_util.py:119-121."""
    return bgr_u8.astype(np.float32) / 255.


def paired_random_crop(img_gt, img_lqs, gt_patch_size, scale, rng):
    """transforms.py:25-81 for one GT image and a list of LQ images; returns (gt, lqs, top, left)."""
    h_lq, w_lq, _ = img_lqs[0].shape
    h_gt, w_gt, _ = img_gt.shape
    lq_patch_size = gt_patch_size // scale
    if h_gt != h_lq * scale or w_gt != w_lq * scale:
        raise ValueError(f'Scale mismatches. GT ({h_gt}, {w_gt}) is not {scale}x multiplication of LQ ({h_lq}, {w_lq}).')
    if h_lq < lq_patch_size or w_lq < lq_patch_size:
        raise ValueError(f'LQ ({h_lq}, {w_lq}) is smaller than patch size ({lq_patch_size}, {lq_patch_size}).')
    top = rng.randint(0, h_lq - lq_patch_size)
    left = rng.randint(0, w_lq - lq_patch_size)
    lqs = [v[top:top + lq_patch_size, left:left + lq_patch_size, ...] for v in img_lqs]
    top_gt, left_gt = int(top * scale), int(left * scale)
    gt = img_gt[top_gt:top_gt + gt_patch_size, left_gt:left_gt + gt_patch_size, ...]
    return gt, lqs, top, left


def augment(imgs, hflip, rotation, rng):
    """transforms.py:84-151 without flows; returns (imgs, (hflip, vflip, rot90))."""
    hflip = hflip and rng.random() < 0.5
    vflip = rotation and rng.random() < 0.5
    rot90 = rotation and rng.random() < 0.5

    def one(img):
        if hflip:
            img = img[:, ::-1]
        if vflip:
            img = img[::-1]
        if rot90:
            img = img.transpose(1, 0, 2)
        return img

    return [one(i) for i in imgs], (hflip, vflip, rot90)


def apply_aug_flags(img, flags):
    """The body of augment()'s _augment for explicit flag bits (1 hflip, 2 vflip, 4 rot90)."""
    if flags & 1:
        img = img[:, ::-1]
    if flags & 2:
        img = img[::-1]
    if flags & 4:
        img = img.transpose(1, 0, 2)
    return img


def img2tensor(img):
    """img_util.py:22-28 with bgr2rgb=True, float32=True."""
    return torch.from_numpy(img[:, :, ::-1].transpose(2, 0, 1).copy()).float()


def reds_keys(meta_lines, val_partition):
    """reds_dataset.py:62-81."""
    keys = []
    for line in meta_lines:
        folder, frame_num, _ = line.split(' ')
        keys.extend(f'{folder}/{i:08d}' for i in range(int(frame_num)))
    if val_partition == 'REDS4':
        val = ['000', '011', '015', '020']
    elif val_partition == 'official':
        val = [f'{v:03d}' for v in range(240, 270)]
    else:
        raise ValueError(f'Wrong validation partition {val_partition}.')
    return [k for k in keys if k.split('/')[0] not in val]


def reds_getitem(keys, opt, index, rng, fetch):
    """reds_dataset.py:106-234 (no flow).  fetch(kind, clip, frame_name) -> uint8 BGR (h, w, 3) stands for file client + cv2.imdecode.
    Returns dict(lq (t, 3, p, p), gt (3, P, P), key, frames, top, left, aug)."""
    scale, gt_size, num_frame = opt['scale'], opt['gt_size'], opt['num_frame']
    half = num_frame // 2
    key = keys[index]
    clip_name, frame_name = key.split('/')
    center = int(frame_name)
    interval = rng.choice(opt['interval_list'])
    start, end = center - half * interval, center + half * interval
    while start < 0 or end > 99:
        center = rng.randint(0, 99)
        start, end = center - half * interval, center + half * interval
    frame_name = f'{center:08d}'
    neighbors = list(range(center - half * interval, center + half * interval + 1, interval))
    if opt['random_reverse'] and rng.random() < 0.5:
        neighbors.reverse()
    assert len(neighbors) == num_frame
    img_gt = imfrombytes_float(fetch('gt', clip_name, frame_name))
    img_lqs = [imfrombytes_float(fetch('lq', clip_name, f'{n:08d}')) for n in neighbors]
    img_gt, img_lqs, top, left = paired_random_crop(img_gt, img_lqs, gt_size, scale, rng)
    imgs, aug = augment(img_lqs + [img_gt], opt['use_flip'], opt['use_rot'], rng)
    tens = [img2tensor(i) for i in imgs]
    return dict(lq=torch.stack(tens[:-1], 0), gt=tens[-1], key=key, frames=neighbors, center=center, top=top, left=left, aug=aug)


def vimeo90k_neighbor_list(num_frame):
    """vimeo90k_dataset.py:69-71."""
    return [i + (9 - num_frame) // 2 for i in range(num_frame)]


def vimeo90k_getitem(keys, opt, index, rng, fetch, neighbor_list):
    """vimeo90k_dataset.py:73-131.  `neighbor_list` is the dataset object's list and is REVERSED IN PLACE when the reverse draw
    hits (:82-83) - the orientation persists into the following samples, as in the reference.  fetch(kind, 'clip/seq', 'im<n>') ->
    uint8 BGR.  The GT frame is always im4."""
    if opt['random_reverse'] and rng.random() < 0.5:
        neighbor_list.reverse()
    key = keys[index]
    img_gt = imfrombytes_float(fetch('gt', key, 'im4'))
    img_lqs = [imfrombytes_float(fetch('lq', key, f'im{n}')) for n in neighbor_list]
    img_gt, img_lqs, top, left = paired_random_crop(img_gt, img_lqs, opt['gt_size'], opt['scale'], rng)
    imgs, aug = augment(img_lqs + [img_gt], opt['use_flip'], opt['use_rot'], rng)
    tens = [img2tensor(i) for i in imgs]
    return dict(lq=torch.stack(tens[:-1], 0), gt=tens[-1], key=key, frames=list(neighbor_list), top=top, left=left, aug=aug)


def read_img_seq(frames_bgr_u8, require_mod_crop=False, scale=1):
    """data_util.py:11-33 on decoded frames: (t, 3, h, w) RGB float32 in [0, 1]."""
    imgs = [imfrombytes_float(f) for f in frames_bgr_u8]
    if require_mod_crop:
        imgs = [i[:i.shape[0] - i.shape[0] % scale, :i.shape[1] - i.shape[1] % scale] for i in imgs]
    return torch.stack([img2tensor(i) for i in imgs], 0)


def enlarged_sampler_indices(dataset_size, num_replicas, rank, ratio, epoch):
    """data_sampler.py:21-46."""
    num_samples = math.ceil(dataset_size * ratio / num_replicas)
    total = num_samples * num_replicas
    g = torch.Generator()
    g.manual_seed(epoch)
    idx = [v % dataset_size for v in torch.randperm(total, generator=g).tolist()]
    return idx[rank:total:num_replicas]
