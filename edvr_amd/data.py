"""Input pipeline of the EDVR hot path, MI355X-first.

Reference: REDSDataset (basicsr/data/reds_dataset.py:12-237) decodes PNGs to float32 HWC in DataLoader worker processes, crops
and flips float arrays per image (transforms.py:25-151), converts to CHW tensors (img_util.py:9-33), the default collate copies
them again and CUDAPrefetcher moves 4 bytes per sample over PCIe (prefetch_dataloader.py:84-126).

Here the host does only what must be on the host - frame selection (the reference's random draws, in its order), PNG decode and
a byte crop straight into pinned staging - in threads of ONE process per GPU (PIL's decoder releases the GIL), uint8 patches
cross PCIe on a side stream (1 byte per sample), and one HIP launch per tensor (`edvr_frames_u8_to_f32`, csrc/data.hip) does
flip / transpose / HWC->CHW / division by 255 for the whole batch.  `REDSDeviceLoader.next()` hands out device tensors like
CUDAPrefetcher.next() does, already waited for on the current stream.

The storage formats are the reference's: PNG folders `<root>/<clip>/<frame:08d>.png` (disk backend) and its LMDB layout (keys
`<clip>/<frame:08d>`, PNG-encoded values; file_client.py:76-144) when the `lmdb` module is importable.
"""
import contextlib
import io
import math
import os
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

AUG_HFLIP, AUG_VFLIP, AUG_ROT90 = 1, 2, 4


# ------------------------------------------------------------------------------------------------ storage + decode (host)
class DiskClient:
    """Disk backend (file_client.py:60-74): `<root>/<clip>/<frame>.png`."""

    def __init__(self, roots):
        self.roots = {k: Path(v) for k, v in roots.items()}

    def path(self, kind, clip, frame):
        return self.roots[kind] / clip / f'{frame}.png'

    def get(self, kind, clip, frame):
        with open(self.path(kind, clip, frame), 'rb') as f:
            return f.read()

    def size(self, kind, clip, frame):
        from PIL import Image
        with Image.open(self.path(kind, clip, frame)) as im:  # reads the header only
            return im.size[1], im.size[0]


class LmdbClient:
    """LMDB backend (file_client.py:76-144): one environment per kind, key `<clip>/<frame>`, value = PNG bytes."""

    def __init__(self, roots):
        try:
            import lmdb
        except ImportError:
            raise ImportError('Please install lmdb to enable the lmdb io_backend.')
        self.envs = {k: lmdb.open(str(v), readonly=True, lock=False, readahead=False) for k, v in roots.items()}

    def get(self, kind, clip, frame):
        with self.envs[kind].begin(write=False) as txn:
            return bytes(txn.get(f'{clip}/{frame}'.encode('ascii')))

    def size(self, kind, clip, frame):
        return image_size(self.get(kind, clip, frame))


def decode_image(content):
    """PNG/JPEG bytes -> uint8 (h, w, 3) RGB.  Lossless formats decode to the bytes cv2.imdecode gives (imfrombytes,
    img_util.py:101-123), in RGB order instead of BGR - which is where img2tensor's bgr2rgb ends up anyway."""
    from PIL import Image
    with Image.open(io.BytesIO(content)) as im:
        return np.asarray(im.convert('RGB'))


def image_size(content):
    """(h, w) from the header only."""
    from PIL import Image
    with Image.open(io.BytesIO(content)) as im:
        return im.size[1], im.size[0]


# ------------------------------------------------------------------------------------------------ sample planning (host)
def reds_keys(meta_info_file, val_partition):
    """Keys `clip/frame` of the training split (reds_dataset.py:62-81)."""
    keys = []
    with open(meta_info_file, 'r') as fin:
        for line in fin:
            folder, frame_num, _ = line.split(' ')
            keys.extend(f'{folder}/{i:08d}' for i in range(int(frame_num)))
    if val_partition == 'REDS4':
        val = {'000', '011', '015', '020'}
    elif val_partition == 'official':
        val = {f'{v:03d}' for v in range(240, 270)}
    else:
        raise ValueError(f'Wrong validation partition {val_partition}.Supported ones are [\'official\', \'REDS4\'].')
    return [k for k in keys if k.split('/')[0] not in val]


class ClipPlan:
    """Every random decision of one training sample: which frames, where to crop, how to augment."""
    __slots__ = ('key', 'clip', 'center', 'frames', 'top', 'left', 'flags')

    def __init__(self, key, clip, center, frames, top, left, flags):
        self.key, self.clip, self.center, self.frames, self.top, self.left, self.flags = key, clip, center, frames, top, left, flags

    def __repr__(self):
        return f'ClipPlan({self.key}: frames {self.frames} of {self.clip}, crop ({self.top}, {self.left}), aug {self.flags})'


class REDSClipPlanner:
    """The decisions of REDSDataset.__getitem__ (reds_dataset.py:106-234), separated from the pixels.  `opt` has the reference's
    keys (dataroot_gt, dataroot_lq, meta_info_file, val_partition, io_backend, num_frame, gt_size, interval_list, random_reverse,
    use_flip, use_rot, scale).  plan(index, rng) draws from `rng` (a random.Random or the `random` module) exactly the values the
    reference draws, in its order: interval, [new centre frames while the window leaves 0..99], [reverse], crop top, crop left,
    [hflip], [vflip], [rot90]."""

    def __init__(self, opt, client=None):
        if opt.get('dataroot_flow') is not None:
            raise NotImplementedError('optical-flow inputs (dataroot_flow) are not on the EDVR path')
        assert opt['num_frame'] % 2 == 1, f'num_frame should be odd number, but got {opt["num_frame"]}'
        self.opt = opt
        self.num_frame, self.half = opt['num_frame'], opt['num_frame'] // 2
        self.scale, self.gt_size = opt['scale'], opt['gt_size']
        self.lq_size = self.gt_size // self.scale
        self.keys = reds_keys(opt['meta_info_file'], opt['val_partition'])
        self.interval_list, self.random_reverse = opt['interval_list'], opt['random_reverse']
        roots = {'lq': opt['dataroot_lq'], 'gt': opt['dataroot_gt']}
        backend = dict(opt['io_backend'])['type']
        if client is not None:  # any object with get(kind, clip, frame) -> bytes and size(kind, clip, frame) -> (h, w)
            self.client = client
        elif backend == 'lmdb':
            self.client = LmdbClient(roots)
        elif backend == 'disk':
            self.client = DiskClient(roots)
        else:
            raise ValueError(f'io_backend {backend} is not supported (disk, lmdb)')
        self._shape = {}  # clip -> ((h_lq, w_lq), (h_gt, w_gt)): all frames of a REDS clip have one size

    def __len__(self):
        return len(self.keys)

    def clip_shapes(self, clip, frame):
        if clip not in self._shape:
            self._shape[clip] = (self.client.size('lq', clip, frame), self.client.size('gt', clip, frame))
        return self._shape[clip]

    def plan(self, index, rng=random):
        key = self.keys[index]
        clip, frame_name = key.split('/')
        center = int(frame_name)
        interval = rng.choice(self.interval_list)
        while center - self.half * interval < 0 or center + self.half * interval > 99:  # each clip has frames 0..99
            center = rng.randint(0, 99)
        frames = list(range(center - self.half * interval, center + self.half * interval + 1, interval))
        if self.random_reverse and rng.random() < 0.5:
            frames.reverse()
        assert len(frames) == self.num_frame, f'Wrong length of neighbor list: {len(frames)}'
        (h_lq, w_lq), (h_gt, w_gt) = self.clip_shapes(clip, f'{center:08d}')
        if h_gt != h_lq * self.scale or w_gt != w_lq * self.scale:
            raise ValueError(f'Scale mismatches. GT ({h_gt}, {w_gt}) is not {self.scale}x multiplication of LQ ({h_lq}, {w_lq}).')
        if h_lq < self.lq_size or w_lq < self.lq_size:
            raise ValueError(f'LQ ({h_lq}, {w_lq}) is smaller than patch size ({self.lq_size}, {self.lq_size}). '
                             f'Please remove {clip}/{center:08d}.')
        top = rng.randint(0, h_lq - self.lq_size)
        left = rng.randint(0, w_lq - self.lq_size)
        flags = 0
        if self.opt['use_flip'] and rng.random() < 0.5:
            flags |= AUG_HFLIP
        if self.opt['use_rot'] and rng.random() < 0.5:
            flags |= AUG_VFLIP
        if self.opt['use_rot'] and rng.random() < 0.5:
            flags |= AUG_ROT90
        return ClipPlan(key, clip, center, frames, top, left, flags)

    def load(self, plan, lq_out=None, gt_out=None):
        """Decode the frames of `plan` and crop bytes: lq (t, p, p, 3), gt (P, P, 3) uint8 RGB, written into lq_out / gt_out
        (views of pinned staging) when given."""
        p, P, s = self.lq_size, self.gt_size, self.scale
        if lq_out is None:
            lq_out = np.empty((self.num_frame, p, p, 3), np.uint8)
        if gt_out is None:
            gt_out = np.empty((P, P, 3), np.uint8)
        for i, f in enumerate(plan.frames):
            lq_out[i] = decode_image(self.client.get('lq', plan.clip, f'{f:08d}'))[plan.top:plan.top + p, plan.left:plan.left + p]
        gt_out[...] = decode_image(self.client.get('gt', plan.clip, f'{plan.center:08d}'))[plan.top * s:plan.top * s + P,
                                                                                          plan.left * s:plan.left * s + P]
        return lq_out, gt_out


class Vimeo90KClipPlanner:
    """The decisions of Vimeo90KDataset.__getitem__ (the TRAINING set, basicsr/data/vimeo90k_dataset.py:10-134) for the same loader:
    7-frame sequences `<root>/<clip>/<seq>/im1.png .. im7.png`, keys from the meta file (`00001/0001 7 (256,448,3)`), GT = im4,
    the centred window of `num_frame` frames, and the reference's draws in its order: [reverse], crop top, crop left, [hflip],
    [vflip], [rot90].  The reference reverses its neighbour list IN PLACE (:82-83), so the orientation persists from one sample to
    the next: plan() keeps that state too (one planner = one dataset object)."""

    def __init__(self, opt, client=None):
        self.opt = opt
        self.num_frame = opt['num_frame']
        self.scale, self.gt_size = opt['scale'], opt['gt_size']
        self.lq_size = self.gt_size // self.scale
        with open(opt['meta_info_file'], 'r') as fin:
            self.keys = [line.split(' ')[0] for line in fin]
        self.neighbor_list = [i + (9 - self.num_frame) // 2 for i in range(self.num_frame)]
        self.random_reverse = opt['random_reverse']
        roots = {'lq': opt['dataroot_lq'], 'gt': opt['dataroot_gt']}
        backend = dict(opt['io_backend'])['type']
        if client is not None:
            self.client = client
        elif backend == 'lmdb':
            self.client = LmdbClient(roots)
        elif backend == 'disk':
            self.client = DiskClient(roots)
        else:
            raise ValueError(f'io_backend {backend} is not supported (disk, lmdb)')

    def __len__(self):
        return len(self.keys)

    def reset_epoch(self):
        """Canonical orientation of the neighbour list at the start of an epoch: what a fresh DataLoader worker's copy of the
        dataset has in the reference (workers are re-created per epoch), and what makes an epoch a function of (seed, epoch) only."""
        self.neighbor_list = [i + (9 - self.num_frame) // 2 for i in range(self.num_frame)]

    def plan(self, index, rng=random):
        if self.random_reverse and rng.random() < 0.5:
            self.neighbor_list.reverse()
        key = self.keys[index]
        frames = list(self.neighbor_list)
        h_lq, w_lq = self.client.size('lq', key, f'im{frames[0]}')
        h_gt, w_gt = self.client.size('gt', key, 'im4')
        if h_gt != h_lq * self.scale or w_gt != w_lq * self.scale:
            raise ValueError(f'Scale mismatches. GT ({h_gt}, {w_gt}) is not {self.scale}x multiplication of LQ ({h_lq}, {w_lq}).')
        if h_lq < self.lq_size or w_lq < self.lq_size:
            raise ValueError(f'LQ ({h_lq}, {w_lq}) is smaller than patch size ({self.lq_size}, {self.lq_size}). Please remove {key}.')
        top = rng.randint(0, h_lq - self.lq_size)
        left = rng.randint(0, w_lq - self.lq_size)
        flags = 0
        if self.opt['use_flip'] and rng.random() < 0.5:
            flags |= AUG_HFLIP
        if self.opt['use_rot'] and rng.random() < 0.5:
            flags |= AUG_VFLIP
        if self.opt['use_rot'] and rng.random() < 0.5:
            flags |= AUG_ROT90
        return ClipPlan(key, key, 4, frames, top, left, flags)

    def load(self, plan, lq_out=None, gt_out=None):
        p, P, s = self.lq_size, self.gt_size, self.scale
        if lq_out is None:
            lq_out = np.empty((self.num_frame, p, p, 3), np.uint8)
        if gt_out is None:
            gt_out = np.empty((P, P, 3), np.uint8)
        for i, f in enumerate(plan.frames):
            lq_out[i] = decode_image(self.client.get('lq', plan.clip, f'im{f}'))[plan.top:plan.top + p, plan.left:plan.left + p]
        gt_out[...] = decode_image(self.client.get('gt', plan.clip, 'im4'))[plan.top * s:plan.top * s + P, plan.left * s:plan.left * s + P]
        return lq_out, gt_out


def make_planner(opt, client=None):
    """The planner of a training dataset option block, chosen like the reference's dataset registry does: by `type`
    (`REDSDataset` - the default - or `Vimeo90KDataset`; options/train/EDVR/*.yml use the former)."""
    kind = opt.get('type', 'REDSDataset')
    if kind == 'REDSDataset':
        return REDSClipPlanner(opt, client)
    if kind == 'Vimeo90KDataset':
        return Vimeo90KClipPlanner(opt, client)
    raise ValueError(f'dataset type {kind} is not supported (REDSDataset, Vimeo90KDataset)')


class EnlargedSampler:
    """Per-rank index order for iteration-based training (data_sampler.py:6-49): a seeded permutation of `ratio` copies of the
    dataset, strided over the ranks - the clip sharding of the data-parallel step (no data-path collective)."""

    def __init__(self, dataset, num_replicas, rank, ratio=1):
        self.dataset_size = len(dataset)
        self.num_replicas, self.rank, self.epoch = num_replicas, rank, 0
        self.num_samples = math.ceil(self.dataset_size * ratio / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        indices = [v % self.dataset_size for v in torch.randperm(self.total_size, generator=g).tolist()]
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


# ------------------------------------------------------------------------------------------------ device side
def frames_to_device(frames_u8, flags=None, device='cuda', stream=None, swap_rb=False):
    """uint8 (n_clips, frames, h, w, 3) host array / tensor (pinned for an asynchronous copy) -> float32 device tensor
    (n_clips, frames, 3, h', w') in [0, 1]; flags: per-clip AUG_* bytes.  Runs on `stream` (default: the current stream)."""
    from . import ops
    t = torch.as_tensor(frames_u8)
    assert t.dtype == torch.uint8 and t.dim() == 5 and t.shape[-1] == 3, f'expected uint8 (n, f, h, w, 3), got {t.dtype} {tuple(t.shape)}'
    if not torch.cuda.is_available():
        raise RuntimeError('edvr_amd.data: no GPU - the conversion runs on the device only (there is no CPU fallback)')
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        dev = t.to(device, non_blocking=True)
        return ops.frames_u8_to_f32(dev, flags, swap_rb=swap_rb)


def read_img_seq(paths, device='cuda', require_mod_crop=False, scale=1, num_threads=8):
    """Validation frames (read_img_seq, data_util.py:11-33): image files -> (t, 3, h, w) RGB float32 in [0, 1] on the device."""
    def one(pth):
        with open(pth, 'rb') as f:
            img = decode_image(f.read())
        if require_mod_crop:
            img = img[:img.shape[0] - img.shape[0] % scale, :img.shape[1] - img.shape[1] % scale]
        return img
    with ThreadPoolExecutor(max(1, min(num_threads, len(paths)))) as pool:
        imgs = list(pool.map(one, [str(p) for p in paths]))
    return frames_to_device(np.stack(imgs)[None], None, device)[0]


class VideoTestClips:
    """VideoTestDataset (basicsr/data/video_test_dataset.py:11-147) with the frames kept on the device.

    Same `opt` keys (dataroot_gt, dataroot_lq, io_backend, cache_data, name, num_frame, padding, optional meta_info_file), same
    `data_info` lists and the same items from __getitem__ ({'lq' (t, c, h, w), 'gt' (c, h, w), 'folder', 'idx', 'border',
    'lq_path'}), except that the tensors live on the GPU: a clip is decoded once on the host (threads), converted by
    edvr_frames_u8_to_f32 and cached there (100 REDS4 frames: 83 MB LQ + 1.1 GB GT in fp32, nothing next to 288 GB), so that
    metrics.validate_clip can batch the windows of a whole clip.  clip(folder) returns the (lq, gt) pair of one folder."""

    def __init__(self, opt, device='cuda'):
        import glob
        import os.path as osp
        from .metrics import generate_frame_indices
        self._indices = generate_frame_indices
        self.opt, self.device = opt, device
        self.cache_data = opt['cache_data']
        self.gt_root, self.lq_root = opt['dataroot_gt'], opt['dataroot_lq']
        assert dict(opt['io_backend'])['type'] != 'lmdb', 'No need to use lmdb during validation/test.'
        self.data_info = {'lq_path': [], 'gt_path': [], 'folder': [], 'idx': [], 'border': []}
        self.imgs_lq, self.imgs_gt, self._cache = {}, {}, {}
        if 'meta_info_file' in opt:
            with open(opt['meta_info_file'], 'r') as fin:
                subfolders = [line.split(' ')[0] for line in fin]
            subfolders_lq = [osp.join(self.lq_root, key) for key in subfolders]
            subfolders_gt = [osp.join(self.gt_root, key) for key in subfolders]
        else:
            subfolders_lq = sorted(glob.glob(osp.join(self.lq_root, '*')))
            subfolders_gt = sorted(glob.glob(osp.join(self.gt_root, '*')))
        if opt['name'].lower() not in ['vid4', 'reds4', 'redsofficial']:
            raise ValueError(f'Non-supported video test dataset: {type(opt["name"])}')

        def files(folder):  # scandir(full_path=True): plain files, no dot files, sorted by the caller
            return sorted(osp.join(folder, e.name) for e in os.scandir(folder) if not e.name.startswith('.') and e.is_file())

        for sub_lq, sub_gt in zip(subfolders_lq, subfolders_gt):
            name = osp.basename(sub_lq)
            paths_lq, paths_gt = files(sub_lq), files(sub_gt)
            max_idx = len(paths_lq)
            assert max_idx == len(paths_gt), f'Different number of images in lq ({max_idx}) and gt folders ({len(paths_gt)})'
            self.data_info['lq_path'].extend(paths_lq)
            self.data_info['gt_path'].extend(paths_gt)
            self.data_info['folder'].extend([name] * max_idx)
            self.data_info['idx'].extend(f'{i}/{max_idx}' for i in range(max_idx))
            border = [0] * max_idx
            for i in range(opt['num_frame'] // 2):
                border[i] = 1
                border[max_idx - i - 1] = 1
            self.data_info['border'].extend(border)
            self.imgs_lq[name], self.imgs_gt[name] = paths_lq, paths_gt

    def __len__(self):
        return len(self.data_info['gt_path'])

    @property
    def folders(self):
        return list(self.imgs_lq)

    def clip(self, folder):
        """(lq (t, 3, h, w), gt (t, 3, H, W)) of one folder on the device."""
        if folder in self._cache:
            return self._cache[folder]
        pair = (read_img_seq(self.imgs_lq[folder], self.device), read_img_seq(self.imgs_gt[folder], self.device))
        if self.cache_data:
            self._cache[folder] = pair
        return pair

    def __getitem__(self, index):
        folder = self.data_info['folder'][index]
        idx, max_idx = (int(v) for v in self.data_info['idx'][index].split('/'))
        select_idx = self._indices(idx, max_idx, self.opt['num_frame'], padding=self.opt['padding'])
        if self.cache_data:
            lq, gt = self.clip(folder)
            imgs_lq = lq.index_select(0, torch.tensor(select_idx, device=lq.device))
            img_gt = gt[idx]
        else:
            imgs_lq = read_img_seq([self.imgs_lq[folder][i] for i in select_idx], self.device)
            img_gt = read_img_seq([self.imgs_gt[folder][idx]], self.device)[0]
        return {'lq': imgs_lq, 'gt': img_gt, 'folder': folder, 'idx': self.data_info['idx'][index],
                'border': self.data_info['border'][index], 'lq_path': self.data_info['lq_path'][index]}


class VideoTestVimeo90KClips:
    """VideoTestVimeo90KDataset (basicsr/data/video_test_dataset.py:150-233; options/test/EDVR/test_EDVR_L_x4_SR_Vimeo90K.yml):
    one item per septuplet listed in `meta_info_file`, the `num_frame` centre frames `im{i}.png` of the sequence as LQ window and
    `im4.png` as GT; same `data_info` and item keys as the reference, tensors decoded on the host and converted on the device."""

    def __init__(self, opt, device='cuda'):
        import os.path as osp
        self.opt, self.device = opt, device
        if opt['cache_data']:
            raise NotImplementedError('cache_data in Vimeo90K-Test dataset is not implemented.')
        assert dict(opt['io_backend'])['type'] != 'lmdb', 'No need to use lmdb during validation/test.'
        self.gt_root, self.lq_root = opt['dataroot_gt'], opt['dataroot_lq']
        self.data_info = {'lq_path': [], 'gt_path': [], 'folder': [], 'idx': [], 'border': []}
        neighbor_list = [i + (9 - opt['num_frame']) // 2 for i in range(opt['num_frame'])]
        with open(opt['meta_info_file'], 'r') as fin:
            subfolders = [line.split(' ')[0] for line in fin]
        for idx, subfolder in enumerate(subfolders):
            self.data_info['gt_path'].append(osp.join(self.gt_root, subfolder, 'im4.png'))
            self.data_info['lq_path'].append([osp.join(self.lq_root, subfolder, f'im{i}.png') for i in neighbor_list])
            self.data_info['folder'].append('vimeo90k')
            self.data_info['idx'].append(f'{idx}/{len(subfolders)}')
            self.data_info['border'].append(0)

    def __len__(self):
        return len(self.data_info['gt_path'])

    def __getitem__(self, index):
        lq_path = self.data_info['lq_path'][index]
        return {'lq': read_img_seq(lq_path, self.device), 'gt': read_img_seq([self.data_info['gt_path'][index]], self.device)[0],
                'folder': self.data_info['folder'][index], 'idx': self.data_info['idx'][index],
                'border': self.data_info['border'][index], 'lq_path': lq_path[self.opt['num_frame'] // 2]}


def epoch_rng(seed, epoch):
    """The random stream of one epoch of one rank: a function of (seed, epoch) only, so an epoch is reproducible however far the
    prefetch of the previous one had run when reset() cut it."""
    return random.Random(seed * 1000003 + epoch)


class REDSDeviceLoader:
    """REDSDataset (or Vimeo90KDataset: `opt['type']`) + DataLoader + EnlargedSampler + CUDAPrefetcher in one object, one per rank.

    next() -> {'lq': (b, t, 3, p, p), 'gt': (b, 3, P, P) float32 device tensors, 'key': [str]} or None at the end of the epoch
    (CUDAPrefetcher.next, prefetch_dataloader.py:118-122); reset() starts the next epoch.  A planner thread makes the random
    decisions serially from epoch_rng(seed + rank, epoch) (deterministic whatever the thread count and prefetch depth), `num_threads` workers decode into pinned staging
    slots, the copy + conversion of batch k+1 run on a side stream while batch k trains."""

    def __init__(self, opt, batch_size, device='cuda', rank=0, world_size=1, ratio=1, seed=0, num_threads=8, depth=3, drop_last=True,
                 client=None):
        if not torch.cuda.is_available():
            raise RuntimeError('REDSDeviceLoader needs a GPU (uint8 staging is converted on the device; there is no CPU fallback)')
        self.planner = make_planner(opt, client)  # REDSDataset (default) or Vimeo90KDataset, by opt['type']
        self.sampler = EnlargedSampler(self.planner, world_size, rank, ratio)
        self.batch_size, self.device, self.drop_last = batch_size, torch.device(device), drop_last
        self.seed = seed + rank  # the reference seeds every worker with seed + rank * workers + id (data/__init__.py)
        self.pool = ThreadPoolExecutor(num_threads)
        t, p, P = self.planner.num_frame, self.planner.lq_size, self.planner.gt_size
        self.slots = [(torch.empty((batch_size, t, p, p, 3), dtype=torch.uint8).pin_memory(),
                       torch.empty((batch_size, 1, P, P, 3), dtype=torch.uint8).pin_memory()) for _ in range(depth)]
        self.stream = torch.cuda.Stream(self.device)
        self.epoch = 0
        self._start()

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    # ---- producer
    def _start(self):
        self.free = queue.Queue()
        for i in range(len(self.slots)):
            self.free.put((i, None))
        self.ready = queue.Queue()
        self.stop = threading.Event()
        self.sampler.set_epoch(self.epoch)
        self.rng = epoch_rng(self.seed, self.epoch)
        if hasattr(self.planner, 'reset_epoch'):
            self.planner.reset_epoch()
        self.thread = threading.Thread(target=self._produce, args=(list(self.sampler), self.stop, self.free, self.ready), daemon=True)
        self.thread.start()
        self.batch = None
        self._preload()

    def _produce(self, order, stop, free, ready):
        try:
            for b0 in range(0, len(order), self.batch_size):
                idx = order[b0:b0 + self.batch_size]
                if len(idx) < self.batch_size and self.drop_last:
                    break
                plans = [self.planner.plan(i, self.rng) for i in idx]
                slot, copied = free.get()
                if stop.is_set():
                    return
                if copied is not None:
                    copied.synchronize()  # the H2D copy that last read this slot
                lq, gt = self.slots[slot]
                lqn, gtn = lq.numpy(), gt.numpy()
                list(self.pool.map(lambda a: self.planner.load(a[1], lqn[a[0]], gtn[a[0], 0]), enumerate(plans)))
                ready.put((slot, plans))
            ready.put(None)
        except BaseException as e:  # surfaces in next()
            ready.put(e)

    # ---- consumer
    def _preload(self):
        from . import ops
        item = self.ready.get()
        if isinstance(item, BaseException):
            raise item
        if item is None:
            self.batch = None
            return
        slot, plans = item
        lq, gt = self.slots[slot]
        n = len(plans)
        flags = bytes(p.flags for p in plans)
        with torch.cuda.stream(self.stream):
            lq_d, gt_d = lq[:n].to(self.device, non_blocking=True), gt[:n].to(self.device, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.stream)
            out = {'lq': ops.frames_u8_to_f32(lq_d, flags), 'gt': ops.frames_u8_to_f32(gt_d, flags)[:, 0], 'key': [p.key for p in plans]}
        self.free.put((slot, copied))
        self.batch = out

    def next(self):
        batch = self.batch
        if batch is None:
            return None
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
        self._preload()
        return batch

    def reset(self, epoch=None):
        """Start the next epoch (CUDAPrefetcher.reset after train_sampler.set_epoch(epoch) in train.py)."""
        self.stop.set()
        self.free.put((0, None))  # wake a producer waiting for a slot
        self.thread.join()
        torch.cuda.synchronize(self.device)
        self.epoch = self.epoch + 1 if epoch is None else epoch
        self._start()

    def close(self):
        self.stop.set()
        self.free.put((0, None))
        self.thread.join()
        self.pool.shutdown()
