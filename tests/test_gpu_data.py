"""-m gpu: the device side of the input pipeline (edvr_frames_u8_to_f32 through edvr_amd.ops / edvr_amd.data), bit-exact against
the oracle and against the reference's own outputs (tests/golden/data_pipeline.pt)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data_oracle as DO
from util_data import SyntheticClient, base_opt, fetch_bgr, png_bytes, video_test_opt, write_png_dataset, write_video_test_tree

pytestmark = pytest.mark.gpu
GOLD = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'data_pipeline.pt'))


def _ref(frames_u8, flags, swap_rb):
    """numpy: (n, f, h, w, 3) uint8 -> (n, f, 3, h', w') float32 the reference's way (float first, then permute)."""
    out = []
    for c, clip in enumerate(frames_u8):
        row = []
        for img in clip:
            x = DO.imfrombytes_float(img)
            x = DO.apply_aug_flags(x, flags[c] if flags is not None else 0)
            row.append(DO.img2tensor(x) if swap_rb else torch.from_numpy(x.transpose(2, 0, 1).copy()))
        out.append(torch.stack(row, 0))
    return torch.stack(out, 0)


@pytest.mark.parametrize('shape', [(8, 5, 8, 8), (8, 1, 32, 32), (9, 2, 45, 45), (8, 5, 64, 64), (8, 1, 256, 256), (300, 1, 7, 7)])
@pytest.mark.parametrize('swap_rb', [False, True])
def test_frames_kernel_all_augmentations(gpu, shape, swap_rb):
    from edvr_amd import ops
    n, f, h, w = shape
    rs = np.random.RandomState(n * 1000 + h)
    x = rs.randint(0, 256, (n, f, h, w, 3)).astype(np.uint8)
    for flags in ([i % 8 for i in range(n)], [4] * n, None):  # mixed states incl. transposed ones (square: shapes agree)
        y = ops.frames_u8_to_f32(torch.from_numpy(x).to(gpu), flags, swap_rb=swap_rb)
        assert torch.equal(y.cpu(), _ref(x, flags, swap_rb))


@pytest.mark.parametrize('hw', [(18, 30), (33, 97), (180, 320), (1, 5)])
def test_frames_kernel_rectangular(gpu, hw):
    from edvr_amd import ops
    h, w = hw
    x = np.random.RandomState(h).randint(0, 256, (3, 2, h, w, 3)).astype(np.uint8)
    for flags in (None, [0, 1, 2], [3, 3, 1]):
        y = ops.frames_u8_to_f32(torch.from_numpy(x).to(gpu), flags)
        assert y.shape == (3, 2, 3, h, w) and torch.equal(y.cpu(), _ref(x, flags, False))
    with pytest.raises(RuntimeError, match='square'):
        ops.frames_u8_to_f32(torch.from_numpy(x).to(gpu), [0, 4, 0])
    y = ops.frames_u8_to_f32(torch.from_numpy(x[:, :, :, :h]).to(gpu).contiguous(), [4, 5, 6]) if h <= w else None
    if y is not None:
        assert torch.equal(y.cpu(), _ref(np.ascontiguousarray(x[:, :, :, :h]), [4, 5, 6], False))


def test_division_is_numpy_division(gpu):
    """All 256 byte values: the kernel's float32 quotient equals the reference's `img.astype(np.float32) / 255.` bit for bit."""
    from edvr_amd import ops
    ramp = torch.arange(256, dtype=torch.uint8).reshape(1, 1, 1, 256, 1).repeat(1, 1, 1, 1, 3)
    y = ops.frames_u8_to_f32(ramp.to(gpu))
    assert torch.equal(y[0, 0, 0, 0].cpu(), GOLD['div255'])


def test_reference_samples_through_the_device(gpu, tmp_path):
    """The 40 samples the reference's REDSDataset produced: planner + byte crop + device conversion give the same tensors."""
    from edvr_amd import data as D
    meta = tmp_path / 'meta.txt'
    meta.write_text(''.join(GOLD['meta']))
    client = SyntheticClient(GOLD['lq_hw'], GOLD['scale'])
    planners = {}
    for s in GOLD['samples']:
        pl = planners.setdefault(s['variant'], D.REDSClipPlanner(base_opt(meta=str(meta), **s['opt']), client))
        plan = pl.plan(s['index'], random.Random(s['seed']))
        lq, gt = pl.load(plan)
        lq_d = D.frames_to_device(lq[None], [plan.flags], gpu)[0]
        gt_d = D.frames_to_device(gt[None, None], [plan.flags], gpu)[0, 0]
        assert torch.equal(lq_d.cpu(), s['lq_u8'].float() / 255.) and torch.equal(gt_d.cpu(), s['gt_u8'].float() / 255.)


@pytest.mark.parametrize('world', [(0, 1), (1, 2)])
def test_device_loader_epochs(gpu, tmp_path, world):
    """REDSDeviceLoader (sampler + planner + threaded decode + pinned staging + side stream + kernel) against the oracle run
    over the same index order and random stream, two epochs."""
    from edvr_amd import data as D
    rank, ws = world
    lq_hw, scale, bs = (12, 20), 4, 4
    meta = write_png_dataset(str(tmp_path), ['001', '002'], lq_hw, scale, frames=100)
    opt = base_opt(root=str(tmp_path), meta=meta, gt_size=32, num_frame=5, interval_list=[1, 2], random_reverse=True)
    loader = D.REDSDeviceLoader(opt, bs, device=gpu, rank=rank, world_size=ws, seed=10, num_threads=4, depth=2)
    fetch = fetch_bgr(lq_hw, scale)
    keys = DO.reds_keys(open(meta).readlines(), 'REDS4')
    assert len(loader) == (200 // ws) // bs
    for epoch in (0, 1):
        order = DO.enlarged_sampler_indices(200, ws, rank, 1, epoch)
        rng = D.epoch_rng(10 + rank, epoch)
        nb = 0
        while True:
            batch = loader.next()
            if batch is None:
                break
            ref = [DO.reds_getitem(keys, opt, i, rng, fetch) for i in order[nb * bs:(nb + 1) * bs]]
            assert batch['key'] == [r['key'] for r in ref]
            assert batch['lq'].shape == (bs, 5, 3, 8, 8) and batch['gt'].shape == (bs, 3, 32, 32)
            assert torch.equal(batch['lq'].cpu(), torch.stack([r['lq'] for r in ref]))
            assert torch.equal(batch['gt'].cpu(), torch.stack([r['gt'] for r in ref]))
            nb += 1
            if nb == 5 and epoch == 0:
                break  # leave the first epoch early (train.py stops at total_iter): reset() has to drain the producer
        if epoch == 0:
            loader.reset()
        else:
            assert nb == len(loader)
    loader.close()


def test_read_img_seq(gpu, tmp_path):
    from edvr_amd import data as D
    paths = []
    for i in range(2):
        p = tmp_path / f'{i:08d}.png'
        p.write_bytes(png_bytes(DO.synthetic_frame('lq', '000', f'{i:08d}', 18, 30)))
        paths.append(str(p))
    assert torch.equal(D.read_img_seq(paths, gpu).cpu(), GOLD['read_img_seq'])
    y = D.read_img_seq(paths, gpu, require_mod_crop=True, scale=4)
    assert torch.equal(y.cpu(), GOLD['read_img_seq'][:, :, :16, :28])


def test_video_test_clips_items_equal_reference(gpu, tmp_path):
    """Items of VideoTestClips (frames decoded on the host, converted + cached on the device) == the reference's VideoTestDataset."""
    from edvr_amd import data as D
    gold = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'video_test.pt'))
    write_video_test_tree(str(tmp_path), gold['spec'])
    for run in gold['runs']:
        ds = D.VideoTestClips(video_test_opt(str(tmp_path), run), device=gpu)
        for ref in run['items']:
            it = ds[ref['index']]
            assert it['lq'].is_cuda and torch.equal(it['lq'].cpu(), ref['lq_u8'].float() / 255.)
            assert torch.equal(it['gt'].cpu(), ref['gt_u8'].float() / 255.)
            assert (it['folder'], it['idx'], it['border'], os.path.relpath(it['lq_path'], str(tmp_path))) == \
                   (ref['folder'], ref['idx'], ref['border'], ref['lq_path'])
        lq, gt = ds.clip('011')
        assert lq.shape == (7, 3, 10, 14) and gt.shape == (7, 3, 40, 56)
        assert (ds.clip('011')[0] is lq) == run['cache_data']


def test_validate_clip_from_png_folders(gpu, tmp_path):
    """Folder of PNGs -> VideoTestClips -> validate_clip (batched windows, device PSNR): the same frames and scores as the
    reference's per-frame loop over dataset items."""
    from edvr_amd import data as D, metrics
    from util_edvr import build
    gold = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'video_test.pt'))
    spec = dict(gold['spec'], lq_hw=(16, 20))
    write_video_test_tree(str(tmp_path), spec)
    run = gold['runs'][0]
    ds = D.VideoTestClips(video_test_opt(str(tmp_path), run), device=gpu)
    net, _, _ = build('M_T5')
    net = net.to(gpu).eval()
    lq, gt = ds.clip('000')
    outs, scores = metrics.validate_clip(net, lq, gt, num_frame=5, padding=run['padding'], batch=3)
    for i in (0, 3, 6):
        it = ds[i]
        with torch.no_grad():
            o = net(it['lq'][None])
        assert (o[0] - outs[i]).abs().max().item() < 1e-5
        assert abs(metrics.calculate_psnr(o, it['gt'][None])[0] - scores[i]) < 1e-2


def test_training_loop_from_png_folders(gpu, tmp_path):
    """scripts/train_reds.py on a tiny REDS-shaped PNG tree: loader -> EDVR -> Charbonnier -> FusedAdam -> scheduler, TSA warm-up,
    validation from folders, checkpoint + resume.  A smoke test of the pieces working together (values are covered elsewhere)."""
    import argparse
    import importlib.util
    import math
    spec = importlib.util.spec_from_file_location('train_reds', os.path.join(os.path.dirname(__file__), '..', 'scripts', 'train_reds.py'))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    root = str(tmp_path)
    meta = write_png_dataset(root, ['001', '002'], (20, 24), 4, frames=100)
    vt = dict(folders=['000'], frames=6, lq_hw=(16, 20), scale=4)
    write_video_test_tree(os.path.join(root, 'val'), vt)
    args = argparse.Namespace(gt=os.path.join(root, 'gt'), lq=os.path.join(root, 'lq'), meta=meta, val_gt=os.path.join(root, 'val', 'gt'),
                              val_lq=os.path.join(root, 'val', 'lq'), val_partition='REDS4', num_feat=64, num_reconstruct_block=2, num_frame=5,
                              gt_size=64, batch=2, threads=4, enlarge_ratio=1, iters=4, lr=4e-4, dcn_lr_mul=0.25, periods=[4, 4],
                              restart_weights=[1, 0.5], tsa_iter=3, print_freq=1, save_freq=4, val_freq=4, val_batch=3,
                              save_dir=os.path.join(root, 'ckpt'), resume=None, pretrain=None, seed=10)
    lines = []
    losses = tr.train(args, log=lines.append)
    assert len(losses) == 4 and all(math.isfinite(v) and v > 0 for v in losses)
    assert any('validation PSNR' in ln for ln in lines) and sum('TSA schedule' in ln for ln in lines) == 2
    assert os.path.exists(os.path.join(root, 'ckpt', 'net_g_4.pth')) and os.path.exists(os.path.join(root, 'ckpt', '4.state'))
    args.resume, args.pretrain, args.iters, args.save_dir = os.path.join(root, 'ckpt', '4.state'), os.path.join(root, 'ckpt', 'net_g_4.pth'), 6, None
    more = tr.train(args, log=lines.append)
    assert len(more) == 2 and all(math.isfinite(v) for v in more)


def test_vimeo90k_training_loader(gpu, tmp_path):
    """The same device loader on the Vimeo90K TRAINING set (`type: Vimeo90KDataset`, basicsr/data/vimeo90k_dataset.py): PNG tree ->
    planner (the reference's draws, its in-place reversal of the neighbour list) -> threaded decode -> device conversion, against the
    oracle restatement (pinned on the reference's own class: tests/golden/vimeo90k_train.pt) over the same order and random stream."""
    from edvr_amd import data as D
    from util_data import write_vimeo_train_tree
    lq_hw, scale, bs = (12, 20), 4, 3
    keys = [f'{c:05d}/{q:04d}' for c in (1, 2) for q in (1, 2, 3, 4, 5, 6)]
    meta = write_vimeo_train_tree(str(tmp_path), keys, lq_hw, scale)
    opt = dict(type='Vimeo90KDataset', dataroot_gt=str(tmp_path / 'gt'), dataroot_lq=str(tmp_path / 'lq'), meta_info_file=meta,
               io_backend=dict(type='disk'), num_frame=5, gt_size=32, scale=scale, random_reverse=True, use_flip=True, use_rot=True)
    loader = D.REDSDeviceLoader(opt, bs, device=gpu, rank=0, world_size=1, seed=3, num_threads=4, depth=2)
    fetch = fetch_bgr(lq_hw, scale)
    assert len(loader) == len(keys) // bs
    for epoch in (0, 1):
        order = DO.enlarged_sampler_indices(len(keys), 1, 0, 1, epoch)
        rng = D.epoch_rng(3, epoch)
        nb_list = DO.vimeo90k_neighbor_list(5)
        nb = 0
        while True:
            batch = loader.next()
            if batch is None:
                break
            ref = [DO.vimeo90k_getitem(keys, opt, i, rng, fetch, nb_list) for i in order[nb * bs:(nb + 1) * bs]]
            assert batch['key'] == [r['key'] for r in ref]
            assert batch['lq'].shape == (bs, 5, 3, 8, 8) and batch['gt'].shape == (bs, 3, 32, 32)
            assert torch.equal(batch['lq'].cpu(), torch.stack([r['lq'] for r in ref]))
            assert torch.equal(batch['gt'].cpu(), torch.stack([r['gt'] for r in ref]))
            nb += 1
        assert nb == len(loader)
        loader.reset()
    loader.close()
