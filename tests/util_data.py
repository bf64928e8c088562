"""Shared helpers of the input-pipeline tests: synthetic REDS-shaped storage (in memory or PNG folders)."""
import io
import os

import numpy as np

from oracle import data_oracle as DO


def png_bytes(bgr_u8):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr_u8[:, :, ::-1])).save(buf, format='PNG', compress_level=1)
    return buf.getvalue()


class SyntheticClient:
    """Storage serving PNG encodings of oracle.data_oracle.synthetic_frame - the frames the golden generator fed to the reference."""

    def __init__(self, lq_hw, scale):
        self.lq_hw, self.scale = lq_hw, scale

    def size(self, kind, clip, frame):
        return self.lq_hw if kind == 'lq' else (self.lq_hw[0] * self.scale, self.lq_hw[1] * self.scale)

    def get(self, kind, clip, frame):
        return png_bytes(DO.synthetic_frame(kind, clip, frame, *self.size(kind, clip, frame)))


def fetch_bgr(lq_hw, scale):
    """The oracle's `fetch` for the same storage: decoded BGR frames as cv2.imdecode would return them."""
    def fetch(kind, clip, frame):
        h, w = lq_hw if kind == 'lq' else (lq_hw[0] * scale, lq_hw[1] * scale)
        return DO.synthetic_frame(kind, clip, frame, h, w)
    return fetch


def write_png_dataset(root, clips, lq_hw, scale, frames=100):
    """<root>/{lq,gt}/<clip>/<frame:08d>.png + meta_info.txt in the reference's layout (docs/DatasetPreparation.md)."""
    for kind in ('lq', 'gt'):
        h, w = lq_hw if kind == 'lq' else (lq_hw[0] * scale, lq_hw[1] * scale)
        for clip in clips:
            d = os.path.join(root, kind, clip)
            os.makedirs(d, exist_ok=True)
            for f in range(frames):
                with open(os.path.join(d, f'{f:08d}.png'), 'wb') as fh:
                    fh.write(png_bytes(DO.synthetic_frame(kind, clip, f'{f:08d}', h, w)))
    meta = os.path.join(root, 'meta_info.txt')
    with open(meta, 'w') as fh:
        fh.writelines(f'{clip} {frames} ({lq_hw[0] * scale},{lq_hw[1] * scale},3)\n' for clip in clips)
    return meta


def base_opt(root=None, meta=None, **kw):
    opt = dict(dataroot_gt=os.path.join(root, 'gt') if root else '/nonexistent/gt', dataroot_lq=os.path.join(root, 'lq') if root else '/nonexistent/lq',
               dataroot_flow=None, meta_info_file=meta, io_backend=dict(type='disk'), gt_size=32, scale=4, num_frame=5, interval_list=[1],
               random_reverse=False, use_flip=True, use_rot=True, val_partition='REDS4')
    opt.update(kw)
    return opt


def write_video_test_tree(root, spec):
    """The PNG tree oracle/make_golden.py::video_test_case fed to the reference's VideoTestDataset."""
    for kind in ('lq', 'gt'):
        h, w = spec['lq_hw'] if kind == 'lq' else (spec['lq_hw'][0] * spec['scale'], spec['lq_hw'][1] * spec['scale'])
        for folder in spec['folders']:
            d = os.path.join(root, kind, folder)
            os.makedirs(d, exist_ok=True)
            for f in range(spec['frames']):
                with open(os.path.join(d, f'{f:08d}.png'), 'wb') as fh:
                    fh.write(png_bytes(DO.synthetic_frame(kind, folder, f'{f:08d}', h, w)))
            open(os.path.join(d, '.hidden'), 'w').close()


def video_test_opt(root, run):
    return dict(name='REDS4', dataroot_gt=os.path.join(root, 'gt'), dataroot_lq=os.path.join(root, 'lq'), io_backend=dict(type='disk'),
                cache_data=run['cache_data'], num_frame=run['num_frame'], padding=run['padding'])


def write_vimeo_train_tree(root, keys, lq_hw, scale):
    """<root>/{lq,gt}/<clip>/<seq>/im1.png .. im7.png + the meta file of the Vimeo90K training set (docs/DatasetPreparation.md)."""
    for kind in ('lq', 'gt'):
        h, w = lq_hw if kind == 'lq' else (lq_hw[0] * scale, lq_hw[1] * scale)
        for key in keys:
            d = os.path.join(root, kind, *key.split('/'))
            os.makedirs(d, exist_ok=True)
            for n in range(1, 8):
                with open(os.path.join(d, f'im{n}.png'), 'wb') as fh:
                    fh.write(png_bytes(DO.synthetic_frame(kind, key, f'im{n}', h, w)))
    meta = os.path.join(root, 'meta_info_Vimeo90K_train_GT.txt')
    with open(meta, 'w') as fh:
        fh.writelines(f'{key} 7 ({lq_hw[0] * scale},{lq_hw[1] * scale},3)\n' for key in keys)
    return meta
