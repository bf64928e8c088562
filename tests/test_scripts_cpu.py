"""CPU: control flow of scripts/test_reds.py (clip sharding, per-folder and overall averages) with CPU stand-ins for the GPU-only
pieces (network, frame conversion, device PSNR) - the pieces themselves are covered by the -m gpu tests."""
import argparse
import importlib.util
import os

import torch

from util_data import write_video_test_tree


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(__file__), '..', 'scripts', f'{name}.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_evaluation_script_flow(tmp_path, monkeypatch):
    import edvr_amd
    from edvr_amd import data as D, metrics as M
    spec = dict(folders=['000', '011', '015'], frames=6, lq_hw=(8, 12), scale=4)
    write_video_test_tree(str(tmp_path), spec)

    class Net(torch.nn.Module):  # "restores" by bilinear x4 of the centre frame
        def __init__(self, *a, **k):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            return torch.nn.functional.interpolate(x[:, x.shape[1] // 2], scale_factor=4, mode='bilinear', align_corners=False)

        def to(self, device):
            return self

    def read_img_seq(paths, device='cpu', **k):
        return torch.stack([torch.from_numpy(D.decode_image(open(p, 'rb').read()).transpose(2, 0, 1).copy()).float() / 255 for p in paths])

    def psnr(a, b, crop_border=0, test_y_channel=False):
        return [float(10 * torch.log10(1 / ((x - t) ** 2).mean())) for x, t in zip(a, b)]

    monkeypatch.setattr(edvr_amd, 'EDVR', Net)
    monkeypatch.setattr(D, 'read_img_seq', read_img_seq)
    monkeypatch.setattr(M, 'calculate_psnr', psnr)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    tr = _load('test_reds')
    args = argparse.Namespace(lq=str(tmp_path / 'lq'), gt=str(tmp_path / 'gt'), weights=None, name='REDS4', vimeo_meta=None, num_feat=64,
                              num_reconstruct_block=2, num_frame=5, hr_in=False, with_predeblur=False, no_tsa=False,
                              padding='reflection_circle', crop_border=0, test_y_channel=False, batch=4)
    lines = []
    summary = tr.evaluate(args, log=lines.append)
    assert list(summary) == spec['folders'] and all(5 < v < 60 for v in summary.values())
    assert lines[-1].startswith('average over 3 folder(s)')
    avg = sum(summary.values()) / 3
    assert abs(float(lines[-1].split(': ')[1].split()[0]) - avg) < 1e-3
    # per-folder value = mean of the per-frame PSNRs of that clip
    lq = read_img_seq(sorted(str(p) for p in (tmp_path / 'lq' / '011').glob('*.png')))
    gt = read_img_seq(sorted(str(p) for p in (tmp_path / 'gt' / '011').glob('*.png')))
    _, per_frame = M.validate_clip(Net(), lq, gt, num_frame=5, padding='reflection_circle', batch=4)
    assert abs(sum(per_frame) / len(per_frame) - summary['011']) < 1e-6
