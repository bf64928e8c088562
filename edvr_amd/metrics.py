"""Validation metrics on the device.

Mirrors xinntao/EDVR (BasicSR v1.2.0): tensor2img (basicsr/utils/img_util.py:36-98) followed by calculate_psnr
(basicsr/metrics/psnr_ssim.py:7-51), which VideoBaseModel.dist_validation (video_base_model.py:60-98) runs per frame in NumPy
after copying the frame to the host.  Here the clamp-round-uint8 conversion and the squared differences happen in one HIP
kernel (csrc/metrics.hip) and only one double per image leaves the GPU.  calculate_ssim (psnr_ssim.py:54-141) likewise: the
11x11 Gaussian moments and the SSIM map are computed in float64 on the device, a few partial sums per image leave it.

validate_clip() is the batched form of that loop for one clip: every frame's window of `num_frame` neighbours is built with
generate_frame_indices (basicsr/data/data_util.py:35-88, the padding modes of the REDS4 / Vid4 test sets), windows run through
the network `batch` at a time, and each output frame is scored on the device.
"""
import math

import torch

from . import _lib, ops


def sum_squared_error_uint8(pred, gt, crop_border=0, test_y_channel=False):
    """Per-image sum of squared differences of tensor2img(pred) and tensor2img(gt) inside the crop, and the element count.
    pred, gt: (n, c, h, w) fp32 CUDA tensors in RGB channel order (c = 3 or 1), values nominally in [0, 1] (clamped)."""
    ops.require_gpu(pred, gt)
    if pred.shape != gt.shape:
        raise AssertionError(f'Image shapes are differnet: {tuple(pred.shape)}, {tuple(gt.shape)}.')  # psnr_ssim.py:30-31
    if pred.dim() == 3:
        pred, gt = pred[None], gt[None]
    pred, gt = pred.contiguous(), gt.contiguous()
    n, c, h, w = pred.shape
    blocks = max(1, min(256, (h * w + 4095) // 4096))
    partial = torch.empty(n, blocks, dtype=torch.float64, device=pred.device)
    y = 1 if (test_y_channel and c == 3) else 0
    _lib.check(_lib.lib().edvr_psnr_sse_f32(pred.data_ptr(), gt.data_ptr(), partial.data_ptr(), n, c, h, w, c * h * w, c * h * w,
                                            int(crop_border), y, blocks, torch.cuda.current_stream(pred.device).cuda_stream), 'edvr_psnr_sse_f32')
    count = (h - 2 * crop_border) * (w - 2 * crop_border) * (1 if y else c)
    return partial.sum(1), count


def calculate_psnr(pred, gt, crop_border=0, test_y_channel=False):
    """calculate_psnr(tensor2img(pred), tensor2img(gt), crop_border, test_y_channel=...) of the reference for every image of
    the batch, as a list of floats (inf where the images are identical)."""
    sse, count = sum_squared_error_uint8(pred, gt, crop_border, test_y_channel)
    out = []
    for s in sse.cpu().tolist():
        mse = s / count
        out.append(float('inf') if mse == 0 else 20.0 * math.log10(255.0 / math.sqrt(mse)))
    return out


def calculate_ssim(pred, gt, crop_border=0, test_y_channel=False):
    """calculate_ssim(tensor2img(pred), tensor2img(gt), crop_border, test_y_channel=...) of the reference (psnr_ssim.py:90-141)
    for every image of the batch, as a list of floats.  pred, gt: (n, c, h, w) or (c, h, w) fp32 CUDA tensors, RGB, c = 3 or 1."""
    ops.require_gpu(pred, gt)
    if pred.shape != gt.shape:
        raise AssertionError(f'Image shapes are differnet: {tuple(pred.shape)}, {tuple(gt.shape)}.')  # psnr_ssim.py:118-119
    if pred.dim() == 3:
        pred, gt = pred[None], gt[None]
    pred, gt = pred.contiguous(), gt.contiguous()
    n, c, h, w = pred.shape
    L = _lib.lib()
    tiles = L.edvr_ssim_partials(h, w, int(crop_border))
    if tiles == 0:
        raise ValueError(f'calculate_ssim: {h}x{w} with crop_border {crop_border} leaves no 11x11 window')
    y = 1 if (test_y_channel and c == 3) else 0
    chans = 1 if y else c
    partial = torch.empty(n, chans, tiles, dtype=torch.float64, device=pred.device)
    _lib.check(L.edvr_ssim_f32(pred.data_ptr(), gt.data_ptr(), partial.data_ptr(), n, c, h, w, c * h * w, c * h * w, int(crop_border), y,
                               torch.cuda.current_stream(pred.device).cuda_stream), 'edvr_ssim_f32')
    count = (h - 2 * crop_border - 10) * (w - 2 * crop_border - 10)
    return (partial.sum(2) / count).mean(1).cpu().tolist()


def generate_frame_indices(crt_idx, max_frame_num, num_frames, padding='reflection'):
    """basicsr/data/data_util.py:35-88: indices of the `num_frames` frames around `crt_idx` in a sequence of `max_frame_num`
    frames, out-of-range positions padded by 'replicate' | 'reflection' | 'reflection_circle' | 'circle'."""
    assert num_frames % 2 == 1, 'num_frames should be an odd number.'
    assert padding in ('replicate', 'reflection', 'reflection_circle', 'circle'), f'Wrong padding mode: {padding}.'
    max_frame_num = max_frame_num - 1  # start from 0
    num_pad = num_frames // 2
    indices = []
    for i in range(crt_idx - num_pad, crt_idx + num_pad + 1):
        if i < 0:
            pad_idx = {'replicate': 0, 'reflection': -i, 'reflection_circle': crt_idx + num_pad - i, 'circle': num_frames + i}[padding]
        elif i > max_frame_num:
            pad_idx = {'replicate': max_frame_num, 'reflection': max_frame_num * 2 - i,
                       'reflection_circle': (crt_idx - num_pad) - (i - max_frame_num), 'circle': i - num_frames}[padding]
        else:
            pad_idx = i
        indices.append(pad_idx)
    return indices


@torch.no_grad()
def validate_clip(net, lq, gt=None, num_frame=5, padding='reflection_circle', batch=4, crop_border=0, test_y_channel=False):
    """Restore every frame of one clip and (if `gt` is given) score it: what VideoBaseModel.dist_validation
    (video_base_model.py:44-98) does frame by frame with a host round trip per frame.
    lq: (t, c, h, w) low-quality frames on the GPU, gt: (t, c, H, W).  Returns (outputs (t, c, H, W), [PSNR per frame] or None)."""
    t = lq.shape[0]
    outs, scores = [], []
    for s0 in range(0, t, batch):
        idx = [generate_frame_indices(i, t, num_frame, padding) for i in range(s0, min(s0 + batch, t))]
        windows = lq[torch.tensor(idx, device=lq.device)]  # (b, num_frame, c, h, w)
        out = net(windows)
        outs.append(out)
        if gt is not None:
            scores += calculate_psnr(out, gt[s0:s0 + out.shape[0]], crop_border, test_y_channel)
    return torch.cat(outs, 0), (scores if gt is not None else None)
