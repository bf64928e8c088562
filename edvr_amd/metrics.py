"""Validation metrics on the device.

Mirrors xinntao/EDVR (BasicSR v1.2.0): tensor2img (basicsr/utils/img_util.py:36-98) followed by calculate_psnr
(basicsr/metrics/psnr_ssim.py:7-51), which VideoBaseModel.dist_validation (video_base_model.py:60-98) runs per frame in NumPy
after copying the frame to the host.  Here the clamp-round-uint8 conversion and the squared differences happen in one HIP
kernel (csrc/metrics.hip) and only one double per image leaves the GPU.  SSIM (psnr_ssim.py:54-141) is not implemented.
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
