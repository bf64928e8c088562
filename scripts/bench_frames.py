"""edvr_frames_u8_to_f32 against its HBM roofline (3 B read + 12 B written per pixel): a training batch (32 clips x (5 LQ 64x64 +
1 GT 256x256), mixed augmentation states) and a validation clip (100 frames of 720x1280, no augmentation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from edvr_amd import ops  # noqa: E402


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    dev = torch.device('cuda:0')
    for name, shape, flags in (('train LQ  32x5x64x64', (32, 5, 64, 64), [i % 8 for i in range(32)]),
                               ('train GT  32x1x256x256', (32, 1, 256, 256), [i % 8 for i in range(32)]),
                               ('train GT  32x1x256x256 (all transposed)', (32, 1, 256, 256), [4 + (i % 4) for i in range(32)]),
                               ('val GT 100x720x1280', (1, 100, 720, 1280), None)):
        x = torch.randint(0, 256, shape + (3,), dtype=torch.uint8, device=dev)
        t = timed(lambda: ops.frames_u8_to_f32(x, flags))
        nbytes = x.numel() * 5  # 1 B in + 4 B out per sample
        print(f'{name:42s} {t * 1e6:9.1f} us  {nbytes / t / 1e9:8.1f} GB/s (algorithmic 15 B/pixel; HBM peak 8000, achievable ~6300)', flush=True)


if __name__ == '__main__':
    main()
