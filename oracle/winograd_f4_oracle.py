"""TEST INFRASTRUCTURE ONLY - fp64 restatement of the Winograd F(4x4, 3x3) convolution that csrc/winograd_f4.hip runs, and
of the operand order its weights are packed in.

The reference has no Winograd code of its own: its 3x3 / stride-1 ``nn.Conv2d`` layers (edvr_arch.py:24-71, 190-244, 322-352,
arch_util.py:86-95) go to cuDNN, which picks a Winograd algorithm for them under ``torch.backends.cudnn.benchmark = True``
(basicsr/train.py:132).  What every such algorithm must equal is the convolution itself, so this oracle is pinned two ways:
``conv_f4`` (transform matrices of Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks", F(4x4, 3x3)) against
``torch.nn.functional.conv2d`` in float64 (tests/test_oracle_winograd_f4.py, CPU), and the device-packed weights against
``pack_operand_order`` (tests/test_gpu_conv_f4.py).

Only tests/ may import this module.  edvr_amd/ never does.
"""
import torch

# B^T (6x6), G (6x3), A^T (4x6): interpolation points 0, +-1, +-2, infinity
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                  dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                 dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def weights_f4(w, transpose_flip=False):
    """U = G g G^T per (co, ci): (co, ci, 3, 3) -> (co, ci, 6, 6) in float64.  transpose_flip: the data-gradient kernel
    w'[c][o][i][j] = w[o][c][2-i][2-j] (edvr_conv2d_pack_weight_f32's convention)."""
    w = w.double()
    if transpose_flip:
        w = w.flip(2, 3).transpose(0, 1)
    return torch.einsum('ri,ocij,sj->ocrs', G, w, G)


def conv_f4(x, w, bias=None):
    """3x3 / stride 1 / pad 1 cross-correlation of x (n, ci, h, w) with w (co, ci, 3, 3) by F(4x4, 3x3) tiles, float64."""
    x, U = x.double(), weights_f4(w)
    n, ci, h, wd = x.shape
    co = w.shape[0]
    th, tw = -(-h // 4), -(-wd // 4)
    xp = torch.zeros(n, ci, 4 * th + 2, 4 * tw + 2, dtype=torch.float64)
    xp[:, :, 1:h + 1, 1:wd + 1] = x
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                      # (n, ci, th, tw, 6, 6) patches, stride 4
    V = torch.einsum('ri,nctuij,sj->ncturs', BT, d, BT)         # B^T d B
    M = torch.einsum('ocrs,ncturs->noturs', U, V)               # 36 GEMMs over the input channels
    Y = torch.einsum('ir,noturs,js->notuij', AT, M, AT)         # A^T M A: (n, co, th, tw, 4, 4)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, co, 4 * th, 4 * tw)[:, :, :h, :wd]
    return y if bias is None else y + bias.double().view(1, -1, 1, 1)


def pack_operand_order(w, transpose_flip=False):
    """The buffer edvr_conv2d_pack_weight_f4_f32 writes, as float64: [co block 64][channel pair][row 6][co half 2] blocks of 384
    numbers = [lane 64][4] (positions (row, 0..3)) followed by [lane 64][2] (positions (row, 4..5)); lane = (channel parity, co
    within the half).  Channels are padded to a multiple of 8, output channels to a multiple of 64 (zeros)."""
    U = weights_f4(w, transpose_flip)
    co, ci = U.shape[:2]
    cop, cip = -(-co // 64) * 64, -(-ci // 8) * 8
    Up = torch.zeros(cop, cip, 6, 6, dtype=torch.float64)
    Up[:co, :ci] = U
    # (cb, wm, j) <- o ; (cpair, parity) <- c
    Up = Up.view(cop // 64, 2, 32, cip // 2, 2, 6, 6)            # cb, wm, j, cpair, par, r, c
    blk = Up.permute(0, 3, 5, 1, 4, 2, 6)                        # cb, cpair, r, wm, par, j, c
    a4 = blk[..., :4].reshape(cop // 64, cip // 2, 6, 2, 64 * 4)
    a2 = blk[..., 4:].reshape(cop // 64, cip // 2, 6, 2, 64 * 2)
    return torch.cat([a4, a2], -1).reshape(-1)
