"""CPU: the fp64 restatement of Winograd F(4x4, 3x3) (oracle/winograd_f4_oracle.py) equals the convolution it replaces, and its
operand-order packing addresses every weight exactly once."""
import torch
import torch.nn.functional as F

from oracle import winograd_f4_oracle as W


def test_f4_restatement_equals_the_convolution():
    g = torch.Generator().manual_seed(0)
    for n, ci, h, w, co in [(1, 3, 8, 8, 4), (2, 5, 9, 14, 7), (1, 8, 4, 4, 8), (1, 2, 13, 6, 3)]:
        x = torch.randn(n, ci, h, w, generator=g, dtype=torch.float64)
        wt = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
        b = torch.randn(co, generator=g, dtype=torch.float64)
        ref = F.conv2d(x, wt, b, 1, 1)
        got = W.conv_f4(x, wt, b)
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() < 1e-11 * ref.abs().max().item()


def test_transposed_flipped_weights_give_the_data_gradient():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 7, 9, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(6, 4, 3, 3, generator=g, dtype=torch.float64)
    dy = torch.randn(1, 6, 7, 9, generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, 1, 1).backward(dy)
    U = W.weights_f4(wt, transpose_flip=True)                     # (ci, co, 6, 6): the kernel of the data-gradient convolution
    w_dgrad = wt.flip(2, 3).transpose(0, 1)
    assert torch.equal(U, W.weights_f4(w_dgrad))
    assert (W.conv_f4(dy, w_dgrad) - x.grad).abs().max().item() < 1e-11 * x.grad.abs().max().item()


def test_operand_order_is_a_permutation_of_the_padded_weights():
    g = torch.Generator().manual_seed(2)
    co, ci = 70, 20                                               # padded to 128 x 24
    wt = torch.randn(co, ci, 3, 3, generator=g, dtype=torch.float64)
    packed = W.pack_operand_order(wt)
    assert packed.numel() == 128 * 24 * 36
    U = W.weights_f4(wt)
    assert torch.equal(torch.sort(packed[packed != 0]).values, torch.sort(U.reshape(-1)).values)
    # spot-check the address of one element: o = 69 (block 1, half 0, j = 5), c = 19 (pair 9, parity 1), position (r = 2, col = 4)
    o, c, r, col = 69, 19, 2, 4
    blk = (((o // 64) * (24 // 2) + c // 2) * 6 + r) * 2 + (o % 64) // 32
    lane = (c % 2) * 32 + o % 32
    assert packed[blk * 384 + 256 + lane * 2 + (col - 4)] == U[o, c, r, col]
    col = 1
    assert packed[blk * 384 + lane * 4 + col] == U[o, c, r, col]
