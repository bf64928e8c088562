"""TEST INFRASTRUCTURE ONLY - CPU oracle for the DCNv2 op of the EDVR hot path.

Three witnesses of the same algorithm, all CPU:

* ``c_forward`` / ``c_backward``       our plain-C restatement (oracle/dcnv2_oracle.c)
* ``ref_forward`` / ``ref_backward``   the reference's OWN kernels
  (basicsr/models/ops/dcn/src/deform_conv_cuda_kernel.cu:467-767) compiled as serial
  C++ into oracle/_ref/libdcn_ref.so (oracle/Makefile, oracle/ref_driver.cpp)
* ``dcnv2_torch``                      pure-torch floor/gather restatement whose autograd
  IS the reference backward (SURVEY.md finding 6) - used inside the whole-network oracle.

Pinning: the reference has no golden vectors for this op; the C restatement and
the torch restatement are checked against ``ref_*`` (the reference's own code,
executed) in tests/test_oracle_vs_ref.py, and all three against tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  edvr_amd/ never does.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(ref=True, quiet=True):
    """Compile the C restatement and, when /root/reference is present, oracle/_ref."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(['make', '-C', _HERE, 'all'], stdout=out)
    if ref and os.path.exists('/root/reference/basicsr/models/ops/dcn/src/deform_conv_cuda_kernel.cu'):
        subprocess.check_call(['make', '-C', _HERE, 'ref'], stdout=out)


def _load(name):
    if name in _LIBS:
        return _LIBS[name]
    path = os.path.join(_HERE, 'liboracle_dcn.so' if name == 'oracle' else os.path.join('_ref', 'libdcn_ref.so'))
    if not os.path.exists(path) and name == 'oracle':
        build(ref=False)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    lib = ctypes.CDLL(path)
    _LIBS[name] = lib
    return lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, '_ref', 'libdcn_ref.so'))


def num_threads():
    return int(_load('oracle').oracle_num_threads())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _hw(v):
    """int or (h, w) pair -> (h, w)  (deform_conv.py:33-36 `_pair`)"""
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _out_hw(H, W, kh, kw, stride, pad, dil):
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(pad), _hw(dil)
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def _prep(*ts):
    dt = ts[0].dtype
    assert dt in (torch.float32, torch.float64)
    return [None if t is None else t.detach().to('cpu', dt).contiguous() for t in ts], ('f32' if dt == torch.float32 else 'f64')


def _forward(libname, x, offset, mask, weight, bias, stride, pad, dil, groups, dg):
    (x, offset, mask, weight, bias), sfx = _prep(x, offset, mask, weight, bias)
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = _out_hw(H, W, kh, kw, stride, pad, dil)
    y = torch.empty(B, Co, Ho, Wo, dtype=x.dtype)
    lib = _load(libname)
    ints = [ctypes.c_int(v) for v in (B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg)]
    if libname == 'oracle':
        rc = getattr(lib, 'oracle_dcnv2_forward_' + sfx)(_p(x), _p(offset), _p(mask), _p(weight), _p(bias), _p(y), *ints)
    else:
        rc = getattr(lib, 'ref_mdcn_forward_' + sfx)(_p(x), _p(weight), _p(bias), _p(offset), _p(mask), _p(y), *ints)
    if rc != 0:
        raise RuntimeError(f'{libname} dcnv2 forward failed rc={rc}')
    return y


def _backward(libname, x, offset, mask, weight, dy, with_bias, stride, pad, dil, groups, dg):
    (x, offset, mask, weight, dy), sfx = _prep(x, offset, mask, weight, dy)
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    dx, doff, dmsk, dw = (torch.zeros_like(t) for t in (x, offset, mask, weight))
    db = torch.zeros(Co, dtype=x.dtype) if with_bias else None
    lib = _load(libname)
    ints = [ctypes.c_int(v) for v in (B, C, H, W, Co, kh, kw, stride, pad, dil, groups, dg)]
    if libname == 'oracle':
        rc = getattr(lib, 'oracle_dcnv2_backward_' + sfx)(_p(x), _p(offset), _p(mask), _p(weight), _p(dy), _p(dx),
                                                           _p(doff), _p(dmsk), _p(dw), _p(db), *ints)
    else:
        rc = getattr(lib, 'ref_mdcn_backward_' + sfx)(_p(x), _p(weight), _p(offset), _p(mask), _p(dy), _p(dx), _p(dw),
                                                       _p(db), _p(doff), _p(dmsk), *ints)
    if rc != 0:
        raise RuntimeError(f'{libname} dcnv2 backward failed rc={rc}')
    return dx, doff, dmsk, dw, db


def c_forward(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    return _forward('oracle', x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups)


def c_backward(x, offset, mask, weight, dy, with_bias=True, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    return _backward('oracle', x, offset, mask, weight, dy, with_bias, stride, padding, dilation, groups, deformable_groups)


def ref_forward(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    return _forward('ref', x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups)


def ref_backward(x, offset, mask, weight, dy, with_bias=True, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    return _backward('ref', x, offset, mask, weight, dy, with_bias, stride, padding, dilation, groups, deformable_groups)


class _COracleFn(torch.autograd.Function):
    """autograd wrapper over the C restatement (multi-threaded; used for whole-network CPU runs)."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation, groups, dg):
        ctx.cfg = (stride, padding, dilation, groups, dg)
        ctx.with_bias = bias is not None
        ctx.save_for_backward(x, offset, mask, weight)
        return c_forward(x, offset, mask, weight, bias, stride, padding, dilation, groups, dg)

    @staticmethod
    def backward(ctx, dy):
        x, offset, mask, weight = ctx.saved_tensors
        dx, doff, dmsk, dw, db = c_backward(x, offset, mask, weight, dy, ctx.with_bias, *ctx.cfg)
        return dx, doff, dmsk, dw, db, None, None, None, None, None


def dcnv2_c(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    return _COracleFn.apply(x, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups)


def dcnv2_torch(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """Floor/gather restatement (kernel.cu:570-633,467-497); differentiable in every input.

    The cell is fixed by floor() (detached), so d/d(offset) is the one-sided derivative inside
    [floor p, floor p + 1) exactly as kernel.cu:526-568 computes it - also at integer positions.
    """
    dg = deformable_groups
    B, C, H, W = x.shape
    Co, cig, kh, kw = weight.shape
    K, cpg, dt = kh * kw, C // dg, x.dtype
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(padding), _hw(dilation)  # pairs: DCNv1 (deform_conv.py:33-36), .cu:211-216
    dev = x.device
    ys = (torch.arange(Ho, dtype=dt, device=dev) * sh - ph).view(1, 1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=dt, device=dev) * sw - pw).view(1, 1, 1, 1, Wo)
    ki = (torch.arange(kh, dtype=dt, device=dev) * dh).repeat_interleave(kw).view(1, 1, K, 1, 1)
    kj = (torch.arange(kw, dtype=dt, device=dev) * dw).repeat(kh).view(1, 1, K, 1, 1)
    off = offset.reshape(B, dg, K, 2, Ho, Wo)
    py = ys + ki + off[:, :, :, 0]
    px = xs + kj + off[:, :, :, 1]
    valid = ((py > -1) & (px > -1) & (py < H) & (px < W)).to(dt)
    y0 = torch.floor(py).detach()
    x0 = torch.floor(px).detach()
    ly, lx = py - y0, px - x0
    xg = x.reshape(B, dg, cpg, H * W)

    def corner(yy, xx):
        ok = ((yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)).to(dt)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().view(B, dg, 1, -1).expand(B, dg, cpg, K * Ho * Wo)
        return torch.gather(xg, 3, idx).view(B, dg, cpg, K, Ho, Wo) * ok.unsqueeze(2)

    v = (corner(y0, x0) * ((1 - ly) * (1 - lx)).unsqueeze(2) + corner(y0, x0 + 1) * ((1 - ly) * lx).unsqueeze(2)
         + corner(y0 + 1, x0) * (ly * (1 - lx)).unsqueeze(2) + corner(y0 + 1, x0 + 1) * (ly * lx).unsqueeze(2))
    col = (v * (valid * mask.reshape(B, dg, K, Ho, Wo)).unsqueeze(2)).reshape(B, groups, cig * K, Ho * Wo)
    out = torch.einsum('gok,bgkp->bgop', weight.reshape(groups, Co // groups, cig * K), col).reshape(B, Co, Ho, Wo)
    return out if bias is None else out + bias.view(1, -1, 1, 1)


# ------------------------------------------------------------------------------------------------ DCNv1
def ref_dcn1_forward(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """The reference's OWN deformable_im2col kernel (.cu:190-250) + the restated driver (deform_conv_cuda.cpp:152-243), fp64."""
    lib = _load('ref')
    (x, offset, weight), _ = _prep(x.double(), offset.double(), weight.double())
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    y = torch.empty(B, Co, Ho, Wo, dtype=torch.float64)
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(padding), _hw(dilation)
    rc = lib.ref_dcn1_forward_rect_f64(_p(x), _p(weight), _p(offset), _p(y), B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                       deformable_groups)
    assert rc == 0
    return y


def ref_dcn1_backward(x, offset, weight, dy, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """(dx, doffset, dweight) from the reference's own col2im / col2im_coord / im2col kernels (.cu:280-465), fp64."""
    lib = _load('ref')
    (x, offset, weight, dy), _ = _prep(x.double(), offset.double(), weight.double(), dy.double())
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    dx, dw, doff = torch.zeros_like(x), torch.zeros_like(weight), torch.zeros_like(offset)
    (sh, sw), (ph, pw), (dh, dw_) = _hw(stride), _hw(padding), _hw(dilation)
    rc = lib.ref_dcn1_backward_rect_f64(_p(x), _p(weight), _p(offset), _p(dy), _p(dx), _p(dw), _p(doff), B, C, H, W, Co, kh, kw, sh, sw,
                                        ph, pw, dh, dw_, groups, deformable_groups)
    assert rc == 0
    return dx, doff, dw


def c_dcn1_forward(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """DCNv1 on the plain-C oracle: DCNv2 with an all-ones mask and no bias (pinned against ref_dcn1_* in tests/test_oracle_vs_ref.py)."""
    B, _, Ho, Wo = offset.shape
    ones = torch.ones(B, offset.shape[1] // 2, Ho, Wo, dtype=offset.dtype)
    return c_forward(x, offset, ones, weight, None, stride, padding, dilation, groups, deformable_groups)


def c_dcn1_backward(x, offset, weight, dy, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    B, _, Ho, Wo = offset.shape
    ones = torch.ones(B, offset.shape[1] // 2, Ho, Wo, dtype=offset.dtype)
    dx, doff, _, dw, _ = c_backward(x, offset, ones, weight, dy, False, stride, padding, dilation, groups, deformable_groups)
    return dx, doff, dw


def torch_dcn1_forward(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """DCNv1 on the floor/gather restatement (dcnv2_torch with an all-ones mask): the one oracle that takes rectangular
    stride / padding / dilation pairs (pinned against the reference's own kernels on tests/golden/dcn1_rect.pt)."""
    B, _, Ho, Wo = offset.shape
    ones = torch.ones(B, offset.shape[1] // 2, Ho, Wo, dtype=offset.dtype, device=offset.device)
    return dcnv2_torch(x, offset, ones, weight, None, stride, padding, dilation, groups, deformable_groups)


def torch_dcn1_backward(x, offset, weight, dy, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    t = [v.detach().clone().requires_grad_() for v in (x, offset, weight)]
    torch_dcn1_forward(*t, stride, padding, dilation, groups, deformable_groups).backward(dy)
    return t[0].grad, t[1].grad, t[2].grad
