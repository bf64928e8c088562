"""`deform_conv_ext` over libedvr_amd.so: the five functions of the reference's pybind11 CUDA extension
(basicsr/models/ops/dcn/src/deform_conv_ext.cpp:149-164) with the SAME argument lists, so that the reference's own Python
(basicsr/models/ops/dcn/deform_conv.py, unchanged) runs on the HIP kernels.  Drop this file in as
`basicsr/models/ops/dcn/deform_conv_ext.py` (or put `edvr_amd.compat` on the import path under that name); nothing is compiled
on the BasicSR side.

    modulated_deform_conv_forward / _backward     deform_conv_ext.cpp:106-147  ->  edvr_dcnv2_{fwd,bwd}_f32
    deform_conv_forward                           deform_conv_ext.cpp:51-66    ->  edvr_dcnv1_fwd_f32
    deform_conv_backward_input / _parameters      deform_conv_ext.cpp:68-104   ->  edvr_dcnv1_bwd_f32

Semantics kept from deform_conv_cuda.cpp (the callers rely on them):
  * every tensor is allocated by the caller (deform_conv.py:37-41,71-86,138-140,154-158); `output` is overwritten;
  * the gradients the reference ACCUMULATES into - grad_input (col2im's atomicAdd, .cu:688 / :322), grad_weight and grad_bias
    (addmm_ with beta = 1, deform_conv_cuda.cpp:460-468,659-672; `scale` for DCNv1) - are accumulated here too: the caller's
    pre-zeroed buffers give the plain gradient, non-zero buffers keep their contents; grad_offset / grad_mask are overwritten
    as the reference's coord kernels do;
  * `columns` / `ones` / `im2col_step` are scratch / blocking arguments of the reference's implementation: accepted, unused (the
    workspace is the explicit, grow-only one of edvr_amd.ops);
  * launches go to the current stream under the tensors' device; no host synchronisation (the DCNv2 forward leaves the mean
    |offset| of its call in pinned memory behind an event; the backward of the same offsets reads it if it has arrived and
    passes it on as the scatter hint of include/edvr_amd.h - the strategy of dX, never its value);
  * contiguity of input / weight is required like deform_conv_cuda.cpp:497-498, kernel sizes are checked against the weight like
    :507-512; CPU tensors are refused ("not implemented on CPU", deform_conv_ext.cpp:66,85,103,123,145).
"""
import torch

from .. import ops


_OFFSET_STATS = {}  # _stat_key(offset) -> ops.note_abs_mean record: forward -> backward of the same call


def _note_offsets(offset):
    if offset.dtype != torch.float32 or offset.dim() != 4:
        return
    if len(_OFFSET_STATS) > 256:  # forwards whose backward never came (inference through the training entry point)
        _OFFSET_STATS.clear()
    _OFFSET_STATS[_stat_key(offset)] = ops.note_abs_mean(offset)


def _stat_key(offset):
    # (address, version, shape): a different tensor that happens to reuse the address does not pick up this one's statistic
    return (offset.data_ptr(), offset._version, tuple(offset.shape))


def _scatter_hint(offset):
    st = ops.offset_stats_if_ready(_OFFSET_STATS.pop(_stat_key(offset), None))
    if st is None:  # unknown, or still on its way: the default strategy
        return ops.DCN_SCATTER_AUTO
    from ..functional import scatter_hint_from_stats
    return scatter_hint_from_stats(*st)


_DW_MEMO = []  # [(key, d(weight))] of the latest deform_conv_backward_input call


def _memo_key(input, offset, grad_output, geometry):
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (input, offset, grad_output)) + (geometry,)


def _out_view(output, input, weight, offset):
    """The caller's `output` buffer as the (B, Co, Ho, Wo) tensor the kernels write in place (the reference resizes / views it,
    deform_conv_cuda.cpp:530-536); a buffer of another size or layout is replaced like `output.resize_` would."""
    shape = (input.shape[0], weight.shape[0], offset.shape[2], offset.shape[3])
    if output.numel() != shape[0] * shape[1] * shape[2] * shape[3]:
        output.resize_(shape)
    if not output.is_contiguous():  # a viewable but strided buffer of the right size: compute into a temporary, copy back (_finish_out)
        return torch.empty(shape, dtype=output.dtype, device=output.device)
    return output.view(shape)


def _finish_out(tmp, output):
    if tmp.data_ptr() != output.data_ptr():
        output.view_as(tmp).copy_(tmp)


def _check(input, weight, kh, kw, group):
    for t in (input, weight):
        if not t.is_cuda:
            raise RuntimeError('deformable conv is not implemented on CPU')  # deform_conv_ext.cpp:66
        if not t.is_contiguous():
            raise RuntimeError('input tensor has to be contiguous')
    if weight.shape[2] != kh or weight.shape[3] != kw:
        raise RuntimeError(f'Input shape and kernel shape wont match: ({kh} x {kw} vs {weight.shape[2]} x {weight.shape[3]}).')
    if input.shape[1] != weight.shape[1] * group:
        raise RuntimeError(f'Input shape and kernel channels wont match: ({input.shape[1]} vs {weight.shape[1] * group}).')


# ------------------------------------------------------------------------------------------------ DCNv2
def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w, stride_h, stride_w,
                                  pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, with_bias):
    _check(input, weight, kernel_h, kernel_w, group)
    out = _out_view(output, input, weight, offset)
    ops.dcnv2_forward(input, offset, mask, weight, bias if with_bias else None, (stride_h, stride_w), (pad_h, pad_w),
                      (dilation_h, dilation_w), group, deformable_group, out=out)
    _finish_out(out, output)
    _note_offsets(offset)


def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight, grad_bias, grad_offset,
                                   grad_mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                   group, deformable_group, with_bias):
    _check(input, weight, kernel_h, kernel_w, group)
    # d(offset) / d(mask) are overwritten by the reference's col2im_coord (.cu:696-767): written in place here; d(input) /
    # d(weight) / d(bias) are ACCUMULATED by the reference (atomicAdd / addmm_ on the caller's buffers): computed, then added
    direct = all(t.is_contiguous() and t.dtype == offset.dtype for t in (grad_offset, grad_mask)) and \
        tuple(grad_offset.shape) == tuple(offset.shape) and tuple(grad_mask.shape) == tuple(mask.shape)
    dx, doff, dmsk, dw, db = ops.dcnv2_backward(input, offset, mask, weight, grad_output, bool(with_bias), (stride_h, stride_w),
                                                (pad_h, pad_w), (dilation_h, dilation_w), group, deformable_group,
                                                doffset=grad_offset if direct else None, dmask=grad_mask if direct else None,
                                                scatter_hint=_scatter_hint(offset))
    grad_input.view_as(dx).add_(dx)
    grad_weight.add_(dw)
    if with_bias:
        grad_bias.add_(db)
    if not direct:
        grad_offset.copy_(doff)
        grad_mask.copy_(dmsk)


# ------------------------------------------------------------------------------------------------ DCNv1
def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW, dilationH, group,
                        deformable_group, im2col_step):
    _check(input, weight, kH, kW, group)
    _DW_MEMO.clear()  # (a d(weight) kept for a backward_parameters call that never came - weights without gradient - is dropped here)
    out = _out_view(output, input, weight, offset)
    ops.dcnv1_forward(input, offset, weight, (dH, dW), (padH, padW), (dilationH, dilationW), group, deformable_group, out=out)
    _finish_out(out, output)
    return 1  # the reference returns 1 on success (deform_conv_cuda.cpp:242)


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW, padH, dilationW,
                               dilationH, group, deformable_group, im2col_step):
    _check(input, weight, kH, kW, group)
    dx, doff, dw = ops.dcnv1_backward(input, offset, weight, gradOutput, (dH, dW), (padH, padW), (dilationH, dilationW), group,
                                      deformable_group)
    # the reference's Function calls backward_input and backward_parameters back to back on the same tensors (deform_conv.py:77-97);
    # the one launch sequence above produced d(weight) as well: kept for that second call instead of running everything again
    _DW_MEMO[:] = [(_memo_key(input, offset, gradOutput, (kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group)), dw)]
    gradInput.view_as(dx).add_(dx)
    gradOffset.copy_(doff)
    return 1


def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                                    dilationH, group, deformable_group, scale, im2col_step):
    _check(input, gradWeight, kH, kW, group)
    key = _memo_key(input, offset, gradOutput, (kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group))
    memo = _DW_MEMO.pop() if _DW_MEMO else None
    if memo is not None and memo[0] == key and tuple(memo[1].shape) == tuple(gradWeight.shape):
        dw = memo[1]
    else:
        weight = torch.zeros_like(gradWeight)  # d(weight) does not depend on the weight values; the one-call ABI wants a pointer
        _, _, dw = ops.dcnv1_backward(input, offset, weight, gradOutput, (dH, dW), (padH, padW), (dilationH, dilationW), group,
                                      deformable_group)
    gradWeight.add_(dw, alpha=float(scale))
    return 1
