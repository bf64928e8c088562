"""Tensor-level wrappers over the C ABI (no autograd here; see edvr_amd/functional.py).

PyTorch is plumbing only: it owns device memory and the current HIP stream.  Every
function launches hand-written HIP kernels from libedvr_amd.so on
``torch.cuda.current_stream()`` and never synchronises.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONV_AUTO, CONV_DIRECT, CONV_WINOGRAD, CONV_WINOGRAD_F4, CONV_WINOGRAD_F4S, OUT_NCHW,  # noqa: F401
                   OUT_PIXEL_SHUFFLE2)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_gpu(*tensors, dtypes=(torch.float32,)):
    """Same refusal as the reference (deform_conv.py:133-134): no CPU path exists.  Everything is fp32 except the DCN operators,
    which also take float64 / float16 like the reference's dispatch (`dtypes`); the tensors of one call share one type."""
    seen = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError('edvr_amd ops run on the GPU only (HIP/gfx950); got a CPU tensor')
        if t.dtype not in dtypes:
            raise NotImplementedError(f'edvr_amd: {t.dtype} is not supported here (supported: {", ".join(map(str, dtypes))})')
        if seen is not None and t.dtype != seen:
            raise RuntimeError(f'expected every tensor of the call to be {seen}, got {t.dtype}')  # (the reference: "expected scalar type ...")
        seen = t.dtype
    return seen


DCN_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.float64: _lib.DTYPE_F64, torch.float16: _lib.DTYPE_F16}


def _plane_contig(t):
    """(n, c, h, w) whose (c, h, w) block is dense; the image stride may be larger (channel-sliced view)."""
    n, c, h, w = t.shape
    return t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == h * w and (n == 1 or t.stride(0) >= c * h * w)


def _as_planes(t):
    return t if _plane_contig(t) else t.contiguous()


def _img_stride(t):
    return t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2] * t.shape[3]


def _run(name, launch, flops=0.0, nbytes=0.0, executed=None):
    """Every launch of this module goes through here: LAUNCH_HOOK (bench.py's instrumented pass) brackets it with events on
    the launch stream and books `flops` / `nbytes` (ALGORITHMIC work of the call: each input / output element once) and
    `executed` (flops the matrix cores issue for it, padding included; None = not known here) under `name`."""
    if LAUNCH_HOOK is not None:
        LAUNCH_HOOK(name, float(flops), launch, float(nbytes), executed)
    else:
        launch()


def _nb(*ts):
    return 4.0 * sum(t.numel() for t in ts if t is not None)


# ------------------------------------------------------------------------------------------------ workspace
_WS = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream).  288 GB of HBM: keep it resident, never free per call."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _WS.pop(key, None)
        buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------ weights
_PACKED = {}  # id(parameter) -> (weakref, {transpose_flip: (version, packed, data_ptr)})


def pack_conv_weight(weight, transpose_flip=False, f4=False, f4s=False):
    """(co, ci, k, k) parameter -> MFMA-friendly [ci_pad][k*k][co_pad] array, cached per parameter version.
    f4: the F(4x4,3x3) Winograd weights of a 3x3 kernel instead (conv2d's `wpk_f4`), a separate buffer with its own cache slot.
    f4s: the same weights for the split-operand kernel (conv2d's `wpk_f4s`: scaled, split into f16 (hi, lo) pairs; int32 buffer); of a
    1x1 kernel: the split-operand packing of the streaming 1x1 kernel (csrc/conv1x1_s.hip), same field."""
    require_gpu(weight)
    key, wid, ver = (bool(transpose_flip), 2 if f4s else bool(f4)), id(weight), weight._version
    ent = _PACKED.get(wid)
    if ent is not None and ent[0]() is weight:
        hit = ent[1].get(key)
        if hit is not None and hit[0] == ver and hit[2] == weight.data_ptr():
            return hit[1]
    else:
        ent = (weakref.ref(weight, lambda _r, wid=wid: _PACKED.pop(wid, None)), {})
        _PACKED[wid] = ent
    L = _lib.lib()
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    o, i, k, k2 = w.shape
    assert k == k2, 'square kernels only'
    co, ci = (i, o) if transpose_flip else (o, i)
    if f4s and k == 1:  # the split-operand packing of a 1x1 conv (csrc/conv1x1_s.hip)
        assert not transpose_flip, 'the split 1x1 kernel serves forward convs'
        n_el = L.edvr_conv2d_packed_weight_1x1s_elems(co, ci)
        stale = ent[1].get(key)  # this layout is not part of prepack_conv_weights' table: rewritten in place here under the same rule
        if stale is not None and stale[1].numel() == n_el and stale[1].device == w.device and _sole_owner(stale):
            out = stale[1]
        else:
            out = torch.empty(n_el, dtype=torch.int32, device=w.device)
        _lib.check(L.edvr_conv2d_pack_weight_1x1s_f32(_ptr(w), _ptr(out), co, ci, _stream()), 'edvr_conv2d_pack_weight_1x1s_f32')
    elif f4s:
        assert k == 3, 'F(4x4,3x3) weights are for 3x3 kernels'
        out = torch.empty(L.edvr_conv2d_packed_weight_f4s_elems(co, ci), dtype=torch.int32, device=w.device)
        _lib.check(L.edvr_conv2d_pack_weight_f4s_f32(_ptr(w), _ptr(out), co, ci, 1 if transpose_flip else 0, _stream()),
                   'edvr_conv2d_pack_weight_f4s_f32')
    elif f4:
        assert k == 3, 'F(4x4,3x3) weights are for 3x3 kernels'
        out = torch.empty(L.edvr_conv2d_packed_weight_f4_elems(co, ci), dtype=torch.float32, device=w.device)
        _lib.check(L.edvr_conv2d_pack_weight_f4_f32(_ptr(w), _ptr(out), co, ci, 1 if transpose_flip else 0, _stream()),
                   'edvr_conv2d_pack_weight_f4_f32')
    else:
        out = torch.empty(L.edvr_conv2d_packed_weight_elems(co, ci, k), dtype=torch.float32, device=w.device)
        _lib.check(L.edvr_conv2d_pack_weight_f32(_ptr(w), _ptr(out), co, ci, k, 1 if transpose_flip else 0, _stream()),
                   'edvr_conv2d_pack_weight_f32')
    ent[1][key] = (ver, out, weight.data_ptr())
    return out


_PACK_CALLS = 0   # edvr_conv2d_pack_weights_multi's split_parity counter
_PACK_TABLES = {}  # job signature -> (device table, total blocks): the table of a training iteration never changes


_PINNED_PACKED = set()  # data_ptr()s of packed buffers somebody outside the cache relies on (a captured hipGraph): never rewritten


def pin_packed_weights(weights):
    """Keep the CURRENT packed layouts of `weights` alive and unmodified: returns the list of buffers (hold it as long as they are
    needed - e.g. a captured hipGraph whose launches read them, edvr_amd/graphs.py) after registering them so that
    prepack_conv_weights allocates fresh buffers instead of rewriting these.  Release with unpin_packed_weights(list)."""
    bufs = []
    for w in weights:
        ent = _PACKED.get(id(w))
        if ent is not None and ent[0]() is w:
            bufs.extend(hit[1] for hit in ent[1].values())
    _PINNED_PACKED.update(b.data_ptr() for b in bufs)
    return bufs


def unpin_packed_weights(bufs):
    for b in bufs:
        _PINNED_PACKED.discard(b.data_ptr())


def _sole_owner(hit):
    """The cache entry `hit` = (version, buffer, pointer) is the only holder of its buffer: no other Python reference, no view, no
    C++ holder (autograd node, graph), not pinned.  Only then may the buffer be rewritten in place."""
    import sys
    return hit[1]._use_count() == 1 and sys.getrefcount(hit[1]) == 2 and hit[1].data_ptr() not in _PINNED_PACKED


def prepack_conv_weights(weights, meta=None):
    """Fill the packed-weight cache for a whole network in ONE launch (edvr_conv2d_pack_weights_multi): every (weight, orientation)
    the training step will ask pack_conv_weight / f4_weight for whose cached copy is stale - after an optimizer step: all of them,
    ~480 separate ~5 us launches otherwise.  `weights`: iterable of conv weight Parameters (3x3 or 1x1); `meta` (optional, same
    length): (forward_f4, data_gradient) per weight - False skips layouts the autograd path never requests (the F(4x4) forward
    layout of a stride-2 conv, the flipped layouts of a conv whose input needs no gradient).  Purely an optimisation: whatever is
    not packed here is packed lazily, one launch per request, as before.
    A stale buffer is REWRITTEN IN PLACE when the cache is its only owner (`_sole_owner`: nobody else holds the tensor, it is not
    pinned by a captured graph), so that the job table is built and uploaded once; otherwise a fresh buffer is allocated and the
    old one stays as it was for whoever holds it.  The rewrite is ordered behind the buffer's readers on the CURRENT stream only:
    work queued on another stream that still reads a packed layout must be synchronised with by the caller (single-stream
    assumption of the training loop; GraphedEDVR pins its layouts instead)."""
    import numpy as np
    L = _lib.lib()
    jobs, sig, filled = [], [], []
    meta = list(meta) if meta is not None else None
    for wi, w in enumerate(weights):
        fwd_f4, dgrad = meta[wi] if meta is not None else (True, True)
        if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 4 or w.shape[2] != w.shape[3] or w.shape[2] not in (1, 3) or not w.is_contiguous():
            continue
        wid, ver, k = id(w), w._version, w.shape[2]
        ent = _PACKED.get(wid)
        if ent is None or ent[0]() is not w:
            ent = (weakref.ref(w, lambda _r, wid=wid: _PACKED.pop(wid, None)), {})
            _PACKED[wid] = ent
        for flip in ((False, True) if dgrad else (False,)):
            co, ci = (w.shape[1], w.shape[0]) if flip else (w.shape[0], w.shape[1])
            want_f4 = F4_TRAINING and k == 3 and ci >= 32 and co >= 48 and (flip or fwd_f4)  # = f4_weight()
            # which layouts: the direct one always; of the two F(4x4) layouts the one the training convs will ask for (f4_kwargs)
            kinds = [False] + ([2 if F4S_TRAINING else True] if want_f4 else [])
            outs = []
            for f4 in kinds:
                hit = ent[1].get((flip, f4))
                if hit is not None and hit[0] == ver and hit[2] == w.data_ptr():
                    outs.append(None)  # current
                    continue
                if f4 == 2:
                    n, dt = L.edvr_conv2d_packed_weight_f4s_elems(co, ci), torch.int32
                else:
                    n, dt = (L.edvr_conv2d_packed_weight_f4_elems(co, ci) if f4 else L.edvr_conv2d_packed_weight_elems(co, ci, k)), torch.float32
                if hit is not None and hit[1].numel() == n and hit[1].dtype == dt and hit[1].device == w.device and _sole_owner(hit):
                    buf = hit[1]
                else:
                    buf = torch.empty(n, dtype=dt, device=w.device)
                    if f4 == 2:
                        buf[:16].zero_()  # the header's max |w| slots start at zero (edvr_conv2d_pack_weights_multi)
                outs.append(buf)
                filled.append((ent, (flip, f4), ver, buf, w.data_ptr()))
            wpk = outs[0]
            wf4 = outs[1] if (want_f4 and not F4S_TRAINING) else None
            wf4s = outs[1] if (want_f4 and F4S_TRAINING) else None
            if wpk is None and wf4 is None and wf4s is None:
                continue
            wq = wf4 if wf4 is not None else wf4s
            work = max((wpk.numel() if wpk is not None else 0), (wq.numel() // 36 if wq is not None else 0))
            jobs.append((w.data_ptr(), wpk.data_ptr() if wpk is not None else 0, wf4.data_ptr() if wf4 is not None else 0, co, ci, k,
                         int(flip), max(1, min(64, (work + 255) // 256)), wf4s.data_ptr() if wf4s is not None else 0))
    if not jobs:
        return 0
    key = tuple(jobs)  # (device pointers inside: unique per device)
    tab = _PACK_TABLES.get(key)
    if tab is None:
        rec = np.zeros(len(jobs), dtype=np.dtype([('w', '<u8'), ('wpk', '<u8'), ('f4', '<u8'), ('co', '<i4'), ('ci', '<i4'), ('ks', '<i4'),
                                                  ('flip', '<i4'), ('first', '<i4'), ('nb', '<i4'), ('f4s', '<u8'), ('pad', '<i4', (2,))]))
        assert rec.dtype.itemsize == L.edvr_pack_job_bytes()
        first = 0
        for i, (pw, pk, pf, co, ci, k, flip, nb, ps) in enumerate(jobs):
            rec[i] = (pw, pk, pf, co, ci, k, flip, first, nb, ps, (0, 0))
            first += nb
        dev = filled[0][3].device
        tab = (torch.from_numpy(rec.view(np.uint8).copy()).to(dev), first)
        if len(_PACK_TABLES) > 8:
            _PACK_TABLES.clear()
        _PACK_TABLES[key] = tab
    global _PACK_CALLS
    _PACK_CALLS += 1
    parity = _PACK_CALLS if any(j[8] for j in jobs) else -1
    _lib.check(L.edvr_conv2d_pack_weights_multi(_ptr(tab[0]), len(jobs), tab[1], parity, _stream()), 'edvr_conv2d_pack_weights_multi')
    for ent, k2, ver, buf, ptr in filled:
        ent[1][k2] = (ver, buf, ptr)
    return len(jobs)


def f4_weight(weight, ks, transpose_flip=False):
    """The F(4x4,3x3) packing of a 3x3 weight for the training path, or None when that path is off / the layer too small."""
    if not F4_TRAINING or ks != 3:
        return None
    co, ci = (weight.shape[1], weight.shape[0]) if transpose_flip else (weight.shape[0], weight.shape[1])
    return pack_conv_weight(weight, transpose_flip=transpose_flip, f4=True) if (ci >= 32 and co >= 48) else None


def f4_kwargs(weight, ks, transpose_flip=False):
    """The F(4x4) weights a training-path conv2d() call should get: {'wpk_f4s': ...} (split-operand kernel), {'wpk_f4': ...} (fp32
    kernel) or {} when the F(4x4) path is off / the layer too small."""
    if ks == 1 and F4S_TRAINING and not transpose_flip and weight.shape[1] >= 320 and weight.shape[1] % 8 == 0:
        return {'wpk_f4s': pack_conv_weight(weight, f4s=True)}  # the streaming 1x1 kernel's split form (csrc/conv1x1_s.hip; forward only)
    if not F4_TRAINING or ks != 3:
        return {}
    co, ci = (weight.shape[1], weight.shape[0]) if transpose_flip else (weight.shape[0], weight.shape[1])
    if not (ci >= 32 and co >= 48):
        return {}
    if F4S_TRAINING:
        return {'wpk_f4s': pack_conv_weight(weight, transpose_flip=transpose_flip, f4s=True)}
    return {'wpk_f4': pack_conv_weight(weight, transpose_flip=transpose_flip, f4=True)}


def invalidate_packed_weights():
    """Drop every cached packed weight.  The cache follows a parameter's autograd version and storage pointer, which in-place
    torch ops, optimizers and load_state_dict maintain; a write through `.data` (or through a raw pointer that does not call
    torch.autograd.graph.increment_version afterwards) does not - call this after such a write.  The cached weight norms behind
    linear_bound() (the magnitude bounds of the split-operand kernels) follow the same key and are dropped with them."""
    _PACKED.clear()
    _W_L1.clear()


# ------------------------------------------------------------------------------------------------ conv
DCN_SCATTER_AUTO, DCN_SCATTER_DEVICE, DCN_SCATTER_LDS, DCN_SCATTER_STRIP, DCN_SCATTER_LDS_WIDE = 0, 1, 2, 3, 4  # include/edvr_amd.h EDVR_DCN_SCATTER_*
_SCATTER_NAMES = {0: '[auto]', 1: '[device atomics]', 2: '[lds window]', 3: '[strip / fused]', 4: '[lds window]'}  # measurement label of dcnv2_backward's dX strategy
DCN_HALO_TAPWIN = _lib.DCN_HALO_TAPWIN  # halo_hint of dcnv2_forward: per-tap shifted windows (csrc/dcn_tapwin.hip)
LAUNCH_HOOK = None  # callable(kernel_name, algorithmic_flops, launch_fn, algorithmic_bytes) or None
CONV_ALGO = CONV_AUTO  # default algorithm request of conv2d(); tests flip it to cover both kernels on every shape
F4_INFERENCE = os.environ.get('EDVR_WINOGRAD_F4', '1') != '0'  # functional.conv hands the F(4x4,3x3) weights to no-grad convs
F4S_INFERENCE = os.environ.get('EDVR_WINOGRAD_F4S', '1') != '0'  # functional.conv hands the split-operand F(4x4) weights (csrc/winograd_f4s.hip) to no-grad convs
F4S_TRAINING = os.environ.get('EDVR_WINOGRAD_F4S_TRAIN', os.environ.get('EDVR_WINOGRAD_F4S', '1')) != '0'  # ... and to the forward / data-gradient convs of autograd.py
F4_TRAINING = os.environ.get('EDVR_WINOGRAD_F4_TRAIN', '1') != '0'  # autograd.py: forward and data-gradient convs too (-11 % per iteration;
#                                                                   the gradient parity tests hold at their 1e-5 bounds)


def set_f4(inference=None, training=None):
    """Switch the F(4x4,3x3) Winograd kernel on / off at run time for the no-grad path (`inference`) and for the forward / data-gradient
    convs of the training path (`training`); None leaves a setting alone.  Returns the previous (inference, training) pair.  The
    environment variables EDVR_WINOGRAD_F4 / EDVR_WINOGRAD_F4_TRAIN only give the initial values.  (F(4x4) rounds at ~1e-6 of the
    output scale, F(2x2) at ~2e-7.)"""
    global F4_INFERENCE, F4_TRAINING
    prev = (F4_INFERENCE, F4_TRAINING)
    if inference is not None:
        F4_INFERENCE = bool(inference)
    if training is not None:
        F4_TRAINING = bool(training)
    return prev


def set_f4s(inference=None, training=None):
    """Switch the SPLIT-OPERAND form of the F(4x4,3x3) kernel (csrc/winograd_f4s.hip) on / off at run time for the no-grad path and
    for the training path; with it off the fp32 F(4x4) kernel runs (set_f4).  Returns the previous pair.  EDVR_WINOGRAD_F4S /
    EDVR_WINOGRAD_F4S_TRAIN give the initial values."""
    global F4S_INFERENCE, F4S_TRAINING
    prev = (F4S_INFERENCE, F4S_TRAINING)
    if inference is not None:
        F4S_INFERENCE = bool(inference)
    if training is not None:
        F4S_TRAINING = bool(training)
    return prev


class _AmaxArena:
    """One-element device slots for the `y_amax` epilogue of the split-operand kernels and for the reduction passes: zeroed in blocks of
    2048, each slot handed out once (a slot captured by a hipGraph keeps accumulating maxima over replays: still a bound).  One
    arena per (device, stream): a block is zero-filled on the stream that uses it.  The slots handed out since the last
    guard_submit() are what the overflow guard examines (a non-finite maximum = non-finite values left a split-operand kernel)."""

    BLOCK = 2048

    def __init__(self):
        self.buf, self.i, self.mark, self.retired = None, 0, 0, []

    def fresh(self, device):
        """Start a new block now (GraphedEDVR: BEFORE a capture, so that no allocation / zero-fill is recorded into the graph)."""
        if self.buf is not None and self.i > self.mark:
            self.retired.append(self.buf[self.mark:self.i])
        self.buf, self.i, self.mark = torch.zeros(self.BLOCK, dtype=torch.float32, device=device), 0, 0

    def slot(self, device):
        if self.buf is None or self.i >= self.buf.numel():
            if self.buf is not None and device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('edvr_amd: the magnitude-bound arena ran out of slots inside a hipGraph capture (more than '
                                   f'{self.BLOCK} split-operand launches in one captured region); capture a smaller region')
            self.fresh(device)
        self.i += 1
        return self.buf[self.i - 1:self.i]

    def take_unexamined(self):
        out, self.retired = self.retired, []
        if self.buf is not None and self.i > self.mark:
            out.append(self.buf[self.mark:self.i])
            self.mark = self.i
        return out


_AMAX_ARENAS = {}  # (device index, stream handle) -> _AmaxArena


def _arena(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    a = _AMAX_ARENAS.get(key)
    if a is None:
        a = _AMAX_ARENAS[key] = _AmaxArena()
    return a


def reserve_amax_slots(device):
    """A fresh, zero-filled block of bound slots for the current stream of `device`: call before a hipGraph capture."""
    _arena(device).fresh(device)


BOUND_CHECK = os.environ.get('EDVR_BOUND_CHECK', '0') == '1'  # verify every x_amax against the data (tests / debugging)
BOUND_CHECK_LOG = []  # (shape, bound / true maximum) of every checked conv input
AMAX_LOG = None  # a list: every reduction pass of input_bound() is recorded there (shape, calling functions)
AMAX_PASSES = 0  # how often input_bound() had to run the reduction kernel (measurement: bench.py reports it per forward)


# ---- overflow guard of the split-operand path.  The split kernels place their operands in the f16 range from magnitude BOUNDS; a bound
# that is too small (a stale one: a raw-pointer write behind torch's back) overflows f16 to inf and the products to NaN - silently.
# Every split conv leaves max |y| in an arena slot with non-finite values STICKY (NaN bits order above every finite number), the
# reduction kernel does the same, so one 2048-element max per forward says whether anything non-finite passed through a split
# kernel.  It is read like the offset statistics: copy to pinned memory behind an event, examined when the next forward starts (or by
# split_guard_check(wait=True)); no forward waits for the GPU.
class SplitOperandOverflow(RuntimeError):
    pass


SPLIT_GUARD = os.environ.get('EDVR_SPLIT_GUARD', 'raise')  # 'raise' | 'fallback' (warn once, continue on the fp32 kernels) | 'off'
_GUARD_PENDING = []  # (pinned 1-element host tensor, copy-done event)
GUARD_TRIPS = 0


def split_guard_submit(device, parts=None):
    """Fold the bound slots handed out on the current stream since the last call (or `parts`: the slots of a replayed hipGraph) into
    one flag and send it to the host (no wait)."""
    if SPLIT_GUARD == 'off' or device.type != 'cuda' or torch.cuda.is_current_stream_capturing():
        return
    if parts is None:
        parts = _arena(device).take_unexamined()
    if not parts:
        return
    with torch.no_grad():
        worst = (parts[0] if len(parts) == 1 else torch.cat(parts)).max().reshape(1)  # torch.max propagates NaN
    host = torch.empty(1, dtype=torch.float32, pin_memory=True)
    host.copy_(worst, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    _GUARD_PENDING.append((host, done))
    if len(_GUARD_PENDING) > 64:
        split_guard_check(wait=True)


def split_guard_check(wait=False):
    """Examine the flags that have arrived (wait=True: all of them, one synchronisation).  Non-finite -> SplitOperandOverflow, or with
    EDVR_SPLIT_GUARD=fallback a warning and the fp32 kernels from here on."""
    global GUARD_TRIPS
    while _GUARD_PENDING:
        host, done = _GUARD_PENDING[0]
        if wait:
            done.synchronize()
        elif not done.query():
            return
        _GUARD_PENDING.pop(0)
        v = float(host[0])
        if v != v or v == float('inf'):
            GUARD_TRIPS += 1
            msg = ('edvr_amd: non-finite values left a split-operand kernel (max |y| = %r): either the input holds inf / NaN, or a '
                   'magnitude bound was stale (a tensor rewritten through a raw pointer after its bound was taken: see '
                   'ops.void_bound).  EDVR_WINOGRAD_F4S=0 runs the fp32 kernels.' % v)
            if SPLIT_GUARD == 'fallback':
                import warnings
                warnings.warn(msg + '  Continuing on the fp32 kernels.')
                set_f4s(False, False)
            else:
                raise SplitOperandOverflow(msg)


def _ver(t):
    """The autograd version counter a bound is tied to; inference tensors (torch.inference_mode) do not have one - and cannot be
    written in place outside inference mode: None."""
    return None if t.is_inference() else t._version


def set_bound(t, bound, depth=0):
    """Remember `bound` (1-element device tensor >= max |t|) on the tensor object; valid while t is not written again.
    depth: 0 = measured (the split-operand kernel's y_amax epilogue, the reduction kernel), 1 = derived from a measured one through the
    weights' norms (linear_bound) - typically 10-50x the true maximum, which the split costs nothing (it has 2^18 of slack); a SECOND
    derivation on top would multiply the looseness (1e10 after a few layers), so it is refused and the consumer measures instead.
    Every function of this module that writes into a caller's tensor through a raw pointer replaces or voids its bound (void_bound):
    the version counter does not see those writes."""
    t._edvr_amax = (bound, _ver(t), depth)


def get_bound(t):
    b = getattr(t, '_edvr_amax', None)
    return b[0] if (b is not None and b[1] == _ver(t) and b[0].device == t.device) else None


def void_bound(*ts):
    """Forget the bounds of tensors a kernel has just rewritten through raw pointers (`out=` buffers, in-place kernels)."""
    for t in ts:
        if t is not None and hasattr(t, '_edvr_amax'):
            del t._edvr_amax


def _bound_depth(t):
    return t._edvr_amax[2]


def carry_bound(dst, *srcs, scale=1.0):
    """max |dst| <= scale * (sum of the sources' bounds) by construction of the op that made it (a gate, a convex combination, a
    permutation, a sum): hand the bound on without looking at the data.  Nothing happens when a source has none."""
    bs = [get_bound(t) for t in srcs]
    if not bs or any(b is None for b in bs):
        return dst
    b = bs[0]
    for o in bs[1:]:
        b = b + o
    set_bound(dst, b if scale == 1.0 else b * float(scale), max(_bound_depth(t) for t in srcs))
    return dst


_W_L1 = {}  # id(weight) -> (weakref, {transpose: (version, pointer, l1max)})


def weight_l1max(weight, transpose=False):
    """max over output channels of sum |w| (1-element device tensor), cached per parameter version: |conv(x, w)| <= that * max|x|.
    transpose: of the data-gradient kernel (sums over the output-channel axis of w)."""
    wid, ver = id(weight), weight._version
    ent = _W_L1.get(wid)
    if ent is None or ent[0]() is not weight:
        ent = (weakref.ref(weight, lambda _r, wid=wid: _W_L1.pop(wid, None)), {})
        _W_L1[wid] = ent
    hit = ent[1].get(bool(transpose))
    if hit is not None and hit[0] == ver and hit[1] == weight.data_ptr():
        return hit[2]
    w = weight.detach().abs()
    l1 = w.sum(dim=(0, 2, 3) if transpose else (1, 2, 3)).max().reshape(1)
    ent[1][bool(transpose)] = (ver, weight.data_ptr(), l1)
    return l1


def linear_bound(y, weight, bias, xs, extra=(), transpose=False, scale=1.0, floor=0.0):
    """Bound of y = scale * act(conv / deformable conv of xs with `weight` + bias) + extra tensors, for kernels without the y_amax
    epilogue: |y| <= |scale| (max_co sum|w| * max|x| + max|b|) + sum of the extras' bounds (1-Lipschitz activations through 0;
    `floor` = 1 for a sigmoid epilogue; deformable sampling and masks in [0, 1] are convex combinations of the input).  A few
    1-element torch ops; nothing happens when an input has no bound."""
    srcs = [t for t in xs if t is not None] + [t for t in extra if t is not None]
    bs = [get_bound(t) for t in srcs]
    if any(b is None for b in bs) or any(_bound_depth(t) > 0 for t in xs if t is not None):
        return y
    nx = sum(1 for t in xs if t is not None)
    bx = bs[0]
    for o in bs[1:nx]:
        bx = torch.maximum(bx, o)
    b = weight_l1max(weight, transpose) * bx
    if bias is not None:
        b = b + bias.detach().abs().max()
    if scale != 1.0:
        b = b * abs(float(scale))
    if floor:
        b = b.clamp_min(float(floor))
    for o in bs[nx:]:
        b = b + o
    set_bound(y, b, 1)
    return y


def input_bound(t_in, t_planes=None):
    """max |t| as a 1-element device tensor: the bound its producer left on it (set_bound), else one pass of the reduction kernel."""
    global AMAX_PASSES
    b = get_bound(t_in)
    if b is None:
        AMAX_PASSES += 1
        if AMAX_LOG is not None:  # measurement: who consumes a tensor whose producer left no bound
            import traceback
            AMAX_LOG.append((tuple(t_in.shape), [f.name for f in traceback.extract_stack()[-7:-2]]))
        b = amax(t_in if t_planes is None else t_planes)
        set_bound(t_in, b)
    return b


def amax(x, out=None):
    """max |x| of a (n, c, h, w) tensor (plane-contiguous images) as a 1-element device tensor: conv2d's `x_amax`.  `out`: an
    existing bound to fold this tensor into (max of both)."""
    require_gpu(x)
    x = _as_planes(x)
    n, c, h, w = x.shape
    if out is None:
        out = _arena(x.device).slot(x.device)  # (zeroed; examined by the overflow guard: the kernel keeps non-finite values sticky)
    _run('amax', lambda: _lib.check(_lib.lib().edvr_amax_f32(_ptr(x), _ptr(out), n, c * h * w, _img_stride(x), _stream()), 'edvr_amax_f32'),
         0, _nb(x))
    return out


def conv2d(x1, wpk, bias, co, ks, *, x2=None, x2_map=None, stride=1, act=ACT_NONE, act_from=0, res1=None, res2=None,
           out_mode=OUT_NCHW, out=None, algo=None, gate=None, gate_slope=0.0, y_scale=1.0, wpk_f4=None, abs_sum_channels=0,
           wpk_f4s=None, x_amax=None, want_y_amax=True):
    """y = y_scale * act(conv(cat(x1, x2)) + bias) + res1 + res2 on the fp32 MFMA kernel.
    algo: CONV_AUTO (default; module-level CONV_ALGO overrides it, used by tests), CONV_DIRECT, CONV_WINOGRAD or CONV_WINOGRAD_F4.
    wpk_f4: pack_conv_weight(w, f4=True) - allows the F(4x4,3x3) Winograd kernel (inference; ~1e-6 relative rounding error).

    x2_map = (div, mul, add): image i of x2 is (i // div) * mul + add (broadcast of a reference frame).
    gate (n, co, ho, wo): y *= gate > 0 ? 1 : gate_slope - the backward of a ReLU / LeakyReLU fused into the data-gradient conv
    (3x3 kernels; in the Winograd kernel's epilogue where that kernel applies, else in the direct kernel's).
    y_scale: ResidualBlockNoBN's res_scale (arch_util.py:95), 3x3 kernels only.
    abs_sum_channels > 0: returns (y, stats) with stats (2, n) = abs_stats_per_image(y[:, :abs_sum_channels]) (sums of |y| and of the
    horizontal neighbour differences) - the sums in the conv's own epilogue where the launch runs on the F(4x4) kernel
    (edvr_conv2d_desc.abs_sum), the differences estimated from the first image; by the separate reduction kernel otherwise.
    """
    require_gpu(x1, x2, wpk, bias, res1, res2)
    L = _lib.lib()
    x1_in, x2_in, y_bound = x1, x2, None
    x1 = _as_planes(x1)
    n, c1, h, w = x1.shape
    d = _lib.ConvDesc()
    d.x1, d.c1, d.x1_img_stride = _ptr(x1), c1, _img_stride(x1)
    if x2 is not None:
        x2 = _as_planes(x2)
        d.x2, d.c2, d.x2_img_stride = _ptr(x2), x2.shape[1], _img_stride(x2)
        assert x2.shape[2:] == x1.shape[2:]
        if x2_map is not None:
            d.x2_div, d.x2_mul, d.x2_add = x2_map
        else:
            assert x2.shape[0] == n
    d.n, d.h, d.w = n, h, w
    d.wpk, d.bias = _ptr(wpk), _ptr(bias)
    d.co, d.ks, d.stride = co, ks, stride
    d.act, d.act_from = act, act_from
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    if out is None:
        shape = (n, co // 4, 2 * ho, 2 * wo) if out_mode == OUT_PIXEL_SHUFFLE2 else (n, co, ho, wo)
        out = torch.empty(shape, dtype=torch.float32, device=x1.device)
    else:
        assert _plane_contig(out)
    for name, r in (('res1', res1), ('res2', res2)):
        if r is not None:
            r = _as_planes(r)
            assert tuple(r.shape) == (n, co, ho, wo), f'{name} shape {tuple(r.shape)}'
            setattr(d, name, _ptr(r))
            setattr(d, name + '_img_stride', _img_stride(r))
            if name == 'res1':
                res1 = r
            else:
                res2 = r
    if gate is not None:
        gate = _as_planes(gate)
        assert tuple(gate.shape) == (n, co, ho, wo), f'gate shape {tuple(gate.shape)}'
        d.gate, d.gate_img_stride, d.gate_slope = _ptr(gate), _img_stride(gate), float(gate_slope)
    d.y, d.y_img_stride, d.out_mode = _ptr(out), _img_stride(out), out_mode
    d.y_scale = float(y_scale)
    if wpk_f4 is not None:
        require_gpu(wpk_f4)
        d.wpk_f4 = _ptr(wpk_f4)
    d.algo = CONV_ALGO if algo is None else algo
    split = False
    if wpk_f4s is not None:  # the split-operand F(4x4) kernel: its weights + a bound of the input's magnitude
        require_gpu(wpk_f4s, dtypes=(torch.int32,))
        d.wpk_f4s = _ptr(wpk_f4s)
        d.x_amax = _ptr(wpk_f4s)  # (any non-null pointer: only asks whether the launch would run on that kernel)
        split = bool(L.edvr_conv2d_y_amax_supported(ctypes.byref(d)))
        if split:
            if x_amax is None:
                x_amax = input_bound(x1_in, x1)
                if x2 is not None:
                    x_amax = torch.maximum(x_amax, input_bound(x2_in, x2))
            require_gpu(x_amax)
            if BOUND_CHECK:  # debugging / tests: every bound that reaches the kernel is compared with the data (synchronises)
                true = amax(x1).item() if x2 is None else max(amax(x1).item(), amax(x2).item())
                if not (x_amax.item() >= true):
                    raise AssertionError(f'magnitude bound {x_amax.item():.6g} < max |x| = {true:.6g} for a conv input of shape {tuple(x1.shape)}')
                BOUND_CHECK_LOG.append((tuple(x1.shape), x_amax.item() / max(true, 1e-30)))
            d.x_amax = _ptr(x_amax)
            if want_y_amax:
                y_bound = _arena(x1.device).slot(x1.device)
                d.y_amax = _ptr(y_bound)
        else:
            d.wpk_f4s, d.x_amax = None, None
    sums = None
    if abs_sum_channels > 0 and L.edvr_conv2d_abs_sum_supported(ctypes.byref(d)):
        sums = torch.zeros(2, n, dtype=torch.float32, device=x1.device)
        d.abs_sum, d.abs_sum_channels = _ptr(sums), int(abs_sum_channels)
    name, flops, nbytes, executed = 'conv2d', 0.0, 0.0, None
    if LAUNCH_HOOK is not None:  # measurement only (bench.py): brackets the launch with events on this stream
        buf = ctypes.create_string_buffer(96)
        L.edvr_conv2d_kernel_name(ctypes.byref(d), buf, 96)
        name = buf.value.decode()
        ex = ctypes.c_double(0.0)
        if L.edvr_conv2d_executed_flops(ctypes.byref(d), ctypes.byref(ex)) == 0:
            executed = ex.value
        flops = 2.0 * n * ho * wo * co * (c1 + d.c2) * ks * ks
        # algorithmic HBM bytes: every input / residual / gate / output element once, plus the weights
        nbytes = 4.0 * (n * (c1 + d.c2) * h * w + n * co * ho * wo * (1 + (res1 is not None) + (res2 is not None) + (gate is not None))
                        + co * (c1 + d.c2) * ks * ks)
    _run(name, lambda: _lib.check(L.edvr_conv2d_f32(ctypes.byref(d), _stream()), 'edvr_conv2d_f32'), flops, nbytes, executed)
    if y_bound is not None:
        set_bound(out, y_bound)
    else:
        void_bound(out)  # a caller-provided buffer rewritten by a kernel without the epilogue: its old bound is void
    if abs_sum_channels > 0:
        if sums is None:
            return out, abs_stats_per_image(out[:, :abs_sum_channels])
        # the epilogue took the sums of |y|; the roughness statistic (row 1) is an ESTIMATE from up to four images spread over the batch
        # (one pass over ~4 x 30 MB on the L1 layer), each scaled to stand for its neighbours - one more accumulator in the F(4x4)
        # kernels' staging waves spills (128 registers) and slows every layer of the network by 4 %
        step = max(1, n // 4)
        sel = out[::step, :abs_sum_channels]
        r = abs_stats_per_image(sel)
        sums[1, ::step] = r[1] * (float(n) / sel.shape[0])
        return out, sums
    return out


def conv_gate_supported(n, c, h, w, co, algo=None):
    """Would conv2d(..., gate=...) of a 3x3 / stride-1 conv on (n, c, h, w) -> co channels run in a Winograd kernel's fused
    epilogue?  (C-side rules: sizes, algorithm request, EDVR_CONV_WINOGRAD=0 switch.  The direct kernel accepts a gate as well -
    also together with residuals - but the residual block only records the fused form where it is the fast one.)"""
    d = _lib.ConvDesc()
    d.c1, d.n, d.h, d.w, d.co, d.ks, d.stride = c, n, h, w, co, 3, 1
    d.algo = CONV_ALGO if algo is None else algo
    return bool(_lib.lib().edvr_conv2d_gate_supported(ctypes.byref(d)))


# ------------------------------------------------------------------------------------------------ DCNv1
def _ones_mask(offset):
    return torch.ones(offset.shape[0], offset.shape[1] // 2, offset.shape[2], offset.shape[3], dtype=offset.dtype, device=offset.device)


def dcnv1_forward(x, offset, weight, stride, pad, dil, groups, dg, halo_hint=0, out=None):
    """DeformConv forward (no mask, no bias): edvr_dcnv1_fwd_f32; float64 / float16: DCNv2 with an all-ones mask on edvr_dcnv2_fwd_any.
    out: optional preallocated contiguous output written in place."""
    if require_gpu(x, offset, weight, dtypes=tuple(DCN_DTYPES)) != torch.float32:
        return dcnv2_forward(x, offset, _ones_mask(offset), weight, None, stride, pad, dil, groups, dg, out=out)
    L = _lib.lib()
    offset = _as_planes(offset)
    dims = _dcn_dims(x, weight, stride, pad, dil, groups, dg)
    B, C, H, W, Co, kh, kw = dims[:7]
    ho, wo = offset.shape[2], offset.shape[3]
    if out is not None:
        if tuple(out.shape) != (B, Co, ho, wo) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise RuntimeError(f'out must be a contiguous float32 tensor of shape {(B, Co, ho, wo)} on {x.device}')
        y = out
    else:
        y = torch.empty(B, Co, ho, wo, dtype=torch.float32, device=x.device)
    nbytes = L.edvr_dcnv1_fwd_ws_bytes(*dims)
    ws = workspace(nbytes, x.device)
    _run('dcnv1_fwd', lambda: _lib.check(L.edvr_dcnv1_fwd_f32(_ptr(x), _ptr(offset), _ptr(weight), _ptr(y), *dims, _bstride(offset), halo_hint, _ptr(ws), nbytes,
                                    _stream()),
                                       'edvr_dcnv1_fwd_f32'), 2.0 * B * Co * ho * wo * weight.shape[1] * kh * kw, _nb(x, offset, weight, y))
    void_bound(y)  # (a caller's `out` buffer: whatever bound it carried described its old contents)
    return y


def dcnv1_backward(x, offset, weight, dy, stride, pad, dil, groups, dg, scatter_hint=0):
    """Returns (dx, doffset, dweight) of DeformConv: edvr_dcnv1_bwd_f32 (float64 / float16: through edvr_dcnv2_bwd_any)."""
    if require_gpu(x, offset, weight, dy, dtypes=tuple(DCN_DTYPES)) != torch.float32:
        dx, doff, _, dw, _ = dcnv2_backward(x, offset, _ones_mask(offset), weight, dy, False, stride, pad, dil, groups, dg)
        return dx, doff, dw
    L = _lib.lib()
    offset = _as_planes(offset)
    dy = dy.contiguous()
    dims = _dcn_dims(x, weight, stride, pad, dil, groups, dg)
    dx = torch.empty_like(x)
    doff = torch.empty(offset.shape, dtype=torch.float32, device=x.device)
    dw = torch.empty_like(weight)
    nbytes = L.edvr_dcnv1_bwd_ws_bytes(*dims)
    ws = workspace(nbytes, x.device)
    _run('dcnv1_bwd', lambda: _lib.check(L.edvr_dcnv1_bwd_f32(_ptr(x), _ptr(offset), _ptr(weight), _ptr(dy), _ptr(dx), _ptr(doff), _ptr(dw), *dims,
                                    _bstride(offset), _bstride(doff), int(scatter_hint), _ptr(ws), nbytes, _stream()),
                                       'edvr_dcnv1_bwd_f32'), 4.0 * dy.numel() * weight[0].numel(),  # two GEMMs (dcol, dW)
                                       _nb(x, offset, weight, dy, dx, doff, dw))
    return dx, doff, dw


# ------------------------------------------------------------------------------------------------ DCNv2
def _hw(v):
    """int or (h, w) pair -> (h, w)"""
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _enc_hw(v):
    """EDVR_HW of include/edvr_amd.h: one int for an (h, w) pair; a square pair stays the plain value."""
    h, w = _hw(v)
    return h if h == w else (h | ((w + 1) << 16))


def _dcn_dims(x, weight, stride, pad, dil, groups, dg):
    B, C, H, W = x.shape
    Co, cig, kh, kw = weight.shape
    return [B, C, H, W, Co, kh, kw, _enc_hw(stride), _enc_hw(pad), _enc_hw(dil), groups, dg]


def _bstride(t):
    """offset/mask may be channel slices of one conv_offset output: pass their image stride."""
    return t.stride(0) if t.shape[0] > 1 else 0


def dcnv2_forward(x, offset, mask, weight, bias, stride, pad, dil, groups, dg, act=ACT_NONE, halo_hint=0, out=None, xm_bound=None):
    """out: optional preallocated (B, Co, Ho, Wo) contiguous tensor the kernels write in place (the reference's `output` argument,
    deform_conv_cuda.cpp:530-568).
    halo_hint: which kernel class runs (performance only).  The classes compute the same operator in different summation orders:
    their results agree to fp32 rounding (<= 2e-5 of the output scale, tests/test_gpu_dcn.py), NOT bit for bit.
    xm_bound: 1-element device tensor >= max |x| * max(1, max |mask|) (for sigmoid masks: a bound of |x|, `input_bound(x)`): allows the
    split-operand form of the tap-window kernel (csrc/dcn_tapwin_s.hip) where that kernel applies; None = the fp32 kernels."""
    dt = require_gpu(x, offset, mask, weight, bias, dtypes=tuple(DCN_DTYPES))
    L = _lib.lib()
    if not x.is_contiguous() or not weight.is_contiguous():
        raise RuntimeError('input tensor has to be contiguous')  # deform_conv_cuda.cpp:497-498
    offset, mask = _as_planes(offset), _as_planes(mask)
    dims = _dcn_dims(x, weight, stride, pad, dil, groups, dg)
    B, C, H, W, Co, kh, kw = dims[:7]
    if C != weight.shape[1] * groups:
        raise RuntimeError(f'Input shape and kernel channels wont match: ({C} vs {weight.shape[1] * groups}).')
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(pad), _hw(dil)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if Ho <= 0 or Wo <= 0:
        raise ValueError(f'convolution input is too small (output would be {Ho}x{Wo})')
    if tuple(offset.shape[1:]) != (dg * 2 * kh * kw, Ho, Wo):  # (RuntimeError like the reference's TORCH_CHECKs, deform_conv_cuda.cpp:64-66)
        raise RuntimeError(f'invalid spatial size / channels of offset: got {tuple(offset.shape)}, expected (B, {dg * 2 * kh * kw}, {Ho}, {Wo})')
    if tuple(mask.shape[1:]) != (dg * kh * kw, Ho, Wo):
        raise RuntimeError(f'invalid spatial size / channels of mask: got {tuple(mask.shape)}, expected (B, {dg * kh * kw}, {Ho}, {Wo})')
    if out is not None:
        if tuple(out.shape) != (B, Co, Ho, Wo) or out.dtype != dt or not out.is_contiguous() or out.device != x.device:
            raise RuntimeError(f'out must be a contiguous {dt} tensor of shape {(B, Co, Ho, Wo)} on {x.device}')
        y = out
    else:
        y = torch.empty(B, Co, Ho, Wo, dtype=dt, device=x.device)
    if dt != torch.float32:  # float64 / float16: the reference's other dispatch legs (csrc/dcn_any.hip); no fused activation there
        if act != ACT_NONE:
            raise NotImplementedError('the fused activation of dcnv2_forward exists in fp32 only')
        nbytes = L.edvr_dcnv2_any_ws_bytes(DCN_DTYPES[dt], *dims)
        ws = workspace(nbytes, x.device)
        _lib.check(L.edvr_dcnv2_fwd_any(DCN_DTYPES[dt], _ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(bias), _ptr(y), *dims,
                                        _bstride(offset), _bstride(mask), _ptr(ws), nbytes, _stream()), 'edvr_dcnv2_fwd_any')
        void_bound(y)
        return y
    nbytes = L.edvr_dcnv2_fwd_ws_bytes(*dims)
    ws = workspace(nbytes, x.device)
    name = 'dcnv2_fwd'
    split = xm_bound is not None and bool(L.edvr_dcnv2_fwd_split_applies(_ptr(x), *dims, int(halo_hint)))
    if LAUNCH_HOOK is not None:  # measurement only (bench.py): which kernel class this call runs on
        buf = ctypes.create_string_buffer(64)
        if L.edvr_dcnv2_fwd_kernel_name(_ptr(x), *dims, int(halo_hint), buf, 64) == 0:
            name = f'dcnv2_fwd[{"dcn_tapwin_split_fwd_kernel" if split else buf.value.decode()}]'
    common = (_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(bias), _ptr(y), *dims, _bstride(offset), _bstride(mask), act, halo_hint, _ptr(ws), nbytes)
    if split:
        require_gpu(xm_bound)
        launch = lambda: _lib.check(L.edvr_dcnv2_fwd_split_f32(*common, _ptr(xm_bound), _stream()), 'edvr_dcnv2_fwd_split_f32')
    else:
        launch = lambda: _lib.check(L.edvr_dcnv2_fwd_f32(*common, _stream()), 'edvr_dcnv2_fwd_f32')
    _run(name, launch, 2.0 * B * Co * Ho * Wo * weight.shape[1] * kh * kw, _nb(x, offset, mask, weight, y))
    void_bound(y)  # (a caller's `out` buffer: its old bound is void; the module path attaches the new one, dcn.py)
    return y


def dcnv2_backward(x, offset, mask, weight, dy, with_bias, stride, pad, dil, groups, dg, doffset=None, dmask=None,
                   scatter_hint=0, xm_bound=None, dy_bound=None):
    """Returns (dx, doffset, dmask, dweight, dbias).  `doffset` / `dmask` may be preallocated channel slices of one
    buffer (image-strided views): the kernels write them in place.
    xm_bound / dy_bound (both or neither): 1-element device tensors >= max |x| * max(1, max |mask|) and >= max |dy| - the dW product
    then runs in its split-operand form (csrc/gemm_nt_s.hip)."""
    dt = require_gpu(x, offset, mask, weight, dy, dtypes=tuple(DCN_DTYPES))
    L = _lib.lib()
    offset, mask = _as_planes(offset), _as_planes(mask)
    dy = dy.contiguous()
    dims = _dcn_dims(x, weight, stride, pad, dil, groups, dg)
    dx = torch.empty_like(x)
    doff = doffset if doffset is not None else torch.empty(offset.shape, dtype=dt, device=x.device)
    dmsk = dmask if dmask is not None else torch.empty(mask.shape, dtype=dt, device=x.device)
    assert _plane_contig(doff) and _plane_contig(dmsk)
    dw = torch.empty_like(weight)
    db = torch.empty(weight.shape[0], dtype=dt, device=x.device) if with_bias else None
    if dt != torch.float32:
        nbytes = L.edvr_dcnv2_any_ws_bytes(DCN_DTYPES[dt], *dims)
        ws = workspace(nbytes, x.device)
        _lib.check(L.edvr_dcnv2_bwd_any(DCN_DTYPES[dt], _ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(dy), _ptr(dx), _ptr(doff),
                                        _ptr(dmsk), _ptr(dw), _ptr(db), *dims, _bstride(offset), _bstride(mask), _bstride(doff),
                                        _bstride(dmsk), _ptr(ws), nbytes, _stream()), 'edvr_dcnv2_bwd_any')
        void_bound(doff, dmsk)
        return dx, doff, dmsk, dw, db
    nbytes = L.edvr_dcnv2_bwd_ws_bytes(*dims)
    ws = workspace(nbytes, x.device)
    common = (_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(dy), _ptr(dx), _ptr(doff), _ptr(dmsk), _ptr(dw), _ptr(db), *dims,
              _bstride(offset), _bstride(mask), _bstride(doff), _bstride(dmsk), int(scatter_hint), _ptr(ws), nbytes)
    if xm_bound is not None and dy_bound is not None and L.edvr_dcnv2_bwd_split_applies():
        require_gpu(xm_bound, dy_bound)
        launch = lambda: _lib.check(L.edvr_dcnv2_bwd_split_f32(*common, _ptr(xm_bound), _ptr(dy_bound), _stream()), 'edvr_dcnv2_bwd_split_f32')
    else:
        launch = lambda: _lib.check(L.edvr_dcnv2_bwd_f32(*common, _stream()), 'edvr_dcnv2_bwd_f32')
    _run('dcnv2_bwd' + _SCATTER_NAMES.get(int(scatter_hint), ''), launch, 4.0 * dy.numel() * weight[0].numel(),
         _nb(x, offset, mask, weight, dy, dx, doff, dmsk, dw))
    void_bound(doff, dmsk)
    return dx, doff, dmsk, dw, db


# ------------------------------------------------------------------------------------------------ glue kernels
def tsa_temporal(emb, emb_ref, aligned, want_prob=False):
    """emb/aligned (b, t, c, h, w), emb_ref (b, c, h, w) -> aligned * sigmoid(<emb_t, emb_ref>) and optionally prob."""
    require_gpu(emb, emb_ref, aligned)
    b, t, c, h, w = aligned.shape
    al_in = aligned
    emb, emb_ref, aligned = emb.contiguous(), emb_ref.contiguous(), aligned.contiguous()
    out = torch.empty_like(aligned)
    prob = torch.empty(b, t, h, w, dtype=torch.float32, device=aligned.device) if want_prob else None
    _run('tsa_temporal', lambda: _lib.check(_lib.lib().edvr_tsa_temporal_f32(_ptr(emb), _ptr(emb_ref), _ptr(aligned), _ptr(out), _ptr(prob), b, t, c, h * w,
                                                _stream()),
                                       'edvr_tsa_temporal_f32'), 0, _nb(emb, emb_ref, aligned, out, prob))
    carry_bound(out, al_in)  # aligned * sigmoid(.)
    return (out, prob) if want_prob else out


def pool_maxavg(x):
    require_gpu(x)
    x_in, x = x, x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty(n, 2 * c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, dtype=torch.float32, device=x.device)
    _run('pool_maxavg_3x3s2', lambda: _lib.check(_lib.lib().edvr_pool_maxavg_3x3s2_f32(_ptr(x), _ptr(y), n, c, h, w, _stream()),
                                       'edvr_pool_maxavg_3x3s2_f32'), 0, _nb(x, y))
    return carry_bound(y, x_in)


def upsample2x(x, scale=1.0):
    require_gpu(x)
    x_in, x = x, x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty(n, c, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    _run('upsample2x', lambda: _lib.check(_lib.lib().edvr_upsample2x_f32(_ptr(x), _ptr(y), n * c, h, w, float(scale), _stream()),
                                       'edvr_upsample2x_f32'), 0, _nb(x, y))
    return carry_bound(y, x_in, scale=abs(scale))  # bilinear weights sum to one


def tsa_combine(feat, attn, attn_add):
    require_gpu(feat, attn, attn_add)
    f_in, a_in = feat, attn_add
    feat, attn, attn_add = feat.contiguous(), attn.contiguous(), attn_add.contiguous()
    y = torch.empty_like(feat)
    _run('tsa_combine', lambda: _lib.check(_lib.lib().edvr_tsa_combine_f32(_ptr(feat), _ptr(attn), _ptr(attn_add), _ptr(y), feat.numel(), _stream()),
                                       'edvr_tsa_combine_f32'), 0, _nb(feat, attn, attn_add, y))
    bf, ba = get_bound(f_in), get_bound(a_in)
    if bf is not None and ba is not None:
        set_bound(y, bf * 2.0 + ba, max(_bound_depth(f_in), _bound_depth(a_in)))  # feat * sigmoid(attn) * 2 + attn_add
    return y


def upsample4x_add_(y, base):
    """y += bilinear_x4(base), in place."""
    require_gpu(y, base)
    base = base.contiguous()
    n, c, h, w = base.shape
    assert y.is_contiguous() and tuple(y.shape) == (n, c, 4 * h, 4 * w)
    by, bb = get_bound(y), get_bound(base)  # before the write: |y + up(base)| <= bound(y) + bound(base) (bilinear weights sum to one)
    depth = max(_bound_depth(y), _bound_depth(base)) if (by is not None and bb is not None) else 0
    _run('upsample4x_add', lambda: _lib.check(_lib.lib().edvr_upsample4x_add_f32(_ptr(base), _ptr(y), n * c, h, w, _stream()),
                                       'edvr_upsample4x_add_f32'), 0, _nb(base, y, y))
    void_bound(y)  # the raw-pointer write does not move y's version counter: the old bound must not survive it
    if by is not None and bb is not None:
        set_bound(y, by + bb, depth)
    return y


def add(a, b):
    require_gpu(a, b)
    a_in, b_in = a, b
    a, b = a.contiguous(), b.contiguous()
    assert a.shape == b.shape
    y = torch.empty_like(a)
    _run('add', lambda: _lib.check(_lib.lib().edvr_add_f32(_ptr(a), _ptr(b), _ptr(y), a.numel(), _stream()),
                                       'edvr_add_f32'), 0, _nb(a, b, y))
    return carry_bound(y, a_in, b_in)


def act_backward(dy, y, act, act_from=0, res1=None, res2=None):
    """dz = dy * act'(.) computed from the activation output; y = act(z) + res1 + res2 as the fused conv wrote it."""
    require_gpu(dy, y, res1, res2)
    dy_in = dy
    dy, y = dy.contiguous(), y.contiguous()
    res1 = res1.contiguous() if res1 is not None else None
    res2 = res2.contiguous() if res2 is not None else None
    n, c = y.shape[:2]
    dz = torch.empty_like(dy)
    _run('act_bwd', lambda: _lib.check(_lib.lib().edvr_act_bwd_f32(_ptr(dy), _ptr(y), _ptr(res1), _ptr(res2), _ptr(dz), n, c, y[0, 0].numel(), act, act_from,
                                           _stream()),
                                       'edvr_act_bwd_f32'), 0, _nb(dy, y, res1, res2, dz))
    return carry_bound(dz, dy_in)  # |act'| <= 1 for every epilogue activation


def set_wgrad_algo(algo):
    """Process-wide algorithm request of conv2d_wgrad (CONV_AUTO | CONV_DIRECT | CONV_WINOGRAD); returns the previous one."""
    prev = _lib.lib().edvr_conv2d_wgrad_algo(int(algo))
    if prev < 0:
        _lib.check(prev, 'edvr_conv2d_wgrad_algo')
    return prev


def conv2d_wgrad(x1, x2, x2_map, dz, co, ks, stride, want_db=False):
    """dW (co, c1+c2, ks, ks) of the fused conv: Winograd-domain GEMM over tiles (3x3 / stride 1) or fp32 MFMA implicit GEMM
    over the pixel axis.  want_db: also return db = sum of dz over (n, h, w) -> (dw, db); the Winograd-domain kernel produces it
    in the same two launches."""
    require_gpu(x1, x2, dz)
    L = _lib.lib()
    x1_in, x2_in, dz_in = x1, x2, dz
    x1, dz = _as_planes(x1), _as_planes(dz)
    n, c1, h, w = x1.shape
    c2 = 0
    if x2 is not None:
        x2 = _as_planes(x2)
        c2 = x2.shape[1]
    div, mul, add = x2_map if x2_map is not None else (0, 0, 0)
    dw = torch.empty(co, c1 + c2, ks, ks, dtype=torch.float32, device=x1.device)
    nbytes = L.edvr_conv2d_wgrad_ws_bytes(n, c1 + c2, h, w, co, ks, stride)
    ws = workspace(nbytes, x1.device)
    db = torch.empty(co, dtype=torch.float32, device=x1.device) if want_db else None
    name = 'conv2d_wgrad'
    # the split-operand form of the Winograd-domain kernel needs bounds of both operands' magnitudes (they travel with the tensors)
    split = F4S_TRAINING and bool(L.edvr_conv2d_wgrad_split_applies(n, c1, c2, h, w, co, ks, stride))
    if LAUNCH_HOOK is not None:
        buf = ctypes.create_string_buffer(96)
        L.edvr_conv2d_wgrad_kernel_name(n, c1, c2, h, w, co, ks, stride, buf, 96)
        name = buf.value.decode()
        if split and name == 'conv3x3_winograd_wgrad_kernel':
            name = 'conv3x3_wgrad_direct_split_kernel' if L.edvr_conv2d_wgrad_split_is_direct(h, w) else 'conv3x3_winograd_wgrad_split_kernel'
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    common = (_ptr(x1), _ptr(x2), _ptr(dz), _ptr(dw), c1, c2, n, h, w, co, ks, stride, _img_stride(x1), _img_stride(x2) if x2 is not None else 0,
              div, mul, add, _img_stride(dz), 0, _ptr(db), _ptr(ws), nbytes)
    if split and ks == 1:  # the 1x1 GEMM is too cheap to pay for a reduction pass: split only with both bounds at hand
        bx, bz = get_bound(x1_in), get_bound(dz_in)
        split = bx is not None and bz is not None
        if split and name == 'gemm_nt_kernel':
            name = 'gemm_nt_split_kernel'
    elif split:
        bx = input_bound(x1_in, x1)
        if x2 is not None:
            bx = torch.maximum(bx, input_bound(x2_in, x2))
        bz = input_bound(dz_in, dz)
    if split:
        launch = lambda: _lib.check(L.edvr_conv2d_wgrad_split_f32(*common, _ptr(bx), _ptr(bz), _stream()), 'edvr_conv2d_wgrad_split_f32')
    else:
        launch = lambda: _lib.check(L.edvr_conv2d_wgrad_f32(*common, _stream()), 'edvr_conv2d_wgrad_f32')
    _run(name, launch, 2.0 * n * ho * wo * co * (c1 + c2) * ks * ks, _nb(x1, dz, dw) + (4.0 * n * c2 * h * w if x2 is not None else 0.0))
    return (dw, db) if want_db else dw


def channel_sum(x):
    require_gpu(x)
    x = _as_planes(x)
    n, c, h, w = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = 64 * c * 4
    ws = workspace(nbytes, x.device)
    _run('channel_sum', lambda: _lib.check(_lib.lib().edvr_channel_sum_f32(_ptr(x), _ptr(out), n, c, h * w, _img_stride(x), _ptr(ws), nbytes, _stream()),
                                       'edvr_channel_sum_f32'), 0, _nb(x))
    return out


def pixel_unshuffle2(x):
    require_gpu(x)
    x_in, x = x, x.contiguous()
    n, c, h2, w2 = x.shape
    y = torch.empty(n, 4 * c, h2 // 2, w2 // 2, dtype=torch.float32, device=x.device)
    _run('pixel_unshuffle2', lambda: _lib.check(_lib.lib().edvr_pixel_unshuffle2_f32(_ptr(x), _ptr(y), n, c, h2 // 2, w2 // 2, _stream()),
                                       'edvr_pixel_unshuffle2_f32'), 0, _nb(x, y))
    return carry_bound(y, x_in)


def pixel_unshuffle2_act_backward(dy, y, act):
    """unshuffle(dy * act'(y)): the gradient of PixelShuffle(2)(act(z)) w.r.t. z in one launch (y = the shuffled activation output)."""
    require_gpu(dy)
    dy_in, dy = dy, dy.contiguous()
    n, c, h2, w2 = dy.shape
    if act != ACT_NONE:
        require_gpu(y)
        y = y.contiguous()
        if tuple(y.shape) != tuple(dy.shape):
            raise RuntimeError(f'pixel_unshuffle2_act_backward: y {tuple(y.shape)} vs dy {tuple(dy.shape)}')
    dz = torch.empty(n, 4 * c, h2 // 2, w2 // 2, dtype=torch.float32, device=dy.device)
    _run('pixel_unshuffle2_act_bwd', lambda: _lib.check(_lib.lib().edvr_pixel_unshuffle2_act_bwd_f32(
        _ptr(dy), _ptr(y) if act != ACT_NONE else None, _ptr(dz), n, c, h2 // 2, w2 // 2, int(act), _stream()), 'edvr_pixel_unshuffle2_act_bwd_f32'),
        0, _nb(dy, dz) + (_nb(y) if act != ACT_NONE else 0.0))
    return carry_bound(dz, dy_in)


def zero_stuff2(dz, H, W):
    require_gpu(dz)
    dz_in, dz = dz, dz.contiguous()
    n, c, ho, wo = dz.shape
    z = torch.empty(n, c, H, W, dtype=torch.float32, device=dz.device)
    _run('zero_stuff2', lambda: _lib.check(_lib.lib().edvr_zero_stuff2_f32(_ptr(dz), _ptr(z), n * c, H, W, ho, wo, _stream()),
                                       'edvr_zero_stuff2_f32'), 0, _nb(dz, z))
    return carry_bound(z, dz_in)


def frame_reduce_add_(src, dst, t, center):
    """dst[b, center] += sum_t src[b, t] for (b*t, c, h, w) tensors."""
    require_gpu(src, dst)
    src = src.contiguous()
    assert dst.is_contiguous() and src.shape == dst.shape and src.shape[0] % t == 0
    _run('frame_reduce_add', lambda: _lib.check(_lib.lib().edvr_frame_reduce_add_f32(_ptr(src), _ptr(dst), src.shape[0] // t, t, center, src[0].numel(), _stream()),
                                       'edvr_frame_reduce_add_f32'), 0, _nb(src) + 8.0 * src.numel() / t)
    void_bound(dst)  # rewritten in place through a raw pointer: the consumer measures
    return dst


def upsample2x_backward(dy, scale=1.0):
    require_gpu(dy)
    dy_in, dy = dy, dy.contiguous()
    n, c, h2, w2 = dy.shape
    dx = torch.empty(n, c, h2 // 2, w2 // 2, dtype=torch.float32, device=dy.device)
    _run('upsample2x_bwd', lambda: _lib.check(_lib.lib().edvr_upsample2x_bwd_f32(_ptr(dy), _ptr(dx), n * c, h2 // 2, w2 // 2, float(scale), _stream()),
                                       'edvr_upsample2x_bwd_f32'), 0, _nb(dy, dx))
    return carry_bound(dx, dy_in, scale=4.0 * abs(scale))  # adjoint of the interpolation: the weights into one coarse pixel sum to 4


def pool_maxavg_backward(x, dy):
    require_gpu(x, dy)
    dy_in = dy
    x, dy = x.contiguous(), dy.contiguous()
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    _run('pool_maxavg_3x3s2_bwd', lambda: _lib.check(_lib.lib().edvr_pool_maxavg_3x3s2_bwd_f32(_ptr(x), _ptr(dy), _ptr(dx), n, c, h, w, _stream()),
                                       'edvr_pool_maxavg_3x3s2_bwd_f32'), 0, _nb(x, dy, dx))
    return carry_bound(dx, dy_in, scale=8.0)  # a pixel sits in at most 4 windows of each of the two pooled halves


def tsa_temporal_backward(emb, emb_ref, aligned, dout):
    require_gpu(emb, emb_ref, aligned, dout)
    emb, emb_ref, aligned, dout = emb.contiguous(), emb_ref.contiguous(), aligned.contiguous(), dout.contiguous()
    b, t, c, h, w = aligned.shape
    d_emb, d_ref, d_al = torch.empty_like(emb), torch.empty_like(emb_ref), torch.empty_like(aligned)
    ws = torch.empty(b * t * h * w, dtype=torch.float32, device=emb.device)  # per-(clip, frame, pixel) scalar between the two passes
    _run('tsa_temporal_bwd', lambda: _lib.check(_lib.lib().edvr_tsa_temporal_bwd_f32(_ptr(emb), _ptr(emb_ref), _ptr(aligned), _ptr(dout), _ptr(d_emb), _ptr(d_ref),
                                                    _ptr(d_al), b, t, c, h * w, _ptr(ws), _stream()),
                                       'edvr_tsa_temporal_bwd_f32'), 0, _nb(emb, emb_ref, aligned, dout, d_emb, d_ref, d_al))
    return d_emb, d_ref, d_al


def tsa_combine_backward(feat, attn, dy):
    require_gpu(feat, attn, dy)
    feat, attn, dy = feat.contiguous(), attn.contiguous(), dy.contiguous()
    dfeat, dattn = torch.empty_like(feat), torch.empty_like(attn)
    _run('tsa_combine_bwd', lambda: _lib.check(_lib.lib().edvr_tsa_combine_bwd_f32(_ptr(feat), _ptr(attn), _ptr(dy), _ptr(dfeat), _ptr(dattn), feat.numel(), _stream()),
                                       'edvr_tsa_combine_bwd_f32'), 0, _nb(feat, attn, dy, dfeat, dattn))
    return dfeat, dattn


def charbonnier(pred, target, eps=1e-12, want_grad=True, grad_scale=1.0):
    """CharbonnierLoss(reduction='sum'): returns (loss (1,) device tensor, dloss/dpred or None) from one pass."""
    require_gpu(pred, target)
    pred, target = pred.contiguous(), target.contiguous()
    assert pred.shape == target.shape
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    _run('charbonnier', lambda: _lib.check(_lib.lib().edvr_charbonnier_f32(_ptr(pred), _ptr(target), _ptr(loss), _ptr(dpred), pred.numel(), float(eps),
                                               float(grad_scale), _stream()),
                                       'edvr_charbonnier_f32'), 0, _nb(pred, target, dpred))
    return loss, dpred


def note_abs_mean(x):
    """Start computing the offset statistics (mean |x| and the roughness, `offset_stats`) of a (n, c, h, w) float32 tensor without
    waiting for them: per-image sums -> pinned memory behind an event.  `abs_mean_if_ready` / `offset_stats_if_ready` return the
    values once they have arrived (None before) - how a backward picks up the statistic of its own forward without a host
    synchronisation.  EDVR_DCN_HINT_WAIT=1 makes the pick-up wait for the copy instead (deterministic kernel choice, for
    bisecting timings or rounding-level differences; costs one host synchronisation per backward)."""
    sums = abs_stats_per_image(x)
    host = torch.empty(sums.shape, dtype=sums.dtype, pin_memory=True)
    host.copy_(sums, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return host, done, x.numel()


HINT_WAIT = os.environ.get('EDVR_DCN_HINT_WAIT', '0') == '1'


def offset_stats_if_ready(rec):
    """(mean |x|, roughness or None) of a `note_abs_mean` record, or None while the copy has not landed."""
    if rec is None:
        return None
    if HINT_WAIT:
        rec[1].synchronize()
    elif not rec[1].query():
        return None
    return offset_stats(rec[0], rec[2])


def abs_mean_if_ready(rec):
    st = offset_stats_if_ready(rec)
    return None if st is None else st[0]


def offset_stats(stats, count):
    """(mean |offset|, mean |horizontal neighbour difference| or None) from the (2, n) sums of `abs_stats_per_image` over `count`
    elements in all.  The difference sum covers 3 of every 4 horizontal pairs (include/edvr_amd.h, edvr_abs_stats_f32)."""
    tot = stats.double().sum(-1).tolist() if stats.dim() == 2 else [float(stats.double().sum()), -1.0]
    return tot[0] / count, (tot[1] / (0.75 * count) if tot[1] >= 0 else None)


def abs_sum_per_image(x):
    """sum |x[i]| per image of a (n, c, h, w) tensor (image-strided views allowed)."""
    require_gpu(x)
    x = _as_planes(x)
    n, c, h, w = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    _run('abs_sum', lambda: _lib.check(_lib.lib().edvr_abs_sum_f32(_ptr(x), _ptr(out), n, c * h * w, _img_stride(x), _stream()),
                                       'edvr_abs_sum_f32'), 0, _nb(x))
    return out


def abs_stats_per_image(x):
    """(2, n): row 0 = sum |x[i]| per image, row 1 = sum |x[i, c, r, col] - x[i, c, r, col + 1]| over 3 of every 4 horizontal
    neighbour pairs (edvr_abs_stats_f32; -1 where the rows are not whole 16-byte groups) of a (n, c, h, w) tensor."""
    require_gpu(x)
    x = _as_planes(x)
    n, c, h, w = x.shape
    out = torch.empty(2, n, dtype=torch.float32, device=x.device)
    _run('abs_sum', lambda: _lib.check(_lib.lib().edvr_abs_stats_f32(_ptr(x), _ptr(out), n, c * h * w, w, _img_stride(x), _stream()),
                                       'edvr_abs_stats_f32'), 0, _nb(x))
    return out


# ------------------------------------------------------------------------------------------------ input pipeline
AUG_HFLIP, AUG_VFLIP, AUG_ROT90 = 1, 2, 4


def frames_u8_to_f32(frames_u8, flags=None, swap_rb=False):
    """uint8 device tensor (n_clips, frames, h, w, 3) -> float32 (n_clips, frames, 3, h', w') = bytes / 255, with the per-clip
    augmentation `flags` (bytes / sequence of n_clips OR-ed AUG_* values, host side) applied: hflip, vflip, then transpose, as
    transforms.augment does; the channel order is reversed when swap_rb (edvr_frames_u8_to_f32, csrc/data.hip)."""
    if not frames_u8.is_cuda:
        raise NotImplementedError('edvr_amd ops run on the GPU only (HIP/gfx950); got a CPU tensor')
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 5 or frames_u8.shape[-1] != 3:
        raise ValueError(f'expected a uint8 (n_clips, frames, h, w, 3) tensor, got {frames_u8.dtype} {tuple(frames_u8.shape)}')
    frames_u8 = frames_u8.contiguous()
    n, f, h, w, _ = frames_u8.shape
    fl = None
    if flags is not None:
        fl = bytes(flags)
        if len(fl) != n:
            raise ValueError(f'{len(fl)} augmentation flags for {n} clips')
    rot = fl is not None and any(b & AUG_ROT90 for b in fl)
    out = torch.empty((n, f, 3, w, h) if rot else (n, f, 3, h, w), dtype=torch.float32, device=frames_u8.device)
    if n:
        _lib.check(_lib.lib().edvr_frames_u8_to_f32(_ptr(frames_u8), _ptr(out), n, f, h, w, fl, int(bool(swap_rb)), _stream()),
                   'edvr_frames_u8_to_f32')
    return out
