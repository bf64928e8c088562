"""ctypes binding of libedvr_amd.so (include/edvr_amd.h).

The library is the product: there is no Python/CPU fallback.  If it cannot be
loaded, every op raises - loudly - instead of silently running something else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('EDVR_AMD_LIB') or os.path.join(_HERE, 'lib', 'libedvr_amd.so')  # env: A/B kernel variants

c_float_p = ctypes.c_void_p
i32, i64, f32, sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
vp = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """Mirror of `edvr_conv2d_desc` (include/edvr_amd.h)."""
    _fields_ = [
        ('x1', vp), ('x2', vp), ('c1', i32), ('c2', i32), ('x1_img_stride', i64), ('x2_img_stride', i64),
        ('x2_div', i32), ('x2_mul', i32), ('x2_add', i32), ('n', i32), ('h', i32), ('w', i32), ('wpk', vp),
        ('bias', vp), ('co', i32), ('ks', i32), ('stride', i32), ('act', i32), ('act_from', i32), ('res1', vp),
        ('res2', vp), ('res1_img_stride', i64), ('res2_img_stride', i64), ('y', vp), ('y_img_stride', i64),
        ('out_mode', i32), ('algo', i32), ('gate', vp), ('gate_img_stride', i64), ('gate_slope', f32), ('y_scale', f32),
        ('wpk_f4', vp), ('abs_sum', vp), ('abs_sum_channels', i32), ('wpk_f4s', vp), ('x_amax', vp), ('y_amax', vp),
    ]


# name -> (restype, argtypes); every symbol include/edvr_amd.h declares
PROTOTYPES = {
    'edvr_version': (ctypes.c_char_p, []),
    'edvr_last_error': (ctypes.c_char_p, []),
    'edvr_check_device': (i32, []),
    'edvr_conv2d_packed_weight_elems': (sz, [i32, i32, i32]),
    'edvr_conv2d_pack_weight_f32': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'edvr_conv2d_packed_weight_f4_elems': (sz, [i32, i32]),
    'edvr_conv2d_pack_weight_f4_f32': (i32, [vp, vp, i32, i32, i32, vp]),
    'edvr_conv2d_packed_weight_f4s_elems': (sz, [i32, i32]),
    'edvr_conv2d_pack_weight_f4s_f32': (i32, [vp, vp, i32, i32, i32, vp]),
    'edvr_conv2d_packed_weight_1x1s_elems': (sz, [i32, i32]),
    'edvr_conv2d_pack_weight_1x1s_f32': (i32, [vp, vp, i32, i32, vp]),
    'edvr_amax_f32': (i32, [vp, vp, i32, i64, i64, vp]),
    'edvr_pack_job_bytes': (sz, []),
    'edvr_conv2d_pack_weights_multi': (i32, [vp, i32, i32, i32, vp]),
    'edvr_conv2d_f32': (i32, [ctypes.POINTER(ConvDesc), vp]),
    'edvr_conv2d_gate_supported': (i32, [ctypes.POINTER(ConvDesc)]),
    'edvr_conv2d_abs_sum_supported': (i32, [ctypes.POINTER(ConvDesc)]),
    'edvr_conv2d_y_amax_supported': (i32, [ctypes.POINTER(ConvDesc)]),
    'edvr_conv2d_kernel_name': (i32, [ctypes.POINTER(ConvDesc), ctypes.c_char_p, sz]),
    'edvr_conv2d_executed_flops': (i32, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_double)]),
    'edvr_dcnv2_fwd_ws_bytes': (sz, [i32] * 12),
    'edvr_dcnv2_fwd_f32': (i32, [vp] * 6 + [i32] * 12 + [i64, i64, i32, i32, vp, sz, vp]),
    'edvr_dcnv2_fwd_split_f32': (i32, [vp] * 6 + [i32] * 12 + [i64, i64, i32, i32, vp, sz, vp, vp]),
    'edvr_dcnv2_fwd_split_applies': (i32, [vp] + [i32] * 13),
    'edvr_dcnv2_fwd_kernel_name': (i32, [vp] + [i32] * 13 + [ctypes.c_char_p, sz]),
    'edvr_dcnv2_bwd_ws_bytes': (sz, [i32] * 12),
    'edvr_dcnv2_any_ws_bytes': (sz, [i32] * 13),
    'edvr_dcnv2_fwd_any': (i32, [i32] + [vp] * 6 + [i32] * 12 + [i64, i64, vp, sz, vp]),
    'edvr_dcnv2_bwd_any': (i32, [i32] + [vp] * 10 + [i32] * 12 + [i64, i64, i64, i64, vp, sz, vp]),
    'edvr_psnr_sse_f32': (i32, [vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, vp]),
    'edvr_ssim_partials': (sz, [i32, i32, i32]),
    'edvr_ssim_f32': (i32, [vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, vp]),
    'edvr_frames_u8_to_f32': (i32, [vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    'edvr_adam_chunk_bytes': (sz, []),
    'edvr_adam_multi_f32': (i32, [vp, i32, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]),
    'edvr_dcnv1_fwd_ws_bytes': (sz, [i32] * 12),
    'edvr_dcnv1_fwd_f32': (i32, [vp] * 4 + [i32] * 12 + [i64, i32, vp, sz, vp]),
    'edvr_dcnv1_bwd_ws_bytes': (sz, [i32] * 12),
    'edvr_dcnv1_bwd_f32': (i32, [vp] * 7 + [i32] * 12 + [i64, i64, i32, vp, sz, vp]),
    'edvr_dcnv2_bwd_f32': (i32, [vp] * 10 + [i32] * 12 + [i64, i64, i64, i64, i32, vp, sz, vp]),
    'edvr_dcnv2_bwd_split_f32': (i32, [vp] * 10 + [i32] * 12 + [i64, i64, i64, i64, i32, vp, sz, vp, vp, vp]),
    'edvr_dcnv2_bwd_split_applies': (i32, []),
    'edvr_tsa_temporal_f32': (i32, [vp] * 5 + [i32] * 4 + [vp]),
    'edvr_pool_maxavg_3x3s2_f32': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'edvr_upsample2x_f32': (i32, [vp, vp, i32, i32, i32, f32, vp]),
    'edvr_tsa_combine_f32': (i32, [vp, vp, vp, vp, i64, vp]),
    'edvr_upsample4x_add_f32': (i32, [vp, vp, i32, i32, i32, vp]),
    'edvr_add_f32': (i32, [vp, vp, vp, i64, vp]),
    'edvr_act_bwd_f32': (i32, [vp, vp, vp, vp, vp, i32, i32, i64, i32, i32, vp]),
    'edvr_conv2d_wgrad_ws_bytes': (sz, [i32] * 7),
    'edvr_conv2d_wgrad_algo': (i32, [i32]),
    'edvr_conv2d_wgrad_kernel_name': (i32, [i32] * 8 + [ctypes.c_char_p, sz]),
    'edvr_conv2d_wgrad_f32': (i32, [vp] * 4 + [i32] * 8 + [i64, i64, i32, i32, i32, i64, i32, vp, vp, sz, vp]),
    'edvr_conv2d_wgrad_split_f32': (i32, [vp] * 4 + [i32] * 8 + [i64, i64, i32, i32, i32, i64, i32, vp, vp, sz, vp, vp, vp]),
    'edvr_conv2d_wgrad_split_applies': (i32, [i32] * 8),
    'edvr_conv2d_wgrad_split_is_direct': (i32, [i32] * 2),
    'edvr_channel_sum_f32': (i32, [vp, vp, i32, i32, i64, i64, vp, sz, vp]),
    'edvr_pixel_unshuffle2_f32': (i32, [vp, vp, i32, i32, i32, i32, vp]),
    'edvr_pixel_unshuffle2_act_bwd_f32': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    'edvr_zero_stuff2_f32': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'edvr_frame_reduce_add_f32': (i32, [vp, vp, i32, i32, i32, i64, vp]),
    'edvr_upsample2x_bwd_f32': (i32, [vp, vp, i32, i32, i32, f32, vp]),
    'edvr_pool_maxavg_3x3s2_bwd_f32': (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    'edvr_tsa_temporal_bwd_f32': (i32, [vp] * 7 + [i32] * 4 + [vp, vp]),
    'edvr_tsa_combine_bwd_f32': (i32, [vp] * 5 + [i64, vp]),
    'edvr_charbonnier_f32': (i32, [vp, vp, vp, vp, i64, f32, f32, vp]),
    'edvr_abs_sum_f32': (i32, [vp, vp, i32, i64, i64, vp]),
    'edvr_abs_stats_f32': (i32, [vp, vp, i32, i64, i32, i64, vp]),
}

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3
OUT_NCHW, OUT_PIXEL_SHUFFLE2 = 0, 1
CONV_AUTO, CONV_DIRECT, CONV_WINOGRAD, CONV_WINOGRAD_F4, CONV_WINOGRAD_F4S = 0, 1, 2, 3, 4
DTYPE_F32, DTYPE_F64, DTYPE_F16 = 0, 1, 2  # EDVR_DTYPE_*
DCN_HALO_TAPWIN = 16  # EDVR_DCN_HALO_TAPWIN

_lib = None


class ExtensionMissing(RuntimeError):
    pass


def lib():
    """Return the loaded library; raise ExtensionMissing if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissing(
                f'{LIB_PATH} not found: the HIP extension is not built (run `python -m edvr_amd.build`). '
                'edvr_amd has no CPU/PyTorch fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().edvr_last_error().decode()
        raise RuntimeError(f'{what} failed (code {rc}): {msg}')
