"""ctypes binding of libgrl_hip.so (C ABI: include/grl_hip.h).

There is no fallback: if the shared library is missing or a call fails this
module raises.  Tensors cross the boundary as raw device pointers
(``tensor.data_ptr()``) plus sizes; the stream is torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GRL_HIP_LIB') or os.path.join(_HERE, 'libgrl_hip.so')     # (override: A/B builds, tools/gemm_ko.sh)

ABI_VERSION = 9       # = GRL_ABI_VERSION of include/grl_hip.h this binding was written against

EPI_AFFINE, EPI_NEGDOT, EPI_EUCLID, EPI_SQDIFF = 0, 1, 2, 3

_fp = C.c_void_p      # device float*
_i32 = C.c_int32
_i64 = C.c_int64


class GrlGemm(C.Structure):
    _fields_ = [(n, _fp) for n in ('a', 'w', 'y', 'scale', 'shift', 'res', 'gbias', 'rowscale',
                                   'rnorm', 'cnorm', 'stats')] + \
               [(n, _i32) for n in ('M', 'N', 'K', 'lda', 'ldw', 'ldy', 'ldres', 'rows_per_group',
                                    'relu', 'epilogue', 'conv', 'H', 'W', 'C', 'Ho', 'Wo', 'kh',
                                    'kw', 'stride', 'pad', 'math', 'out_f32', 'res_rows', 'res_gstride', 'kblock')] + \
               [('splitk_ws', _fp), ('splitk_ws_floats', _i64)] + \
               [(n, _fp) for n in ('bn_z', 'bn_mean', 'bn_invstd', 'bn_mscale', 'bn_mbeta', 'bn_bits')]


MATH_F32, MATH_BF16, MATH_BF16X3, MATH_BF16S = 0, 1, 3, 2


class GrlWgrad(C.Structure):
    _fields_ = [(n, _fp) for n in ('dz', 'x', 'dw', 'workspace')] + \
               [(n, _i32) for n in ('M', 'N', 'K', 'ldz', 'ldx', 'k_out', 'accumulate', 'conv', 'H', 'W',
                                    'C', 'Ho', 'Wo', 'kh', 'kw', 'stride', 'pad', 'math', 'in_bf16')]


_TAIL_FIELDS = [(n, _fp) for n in ('t2', 'w3', 'scale3', 'shift3', 'res', 'y', 'w1n', 'scale1n', 'shift1n', 'u')] + \
               [(n, _i32) for n in ('M', 'P', 'C4', 'Pn')]


class GrlBneckTail(C.Structure):
    _fields_ = _TAIL_FIELDS + [(n, _fp) for n in ('x0', 'wd', 'scaled', 'shiftd')] + [('Kd', _i32), ('reserved', _i32)]


class GrlBneckTailF32(C.Structure):
    _fields_ = _TAIL_FIELDS


class GrlPrepEntry(C.Structure):
    _fields_ = [('src', _fp), ('dst', _fp), ('base', _i64), ('strides', _i64 * 4), ('dims', _i32 * 4),
                ('tiled', _i32), ('out_bf16', _i32)]


class GrlJpegFrame(C.Structure):
    """include/grl_hip.h: one parsed JPEG frame (filled by grl_jpeg_parse on the host, read by the decode kernels)"""
    _fields_ = [('scan_off', C.c_uint32), ('scan_len', C.c_uint32), ('width', C.c_uint16), ('height', C.c_uint16),
                ('restart_interval', C.c_uint16), ('ncomp', C.c_uint8), ('hmax', C.c_uint8), ('vmax', C.c_uint8),
                ('rgb', C.c_uint8), ('hs', C.c_uint8 * 4), ('vs', C.c_uint8 * 4), ('tq', C.c_uint8 * 4),
                ('td', C.c_uint8 * 4), ('ta', C.c_uint8 * 4), ('tabset', C.c_uint16), ('pad_', C.c_uint8 * 8), ('q', (C.c_uint16 * 64) * 4),
                ('maxcode', (C.c_int32 * 18) * 4), ('valoff', (C.c_int32 * 18) * 4), ('vals', (C.c_uint8 * 256) * 4)]


GRL_EINVAL = -1
GRL_EUNSUPPORTED = -3

_SIGNATURES = {
    'grl_abi_version': ([], C.c_int),
    'grl_stream_wait_stream': ([_fp, _fp], C.c_int),
    'grl_conv_gemm_f32': ([C.POINTER(GrlGemm), _fp], C.c_int),
    'grl_conv_gemm_f32_stat_rows': ([C.POINTER(GrlGemm)], C.c_int),
    'grl_conv_gemm_f32_group': ([C.POINTER(GrlGemm), C.c_int, _fp], C.c_int),
    'grl_gemm_force_tile': ([C.c_int, C.c_int], C.c_int),
    'grl_conv_gemm_f32_workspace_floats': ([C.POINTER(GrlGemm)], _i64),
    'grl_gemm_bf16_tile_mode': ([C.c_int], C.c_int),
    'grl_pack_conv_weight': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_bn_fold': ([_fp, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp, C.c_int, _fp], C.c_int),
    'grl_stem_conv7x7': ([_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_stem_pack_weight': ([_fp, _fp, _fp], C.c_int),
    'grl_stem_pack_weight_bf16': ([_fp, _fp, _fp], C.c_int),
    'grl_maxpool3x3s2': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_group_mean': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp], C.c_int),
    'grl_gce_gate': ([_fp] * 8 + [C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_temporal_mean': ([_fp, _fp, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_sqdiff_mean': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_channel_atte': ([_fp, _fp, _fp, _fp, _i64, _fp, _fp, _i64, C.c_int, C.c_int, C.c_int,
                          C.c_int, _fp, _fp], C.c_int),
    'grl_add_strided': ([_fp, _fp, _fp, C.c_int, _i64, _i64, _fp], C.c_int),
    'grl_affine_l2norm': ([_fp, _fp, _fp, _fp, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_siamese_attn': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_mean_T': ([_fp, _fp, C.c_int, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_pair_verify': ([_fp] * 7 + [C.c_int] * 4 + [_fp], C.c_int),
    'grl_col_stats_rows': ([C.c_int], C.c_int),
    'grl_col_stats': ([_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_slab_sum': ([_fp, C.c_int, _i64, C.c_int, _fp, C.c_int, _fp], C.c_int),
    'grl_bn_stats_finalize': ([_fp, C.c_int, C.c_int, _i64, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float,
                               _fp, _fp, _fp, _fp, _fp, _fp], C.c_int),
    'grl_bn_finalize_apply_takes': ([C.c_int, C.c_int], C.c_int),
    'grl_bn_finalize_apply_mode': ([C.c_int], C.c_int),
    'grl_bn_finalize_apply': ([_fp, C.c_int, C.c_int, _i64, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _fp,
                               _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_bn_finalize_apply_bf16': ([_fp, C.c_int, C.c_int, _i64, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _fp,
                                    _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_bn_apply': ([_fp, _fp, _fp, _fp, _fp, _i64, C.c_int, C.c_int, _fp], C.c_int),
    'grl_bn_apply_centered': ([_fp, _fp, _fp, _fp, _fp, _fp, _i64, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_bn_bwd': ([_fp] * 11 + [C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp], C.c_int),
    'grl_bn_bwd_finish': ([_fp] * 9 + [C.c_int, _fp, C.c_int, C.c_int, _fp, C.c_int, _fp], C.c_int),
    'grl_bn_bwd_finish_bf16': ([_fp] * 9 + [C.c_int, _fp, C.c_int, C.c_int, _fp, C.c_int, _fp], C.c_int),
    'grl_relu_bwd': ([_fp, _fp, _fp, _i64, C.c_int, _fp], C.c_int),
    'grl_axpby': ([_fp, _fp, _fp, C.c_float, C.c_float, _i64, _fp], C.c_int),
    'grl_axpy_strided': ([_fp, _i64, _fp, _i64, C.c_int, _i64, C.c_float, C.c_int, _fp], C.c_int),
    'grl_transpose': ([_fp, _fp, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_pack_dgrad_weight': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_dilate2': ([_fp, _fp] + [C.c_int] * 9 + [_fp], C.c_int),
    'grl_maxpool3x3s2_bwd': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_bn_relu_maxpool3x3s2': ([_fp] * 6 + [C.c_int] * 4 + [_fp], C.c_int),
    'grl_maxpool3x3s2_bwd_idx': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_bn_relu_maxpool3x3s2_bf16': ([_fp] * 6 + [C.c_int] * 4 + [_fp], C.c_int),
    'grl_maxpool3x3s2_bwd_idx_bf16': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_stem_wgrad_workspace_floats': ([C.c_int, C.c_int, C.c_int], _i64),
    'grl_stem_wgrad': ([_fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_stem_im2col': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_wgrad_workspace_floats': ([C.POINTER(GrlWgrad)], _i64),
    'grl_conv_wgrad_f32': ([C.POINTER(GrlWgrad), _fp], C.c_int),
    'grl_gate_apply': ([_fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp], C.c_int),
    'grl_gate_bwd': ([_fp, _fp, _fp, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_add_rowbcast': ([_fp, _fp, _i64, _i64, _i64, C.c_float, C.c_int, _fp], C.c_int),
    'grl_sqdiff_bwd': ([_fp] * 5 + [C.c_int, C.c_int, C.c_int, _i64, C.c_int, _fp], C.c_int),
    'grl_catte_bwd': ([_fp, _i64, _fp, _i64, _fp, _fp, _fp, _i64, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_l2norm_bwd': ([_fp, _i64, _fp, _i64, _fp, _fp, C.c_int, C.c_int, _fp], C.c_int),
    'grl_siamese_attn_bwd': ([_fp, _fp, _fp, _i64, _fp, _i64, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, _fp], C.c_int),
    'grl_pair_sqdiff': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_pair_sqdiff_bwd': ([_fp] * 5 + [C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_oim_update': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp], C.c_int),
    'grl_softmax_ce': ([_fp, _i64, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _i64, _fp, _fp], C.c_int),
    'grl_oim_grad': ([_fp, _i64, _fp, _fp, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_triplet_fwd': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp, _fp, _fp], C.c_int),
    'grl_triplet_bwd': ([_fp] * 5 + [C.c_int, C.c_float, _fp, C.c_int, C.c_int, _fp], C.c_int),
    'grl_softmax2': ([_fp, _fp, _fp, _i64, _fp], C.c_int),
    'grl_softmax2_bwd': ([_fp, _fp, _fp, _fp, _i64, _fp], C.c_int),
    'grl_normalize_u8': ([_fp, _fp, _fp, C.c_int, _i64, _fp], C.c_int),
    'grl_resize_bilinear_u8': ([_fp, _fp, _fp, _fp, C.c_int, _fp, _fp, C.c_int, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_augment_normalize_u8': ([_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_stem_conv7x7_u8': ([_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_stem_conv7x7_u8_bf16': ([_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp],
                                 C.c_int),
    'grl_pair_bce': ([_fp, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp], C.c_int),
    'grl_scale_dev': ([_fp, _fp, C.c_float, _fp, _i64, _fp], C.c_int),
    'grl_rank_metrics': ([_fp, _i64, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp], C.c_int),
    'grl_rerank_build': ([_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp], C.c_int),
    'grl_rerank_krecip': ([_fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp], C.c_int),
    'grl_rerank_expand': ([_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp], C.c_int),
    'grl_rerank_jaccard': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp], C.c_int),
    'grl_row_argsort': ([_fp, _i64, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_row_argsort_workspace_bytes': ([C.c_int, C.c_int], _i64),
    'grl_row_argsort_wide': ([_fp, _i64, C.c_int, C.c_int, _fp, _fp, _fp], C.c_int),
    'grl_cast_bf16': ([_fp, _fp, _i64, _fp], C.c_int),
    'grl_stem_conv7x7_bf16': ([_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_maxpool3x3s2_bf16': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_group_mean_bf16': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp], C.c_int),
    'grl_sqdiff_mean_bf16': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_gce_gate_bf16': ([_fp] * 8 + [C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_temporal_mean_bf16': ([_fp, _fp, C.c_int, C.c_int, _i64, _fp], C.c_int),
    'grl_add_strided_bf16': ([_fp, _fp, _fp, C.c_int, _i64, _i64, _fp], C.c_int),
    'grl_row_sqnorm': ([_fp, _fp, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    # bf16-storage training twins (train_bf16.hip)
    'grl_bn_apply_centered_bf16': ([_fp, _fp, _fp, _fp, _fp, _fp, _i64, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_col_stats_bf16': ([_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_bn_bwd_bf16': ([_fp] * 11 + [C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp], C.c_int),
    'grl_relu_bwd_bf16': ([_fp, _fp, _fp, _i64, C.c_int, _fp], C.c_int),
    'grl_axpby_bf16': ([_fp, _fp, _fp, C.c_float, C.c_float, _i64, _fp], C.c_int),
    'grl_axpy_strided_bf16': ([_fp, _i64, _fp, _i64, C.c_int, _i64, C.c_float, C.c_int, _fp], C.c_int),
    'grl_dilate2_bf16': ([_fp, _fp] + [C.c_int] * 9 + [_fp], C.c_int),
    'grl_maxpool3x3s2_bwd_bf16': ([_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_stem_im2col_bf16': ([_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_gate_apply_bf16': ([_fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp], C.c_int),
    'grl_gate_bwd_bf16': ([_fp, _fp, _fp, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_add_rowbcast_bf16': ([_fp, _fp, _i64, _i64, _i64, C.c_float, C.c_int, C.c_int, _fp], C.c_int),
    'grl_sqdiff_bwd_bf16': ([_fp] * 5 + [C.c_int, C.c_int, C.c_int, _i64, C.c_int, _fp], C.c_int),
    'grl_cast_f32': ([_fp, _fp, _i64, _fp], C.c_int),
    'grl_weight_prep': ([_fp, C.c_int, _fp], C.c_int),
    # cross-layer fusion (fuse_bf16.hip)
    'grl_bottleneck_tail_bf16': ([C.POINTER(GrlBneckTail), _fp], C.c_int),
    'grl_bottleneck_tail_bf16_supported': ([C.c_int, C.c_int, C.c_int], C.c_int),
    'grl_bneck_perm32': ([_fp, C.c_int, _fp, C.c_int, C.c_int, _fp], C.c_int),
    'grl_stem_pool_bf16': ([_fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_stem_pack_weight_pool': ([_fp, _fp, _fp], C.c_int),
    'grl_stem_pool_f32': ([_fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp], C.c_int),
    'grl_conv3x3_c64_bf16': ([_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'grl_bottleneck_tail_f32': ([C.POINTER(GrlBneckTailF32), _fp], C.c_int),
    'grl_bottleneck_tail_f32_supported': ([C.c_int, C.c_int, C.c_int], C.c_int),
    # frame decode on the device (jpeg.hip)
    'grl_jpeg_parse': ([_fp, _i64, _i64, C.POINTER(GrlJpegFrame)], C.c_int),
    'grl_jpeg_assign_tables': ([C.POINTER(GrlJpegFrame), C.c_int], C.c_int),
    'grl_jpeg_parallel_mode': ([C.c_int], C.c_int),
    'grl_jpeg_parse_batch': ([_fp, _fp, C.c_int, C.POINTER(GrlJpegFrame), _fp], C.c_int),
    'grl_jpeg_workspace_bytes': ([C.POINTER(GrlJpegFrame), C.c_int], _i64),
    'grl_jpeg_decode_batch': ([_fp, _fp, C.POINTER(GrlJpegFrame), C.c_int, _fp, _fp, _i64, _fp], C.c_int),
}

_lib = None


class GrlHipError(RuntimeError):
    pass


def load():
    """Load libgrl_hip.so (built in-tree by ``__graft_entry__.build()`` /
    ``make -C grl_amd/csrc``).  Raises if it is absent -- no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise GrlHipError(
            'libgrl_hip.so not found at %s: build it with `python -c "import '
            '__graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
            'grl_amd has no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.grl_last_error.restype = C.c_char_p
    lib.grl_last_error.argtypes = []
    for name, (args, res) in _SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError = header/library mismatch
        fn.argtypes = args
        fn.restype = res
    got = lib.grl_abi_version()
    if got != ABI_VERSION:                 # a stale .so would read a pointer as the stream / overrun a pack buffer
        raise GrlHipError('libgrl_hip.so at %s has ABI version %d, this grl_amd expects %d: rebuild it '
                          '(make -C grl_amd/csrc)' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def exported_symbols():
    return ['grl_last_error'] + sorted(_SIGNATURES)


def ptr(t):
    """Device pointer of a tensor (or None -> NULL)."""
    if t is None:
        return None
    if t.__class__ is int:               # a raw device address (e.g. a row of a BatchNorm state block: train_engine.bn_finalize)
        return t
    return t.data_ptr()


try:                                     # the raw handle of torch's current HIP stream without building a Stream object:
    _raw_stream, _get_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice     # ~0.3 us instead of ~4 us,
except AttributeError:                   # per launch, ~1500 launches per training step (the step is host-bound in bf16 storage)
    _raw_stream = None


def stream():
    if _raw_stream is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def check(rc, what=''):
    if rc != 0:
        msg = load().grl_last_error().decode('utf-8', 'replace')
        raise GrlHipError('%s failed (%d): %s' % (what, rc, msg))


def require_device(t, what='input', allow_u8=False):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise GrlHipError(
            '%s must live on a HIP device (got %s): grl_amd executes on MI355X only and '
            'has no CPU path' % (what, getattr(t, 'device', type(t))))
    if t.dtype != torch.float32 and not (allow_u8 and t.dtype == torch.uint8):
        raise GrlHipError('%s must be float32%s (got %s)' % (what, ' or raw uint8' if allow_u8 else '', t.dtype))
