"""ctypes binding of oracle/libgrl_oracle.so (ORACLE -- test infrastructure only)."""
import ctypes as C
import os

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        from oracle.build import build
        _LIB = C.CDLL(build())
        _LIB.grl_oracle_chain_gemm.restype = C.c_int
        _LIB.grl_oracle_chain_gemm.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p] * 2 + [C.c_int]
        _LIB.grl_oracle_jpeg_info.restype = C.c_int
        _LIB.grl_oracle_jpeg_info.argtypes = [C.c_char_p, C.c_size_t] + [C.POINTER(C.c_int)] * 5
        _LIB.grl_oracle_jpeg_decode.restype = C.c_int
        _LIB.grl_oracle_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        _LIB.grl_oracle_jpeg_coefficients.restype = C.c_int
        _LIB.grl_oracle_jpeg_coefficients.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        _LIB.grl_oracle_jpeg_blocks.restype = C.c_int
        _LIB.grl_oracle_jpeg_blocks.argtypes = [C.c_char_p, C.c_size_t]
    return _LIB


def chain_gemm(a, w, mode=0, rn=None, cn=None, kblock=False):
    """Bit-exact model of grl_conv_gemm_f32 on dense fp32 operands a [M,K], w [N,K]."""
    a = np.ascontiguousarray(a, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    M, K = a.shape
    N = w.shape[0]
    y = np.empty((M, N), np.float32)
    if rn is not None:
        rn = np.ascontiguousarray(rn, np.float32)
        cn = np.ascontiguousarray(cn, np.float32)
    rc = _lib().grl_oracle_chain_gemm(
        a.ctypes.data, w.ctypes.data, y.ctypes.data, M, N, K, K, K, N, mode,
        rn.ctypes.data if rn is not None else None, cn.ctypes.data if cn is not None else None,
        1 if kblock else 0)
    if rc:
        raise RuntimeError('grl_oracle_chain_gemm failed: %d' % rc)
    return y


def jpeg_decode(data):
    """Baseline-JPEG bytes -> uint8 [H, W, 3] RGB: the C restatement of Pillow's Image.open(..).convert('RGB')
    (oracle/ref_c/jpeg_baseline.c).  Raises ValueError with the oracle's code for streams outside its scope."""
    lib = _lib()
    v = [C.c_int() for _ in range(5)]
    rc = lib.grl_oracle_jpeg_info(data, len(data), *[C.byref(x) for x in v])
    if rc:
        raise ValueError('jpeg oracle: header rejected (%d)' % rc)
    w, h = v[0].value, v[1].value
    out = np.empty((h, w, 3), np.uint8)
    rc = lib.grl_oracle_jpeg_decode(data, len(data), out.ctypes.data)
    if rc:
        raise ValueError('jpeg oracle: decode failed (%d)' % rc)
    return out


def jpeg_coefficients(data):
    """The quantised DCT coefficients of a baseline JPEG, int16 [blocks in scan order][64 natural order]."""
    lib = _lib()
    nb = lib.grl_oracle_jpeg_blocks(data, len(data))
    if nb <= 0:
        raise ValueError('jpeg oracle: header rejected')
    out = np.zeros((nb, 64), np.int16)
    rc = lib.grl_oracle_jpeg_coefficients(data, len(data), out.ctypes.data)
    if rc:
        raise ValueError('jpeg oracle: decode failed (%d)' % rc)
    return out
