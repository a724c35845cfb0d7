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
