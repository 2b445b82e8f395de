"""Loader of oracle/chain_arith.c (the reference's fp32 arithmetic restated in C).  TEST INFRASTRUCTURE ONLY.

``linear`` / ``matmul`` / ``sin`` / ``cos`` on numpy float32 arrays.  The library is compiled on first use with gcc (present in the
build container and on the GPU box) into oracle/libomds_chain.so; __graft_entry__.build() compiles it too."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "chain_arith.c")
_SO = os.path.join(_HERE, "libomds_chain.so")
_lib = None
F32 = np.float32


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        tmp = _SO + f".{os.getpid()}.tmp"
        subprocess.run(["gcc", "-O3", "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", tmp, "-lm"],
                       check=True)
        os.replace(tmp, _SO)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        vp, lg, it = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
        _lib.omds_orc_linear.argtypes = [vp, vp, vp, vp, lg, it, it]
        _lib.omds_orc_matmul.argtypes = [vp, vp, vp, lg, it, it]
        _lib.omds_orc_sin.argtypes = [vp, vp, lg]
        _lib.omds_orc_cos.argtypes = [vp, vp, lg]
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=F32)


def linear(x, W, b=None):
    """x [M, K] . W [N, K]^T (+ b): ascending-k fmaf chains from zero, bias last (= torch addmm for M >= 11)."""
    x, W = _c(x), _c(W)
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    y = np.empty((M, N), F32)
    bb = _c(b) if b is not None else None
    lib().omds_orc_linear(x.ctypes.data, W.ctypes.data, bb.ctypes.data if bb is not None else None, y.ctypes.data, M, K, N)
    return y


def matmul(g, W):
    """g [M, K] @ W [K, N]: ascending-k fmaf chains from zero (= torch mm)."""
    g, W = _c(g), _c(W)
    M, K = g.shape
    assert W.shape[0] == K
    out = np.empty((M, W.shape[1]), F32)
    lib().omds_orc_matmul(g.ctypes.data, W.ctypes.data, out.ctypes.data, M, K, W.shape[1])
    return out


def _un(fn, x):
    x = _c(x)
    y = np.empty_like(x)
    getattr(lib(), fn)(x.ctypes.data, y.ctypes.data, x.size)
    return y


def sin(x):
    return _un("omds_orc_sin", x)


def cos(x):
    return _un("omds_orc_cos", x)
