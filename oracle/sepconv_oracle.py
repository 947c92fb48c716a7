"""ctypes binding of oracle/sepconv_oracle.c -- TEST INFRASTRUCTURE ONLY.

The C file restates src/separable_convolution/cfile/SeparableConvolution_kernel.cu:19-162
of the reference loop for loop; this module only marshals numpy / CPU-torch
arrays into it.  ``build()`` compiles the library with gcc when it is missing.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsepconv_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    src = os.path.join(_HERE, "sepconv_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsepconv_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        i = ctypes.c_int
        for sfx in ("", "_f64"):
            getattr(lib, "sepconv_oracle_forward" + sfx).argtypes = [_F, _F, _F, _F, i, i, i, i, i]
            getattr(lib, "sepconv_oracle_grad_v" + sfx).argtypes = [_F, _F, _F, _F, i, i, i, i, i]
            getattr(lib, "sepconv_oracle_grad_h" + sfx).argtypes = [_F, _F, _F, _F, i, i, i, i, i]
            getattr(lib, "sepconv_oracle_grad_i" + sfx).argtypes = [_F, _F, _F, _F, i, i, i, i, i]
            getattr(lib, "sepconv_oracle_backward" + sfx).argtypes = [_F] * 7 + [i] * 5
            for n in ("forward", "grad_v", "grad_h", "grad_i", "backward"):
                getattr(lib, "sepconv_oracle_%s%s" % (n, sfx)).restype = None
        lib.sepconv_oracle_num_threads.restype = ctypes.c_int
        lib.sepconv_oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib = lib
    return _lib


def num_threads():
    return _load().sepconv_oracle_num_threads()


def set_num_threads(n):
    _load().sepconv_oracle_set_num_threads(int(n))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_F)


def _dims(inp, v, h, ks):
    B, C, Hp, Wp = inp.shape
    H, W = Hp - ks + 1, Wp - ks + 1
    assert v.shape == (B, ks, H, W) and h.shape == (B, ks, H, W), (inp.shape, v.shape, h.shape, ks)
    return B, C, H, W


def forward(inp, v, h, ks, f64=False):
    """out[b,c,y,x] = sum_fy sum_fx in[b,c,y+fy,x+fx] v[b,fy,y,x] h[b,fx,y,x]  (.cu:19-47)."""
    lib = _load()
    inp, pi = _f32(inp); v, pv = _f32(v); h, ph = _f32(h)
    B, C, H, W = _dims(inp, v, h, ks)
    out = np.empty((B, C, H, W), np.float32)
    fn = lib.sepconv_oracle_forward_f64 if f64 else lib.sepconv_oracle_forward
    fn(pi, pv, ph, out.ctypes.data_as(_F), B, C, H, W, ks)
    return out


def backward(gO, inp, v, h, ks, f64=False):
    """(gI, gV, gH) of the reference's three backward kernels (.cu:49-162, launcher :187-242)."""
    lib = _load()
    gO, pg = _f32(gO); inp, pi = _f32(inp); v, pv = _f32(v); h, ph = _f32(h)
    B, C, H, W = _dims(inp, v, h, ks)
    assert gO.shape == (B, C, H, W)
    gI = np.empty_like(inp); gV = np.empty_like(v); gH = np.empty_like(h)
    fn = lib.sepconv_oracle_backward_f64 if f64 else lib.sepconv_oracle_backward
    fn(pg, pi, pv, ph, gI.ctypes.data_as(_F), gV.ctypes.data_as(_F), gH.ctypes.data_as(_F),
       B, C, H, W, ks)
    return gI, gV, gH
