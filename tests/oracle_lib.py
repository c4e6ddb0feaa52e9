"""ctypes binding of oracle/libneb_oracle.so -- the CPU checker (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_LIB = None


class SvgfParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("depthSigma", "alpha", "varianceEps", "phiColor", "phiNormal", "phiDepth")]


class SvgfState(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("levels", C.c_int),
                ("radiance", C.POINTER(C.c_float) * 2), ("normal", C.POINTER(C.c_uint16) * 2),
                ("depth", C.POINTER(C.c_uint32) * 2), ("moments", C.POINTER(C.c_uint16) * 2),
                ("variance", C.POINTER(C.c_uint16)), ("scratch", C.POINTER(C.c_float)),
                ("cur", C.c_int), ("hist", C.c_int), ("params", SvgfParams), ("threads", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libneb_oracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".cpp", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(so)
        L.svgf_ref_create.restype = C.POINTER(SvgfState)
        L.svgf_ref_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.svgf_ref_destroy.argtypes = [C.POINTER(SvgfState)]
        for f in ("svgf_ref_reset_history", "svgf_ref_temporal_pass", "svgf_ref_atrous_pass"):
            getattr(L, f).argtypes = [C.POINTER(SvgfState)]
            getattr(L, f).restype = None
        L.svgf_ref_begin_frame.argtypes = [C.POINTER(SvgfState), C.c_uint32]
        L.svgf_ref_f32_to_f16.argtypes = [C.c_float]
        L.svgf_ref_f32_to_f16.restype = C.c_uint16
        L.svgf_ref_f16_to_f32.argtypes = [C.c_uint16]
        L.svgf_ref_f16_to_f32.restype = C.c_float
        _LIB = L
    return _LIB


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    arr = np.ctypeslib.as_array(ptr, shape=(n,))
    return arr.view(dtype).reshape(shape)


class OracleSVGF:
    """numpy-facing wrapper of svgf_ref_state (mirrors SVGFDenoiser's resource set)."""

    def __init__(self, W, H, levels=4, threads=1, params=None):
        self.L = lib()
        self.p = self.L.svgf_ref_create(W, H, levels)
        self.W, self.H, self.levels = W, H, levels
        s = self.p.contents
        s.threads = threads
        if params:
            for k, v in params.items():
                setattr(s.params, k, float(v))
        self.radiance = [_view(s.radiance[k], (H, W, 4), np.float32) for k in range(2)]
        self.normal = [_view(s.normal[k], (H, W, 4), np.float16) for k in range(2)]
        self.depth = [_view(s.depth[k], (H, W), np.uint32) for k in range(2)]
        self.moments = [_view(s.moments[k], (H, W, 2), np.float16) for k in range(2)]
        self.variance = _view(s.variance, (H, W), np.float16)

    cur = property(lambda self: self.p.contents.cur)
    hist = property(lambda self: self.p.contents.hist)

    def begin_frame(self, f):
        self.L.svgf_ref_begin_frame(self.p, f)

    def reset_history(self):
        self.L.svgf_ref_reset_history(self.p)

    def temporal_pass(self):
        self.L.svgf_ref_temporal_pass(self.p)

    def atrous_pass(self):
        self.L.svgf_ref_atrous_pass(self.p)

    def close(self):
        if self.p:
            self.L.svgf_ref_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
