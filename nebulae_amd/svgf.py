"""Host-side mirror of the reference's ``Neb::SVGFDenoiser`` over the HIP C ABI.

Method names follow /root/reference/src/SVGFDenoiser.h:11-93 (snake_case): ``init``,
``resize``, ``begin_frame``, ``end_frame``, ``reset_history``,
``submit_temporal_accumulation``, ``submit_atrous_compute_wavelet`` and the
``get_*`` accessors.  A D3D12 command list becomes a HIP stream handle (int, 0 = the
null stream): every ``submit_*`` call only enqueues work.  Error behaviour: the
reference throws (ThrowIfFailed / asserts, nri/stdafx.h:44-98); this mirror raises
``NebError``.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (PLANE_DEPTH, PLANE_MOMENTS, PLANE_NORMAL, PLANE_RADIANCE, PLANE_SCRATCH, PLANE_VARIANCE,  # noqa: F401
                   PLANE_ALBEDO, PLANE_ROUGH_METAL, PLANE_WORLDPOS, PLANE_LDR, PLANE_GEOMETRY, SLOT_CURRENT, SLOT_HISTORY, NebError)

# plane -> (numpy dtype, channels)
PLANE_LAYOUT = {
    PLANE_RADIANCE: (np.float32, 4), PLANE_NORMAL: (np.float16, 4), PLANE_DEPTH: (np.uint32, 1),
    PLANE_MOMENTS: (np.float16, 2), PLANE_VARIANCE: (np.float16, 1), PLANE_SCRATCH: (np.float32, 4),
    PLANE_ALBEDO: (np.uint32, 1), PLANE_ROUGH_METAL: (np.float16, 2), PLANE_WORLDPOS: (np.float16, 4),
    PLANE_LDR: (np.uint32, 1), PLANE_GEOMETRY: (np.float32, 4),
}


class SVGFDenoiser:
    NUM_ATROUS_PASSES = 4  # SVGFDenoiser.h:199

    def __init__(self):
        self._lib = None
        self._ctx = C.c_void_p(None)
        self.width = self.height = 0
        self.row_begin = self.row_end = 0
        self.levels = self.NUM_ATROUS_PASSES
        self.device = 0

    # ---- lifecycle (SVGFDenoiser.cpp:14-37) ----
    def is_initialized(self):
        return bool(self._ctx)

    def init(self, width, height, atrous_levels=None, device=0, row_begin=0, row_end=0):
        if self.is_initialized():
            raise NebError("SVGFDenoiser.init: already initialised")  # NEB_ASSERT(!IsInitialized())
        self._lib = _lib.load()
        self.levels = self.NUM_ATROUS_PASSES if atrous_levels is None else int(atrous_levels)
        info = _lib.CreateInfo(device, width, height, row_begin, row_end, self.levels)
        ctx = C.c_void_p(None)
        rc = self._lib.neb_create(C.byref(info), C.byref(ctx))
        _lib.check(self._lib, None, rc, "neb_create")
        self._ctx = ctx
        # like the binding of INTEGRATION.md: this class orders its work on the planes through neb_* calls only, so it opts into
        # the held-back temporal pass (the two submit calls back to back = one fused chain).  The raw C ABI's default is 0.
        self.set_option("svgf_fuse", 1)
        self.device = int(device)
        self.width, self.height = width, height
        self.row_begin, self.row_end = row_begin, (row_end or height)
        return True

    def resize(self, width, height):
        self._check(self._lib.neb_resize(self._ctx, width, height), "neb_resize")
        self.width, self.height = width, height
        self.row_begin, self.row_end = 0, height
        return True  # (the reference returns false on success, SVGFDenoiser.cpp:36 -- a bug we do not copy)

    def destroy(self):
        if self._ctx:
            self._lib.neb_destroy(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def _check(self, rc, what):
        _lib.check(self._lib, self._ctx, rc, what)

    # ---- per-frame bracket (SVGFDenoiser.cpp:39-47) ----
    def begin_frame(self, frame_index):
        self._check(self._lib.neb_begin_frame(self._ctx, frame_index & 0xFFFFFFFF), "neb_begin_frame")

    def end_frame(self):
        self._check(self._lib.neb_end_frame(self._ctx), "neb_end_frame")

    def get_current_resource_index(self):
        return self._lib.neb_current_index(self._ctx)

    def get_history_resource_index(self):
        return self._lib.neb_history_index(self._ctx)

    # ---- tunables (SVGFDenoiser.h:76-93) ----
    def get_constants(self):
        p = _lib.SvgfParams()
        self._check(self._lib.neb_svgf_get_params(self._ctx, C.byref(p)), "neb_svgf_get_params")
        return {n: getattr(p, n) for n, _ in p._fields_}

    def set_constants(self, **kw):
        p = _lib.SvgfParams()
        self._check(self._lib.neb_svgf_get_params(self._ctx, C.byref(p)), "neb_svgf_get_params")
        for k, v in kw.items():
            if not hasattr(p, k):
                raise NebError(f"unknown SVGF constant {k}")
            setattr(p, k, float(v))
        self._check(self._lib.neb_svgf_set_params(self._ctx, C.byref(p)), "neb_svgf_set_params")

    def set_option(self, key, value):
        self._check(self._lib.neb_set_option(self._ctx, key.encode(), int(value)), "neb_set_option")

    # ---- resources (the getters of SVGFDenoiser.h:24-70) ----
    def get_plane(self, plane, slot=SLOT_CURRENT):
        """-> (device pointer of resident row `row_begin`, pitch in bytes, resident rows)."""
        d, pitch, rows = C.c_void_p(), C.c_size_t(), C.c_uint32()
        self._check(self._lib.neb_get_plane(self._ctx, plane, slot, C.byref(d), C.byref(pitch), C.byref(rows)),
                    "neb_get_plane")
        return d.value, pitch.value, rows.value

    def _host_shape(self, plane, nrows):
        dt, ch = PLANE_LAYOUT[plane]
        return dt, ((nrows, self.width, ch) if ch > 1 else (nrows, self.width))

    def upload(self, plane, slot, array, row0=None, stream=0):
        row0 = self.row_begin if row0 is None else row0
        dt, shape = self._host_shape(plane, array.shape[0])
        a = np.ascontiguousarray(array, dtype=dt).reshape(shape)
        self._check(self._lib.neb_upload_rows(self._ctx, plane, slot, row0, a.shape[0], a.ctypes.data_as(C.c_void_p),
                                              C.c_void_p(stream)), "neb_upload_rows")
        self.synchronize(stream)

    def download(self, plane, slot=SLOT_CURRENT, row0=None, nrows=None, stream=0):
        row0 = self.row_begin if row0 is None else row0
        nrows = (self.row_end - row0) if nrows is None else nrows
        dt, shape = self._host_shape(plane, nrows)
        out = np.empty(shape, dt)
        self._check(self._lib.neb_download_rows(self._ctx, plane, slot, row0, nrows, out.ctypes.data_as(C.c_void_p),
                                                C.c_void_p(stream)), "neb_download_rows")
        self.synchronize(stream)
        return out

    def plane_tensor(self, plane, slot=SLOT_CURRENT):
        """Zero-copy torch view [rows, W, C] of a resident plane (device memory stays owned by
        the context).  16-bit float planes come back as float16, depth/albedo as int32."""
        import torch
        ptr, _, rows = self.get_plane(plane, slot)
        dt, ch = PLANE_LAYOUT[plane]
        typestr = {np.float32: "<f4", np.float16: "<f2", np.uint32: "<i4"}[dt]
        shape = (rows, self.width, ch) if ch > 1 else (rows, self.width)

        class _Holder:
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2,
                                      "strides": None}
        t = torch.as_tensor(h, device=f"cuda:{self.device}")  # (the context's own device, whatever torch's current one is)
        t._neb_owner = self  # keep the context alive as long as the view
        return t

    def synchronize(self, stream=0):
        self._check(self._lib.neb_stream_synchronize(self._ctx, C.c_void_p(stream)), "neb_stream_synchronize")

    # ---- passes (enqueue only) ----
    def reset_history(self, stream=0):
        self._check(self._lib.neb_svgf_reset_history(self._ctx, C.c_void_p(stream)), "neb_svgf_reset_history")

    def submit_temporal_accumulation(self, stream=0, rows=None):
        if rows is None:
            rc = self._lib.neb_svgf_temporal(self._ctx, C.c_void_p(stream))
        else:
            rc = self._lib.neb_svgf_temporal_rows(self._ctx, rows[0], rows[1], C.c_void_p(stream))
        self._check(rc, "neb_svgf_temporal")

    def submit_atrous_compute_wavelet(self, stream=0):
        self._check(self._lib.neb_svgf_atrous(self._ctx, C.c_void_p(stream)), "neb_svgf_atrous")

    def submit_denoising(self, stream=0):
        """neb_svgf_denoise: both passes as one explicit call (the fused chain where the context allows it)."""
        self._check(self._lib.neb_svgf_denoise(self._ctx, C.c_void_p(stream)), "neb_svgf_denoise")

    def level_times(self):
        """Durations (us) of the kernels of the last submit_atrous_compute_wavelet chain (option svgf_profile = 1): entry 0 is
        level 0 -- fused with the temporal pass when the two calls ran as one chain.  Option svgf_profile = 2: [the first kernel,
        all the others as one interval]."""
        out = (C.c_float * 32)()
        n = C.c_uint32()
        self._check(self._lib.neb_svgf_level_times(self._ctx, out, 32, C.byref(n)), "neb_svgf_level_times")
        return [float(out[k]) for k in range(n.value)]

    def submit_atrous_level(self, level, rows, stream=0):
        self._check(self._lib.neb_svgf_atrous_level_rows(self._ctx, level, rows[0], rows[1], C.c_void_p(stream)),
                    "neb_svgf_atrous_level_rows")

    def atrous_level_planes(self, level):
        v = [C.c_int() for _ in range(4)]
        self._check(self._lib.neb_svgf_atrous_level_planes(self._ctx, level, *[C.byref(x) for x in v]),
                    "neb_svgf_atrous_level_planes")
        return (v[0].value, v[1].value), (v[2].value, v[3].value)
