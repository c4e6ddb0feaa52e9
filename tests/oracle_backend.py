"""An oracle-backed stand-in for nebulae_amd.svgf.SVGFDenoiser, for CPU (gloo) tests of the multi-GPU
strip logic in nebulae_amd/strips.py.  Test infrastructure only: it lets the host-side partition /
halo-exchange code run without a GPU, with oracle/svgf_ref.c doing the per-row arithmetic."""
import ctypes as C

import numpy as np
import torch

from nebulae_amd.svgf import (PLANE_DEPTH, PLANE_MOMENTS, PLANE_NORMAL, PLANE_RADIANCE, PLANE_SCRATCH, PLANE_VARIANCE)
from oracle_lib import OracleSVGF, SvgfParams, lib


class OracleDenoiser:
    """Holds FULL-image planes on every rank but only ever touches its resident rows; rows outside
    [row_begin, row_end) are poisoned with NaN so that any read of a non-resident row shows up."""

    def __init__(self):
        self.o = None

    def init(self, width, height, atrous_levels=4, device=0, row_begin=0, row_end=0):
        self.width, self.height, self.levels = width, height, atrous_levels
        self.row_begin, self.row_end = row_begin, (row_end or height)
        self.o = OracleSVGF(width, height, atrous_levels)
        self.L = lib()
        self.L.svgf_ref_temporal.argtypes = [C.c_int] * 4 + [C.c_void_p] * 9 + [C.POINTER(SvgfParams)]
        self.L.svgf_ref_atrous.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int, C.POINTER(SvgfParams)]
        self.scratch = np.zeros((height, width, 4), np.float32)
        for arr in list(self.o.radiance) + [self.scratch]:
            arr[:self.row_begin] = np.nan
            arr[self.row_end:] = np.nan
        self.params = SvgfParams(0.002, 0.9, 1e-4, 4.0 / 255.0, 128.0, 0.002)
        return True

    def set_option(self, key, value):
        pass

    def begin_frame(self, f):
        self.o.begin_frame(f)

    def end_frame(self):
        pass

    def get_current_resource_index(self):
        return self.o.cur

    def _plane(self, plane, slot):
        if plane == PLANE_RADIANCE:
            return self.o.radiance[slot]
        if plane == PLANE_SCRATCH:
            return self.scratch
        if plane == PLANE_DEPTH:
            return self.o.depth[slot]
        if plane == PLANE_NORMAL:
            return self.o.normal[slot]
        if plane == PLANE_MOMENTS:
            return self.o.moments[slot]
        if plane == PLANE_VARIANCE:
            return self.o.variance
        raise KeyError(plane)

    def plane_tensor(self, plane, slot=-1):
        if slot == -1:
            slot = self.o.cur
        arr = self._plane(plane, slot)[self.row_begin:self.row_end]
        if arr.dtype == np.uint32:
            arr = arr.view(np.int32)
        return torch.from_numpy(arr)

    def reset_history(self, stream=0):
        self.o.reset_history()

    def submit_temporal_accumulation(self, stream=0, rows=None):
        r0, r1 = rows or (self.row_begin, self.row_end)
        o, c, h = self.o, self.o.cur, self.o.hist
        self.L.svgf_ref_temporal(self.width, self.height, r0, r1, o.radiance[c].ctypes.data, o.radiance[h].ctypes.data,
                                 o.depth[c].ctypes.data, o.depth[h].ctypes.data, o.normal[c].ctypes.data,
                                 o.normal[h].ctypes.data, o.moments[h].ctypes.data, o.moments[c].ctypes.data,
                                 o.variance.ctypes.data, C.byref(self.params))

    def atrous_level_planes(self, level):
        L, c, h = self.levels, self.o.cur, self.o.hist

        def node(i):
            if i == 0 or (i == L and L != 1):
                return (PLANE_RADIANCE, c)
            if L == 1:
                return (PLANE_SCRATCH, 0)
            if L % 2 == 0:
                return (PLANE_RADIANCE, h if i & 1 else c)
            return (PLANE_RADIANCE, h) if i & 1 else (PLANE_SCRATCH, 0)
        return node(level), node(level + 1)

    def submit_atrous_level(self, level, rows, stream=0):
        (sp, ss), (dp, ds) = self.atrous_level_planes(level)
        src, dst = self._plane(sp, ss), self._plane(dp, ds)
        o = self.o
        self.L.svgf_ref_atrous(self.width, self.height, rows[0], rows[1], src.ctypes.data, dst.ctypes.data,
                               o.variance.ctypes.data, o.depth[o.cur].ctypes.data, o.normal[o.cur].ctypes.data,
                               1 << level, C.byref(self.params))

    def submit_atrous_compute_wavelet(self, stream=0):
        """all levels over the whole image (one strip = the whole frame: strips.py then makes the two whole-frame calls)"""
        for level in range(self.levels):
            self.submit_atrous_level(level, (0, self.height))
        if self.levels == 1:
            self.o.radiance[self.o.cur][...] = self.scratch

    def destroy(self):
        if self.o:
            self.o.close()
            self.o = None
