"""ctypes binding of libnebulae_hip.so (include/nebulae_hip.h).

There is no CPU fallback: if the library is missing it is built with hipcc, and if
that fails, or a GPU call is made with no device, the error is raised to the caller.
"""
import ctypes as C
import os

from . import build as _build

NEB_OK = 0
PLANE_RADIANCE, PLANE_NORMAL, PLANE_DEPTH, PLANE_MOMENTS, PLANE_VARIANCE, PLANE_SCRATCH = 0, 1, 2, 3, 4, 5
PLANE_ALBEDO, PLANE_ROUGH_METAL, PLANE_WORLDPOS, PLANE_LDR, PLANE_GEOMETRY = 6, 7, 8, 9, 10
SLOT_CURRENT, SLOT_HISTORY = -1, -2


class NebError(RuntimeError):
    pass


class CreateInfo(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("row_begin", C.c_uint32),
                ("row_end", C.c_uint32), ("atrous_levels", C.c_uint32)]


class HaloPlane(C.Structure):
    _fields_ = [("plane", C.c_int32), ("slot", C.c_int32)]


class HaloSwap(C.Structure):
    _fields_ = [("peer", C.c_int32), ("send_row0", C.c_uint32), ("send_row1", C.c_uint32), ("recv_row0", C.c_uint32), ("recv_row1", C.c_uint32)]


class SvgfParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("depthSigma", "alpha", "varianceEps", "phiColor", "phiNormal", "phiDepth")]


_SIGS = {
    # name: (restype, argtypes)
    "neb_create": (C.c_int, [C.POINTER(CreateInfo), C.POINTER(C.c_void_p)]),
    "neb_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "neb_destroy": (C.c_int, [C.c_void_p]),
    "neb_last_error": (C.c_char_p, [C.c_void_p]),
    "neb_version": (C.c_char_p, []),
    "neb_marker_push": (C.c_int, [C.c_char_p]),
    "neb_marker_pop": (C.c_int, []),
    "neb_begin_frame": (C.c_int, [C.c_void_p, C.c_uint32]),
    "neb_end_frame": (C.c_int, [C.c_void_p]),
    "neb_current_index": (C.c_int, [C.c_void_p]),
    "neb_history_index": (C.c_int, [C.c_void_p]),
    "neb_svgf_default_params": (C.c_int, [C.POINTER(SvgfParams)]),
    "neb_svgf_set_params": (C.c_int, [C.c_void_p, C.POINTER(SvgfParams)]),
    "neb_svgf_get_params": (C.c_int, [C.c_void_p, C.POINTER(SvgfParams)]),
    "neb_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "neb_get_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_uint32)]),
    "neb_upload_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "neb_download_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "neb_stream_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "neb_svgf_reset_history": (C.c_int, [C.c_void_p, C.c_void_p]),
    "neb_svgf_level_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "neb_svgf_temporal": (C.c_int, [C.c_void_p, C.c_void_p]),
    "neb_svgf_atrous": (C.c_int, [C.c_void_p, C.c_void_p]),
    "neb_svgf_denoise": (C.c_int, [C.c_void_p, C.c_void_p]),
    "neb_svgf_temporal_rows": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "neb_svgf_atrous_level_rows": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "neb_svgf_atrous_level_planes": (C.c_int, [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_int)] * 4),
    "neb_strips_unique_id": (C.c_int, [C.c_void_p]),
    "neb_strips_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "neb_strips_comm_destroy": (C.c_int, [C.c_void_p]),
    "neb_strips_group_begin": (C.c_int, []),
    "neb_strips_group_end": (C.c_int, []),
    "neb_strips_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(HaloPlane), C.c_uint32, C.POINTER(HaloSwap), C.c_uint32, C.c_void_p]),
    "neb_strips_last_error": (C.c_char_p, []),
}



class StripPlan(C.Structure):
    """neb_strip_plan"""
    _fields_ = [("n_strips", C.c_uint32), ("strip", C.c_uint32), ("scheme", C.c_uint32), ("flags", C.c_uint32)]


class StripPeers(C.Structure):
    """neb_strip_peers"""
    _fields_ = [("up", C.c_void_p), ("down", C.c_void_p)]


STRIP_SCHEMES = {"once": 0, "per_level": 1, "overlap": 2}
STRIP_RESET_HISTORY = 1


def _gi_sigs():
    from . import scene as S
    return {
        "neb_gi_set_scene": (C.c_int, [C.c_void_p, C.POINTER(S.GeometryDesc), C.c_uint32, C.POINTER(S.MaterialDesc), C.c_uint32,
                                       C.POINTER(S.TextureDesc), C.c_uint32]),
        "neb_gi_build_bvh": (C.c_int, [C.c_void_p, C.c_void_p]),
        "neb_gi_scene_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "neb_gi_bvh_depth": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
        "neb_gi_build_passes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
        "neb_gi_build_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
        "neb_gi_scene_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
        "neb_gi_trace": (C.c_int, [C.c_void_p, C.POINTER(S.GIConstants), C.c_void_p]),
        "neb_gi_trace_rows": (C.c_int, [C.c_void_p, C.POINTER(S.GIConstants), C.c_uint32, C.c_uint32, C.c_void_p]),
        "neb_gi_resolve": (C.c_int, [C.c_void_p, C.c_void_p]),
        "neb_gi_trace_begin": (C.c_int, [C.c_void_p, C.POINTER(S.GIConstants), C.c_uint32, C.c_uint32, C.c_void_p]),
        "neb_gi_trace_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
        "neb_gi_ray_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]),
        "neb_gi_traversal_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
        "neb_gi_wave_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
        "neb_gi_node_index_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
        "neb_gi_debug_set_tile_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
        "neb_gi_debug_sun_walk_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
        "neb_gi_sun_table_build_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
        "neb_gi_shadow_tail_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
        "neb_gi_sun_table_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
        "neb_gi_download_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
        "neb_debug_sort_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
        "neb_pbr_direct": (C.c_int, [C.c_void_p, C.POINTER(S.GIConstants), C.c_void_p]),
        "neb_tonemap": (C.c_int, [C.c_void_p, C.c_void_p]),
        "neb_gbuffer_raycast": (C.c_int, [C.c_void_p, C.POINTER(S.CameraDesc), C.c_void_p]),
        "neb_strip_frame": (C.c_int, [C.c_void_p, C.POINTER(S.GIConstants), C.c_void_p, C.POINTER(StripPlan), C.c_void_p]),
        "neb_strip_frame_begin": (C.c_int, [C.c_void_p, C.POINTER(S.GIConstants), C.POINTER(StripPlan), C.POINTER(StripPeers), C.c_void_p]),
        "neb_strip_frame_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(StripPlan), C.POINTER(StripPeers), C.c_void_p]),
        "neb_strip_rows": (C.c_int, [C.c_void_p, C.POINTER(StripPlan), C.POINTER(C.c_uint32)]),
    }


_LIB = None


def exported_symbols():
    return sorted(set(_SIGS) | set(_gi_sigs()))


def load(build_if_missing=True):
    """Load (building first if needed) libnebulae_hip.so.  Raises if it cannot be had."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("NEB_LIB_PATH")  # tuning only: A/B another build of the same sources (still the HIP library)
    if not path:
        path = _build.LIB_PATH
        if build_if_missing and _build.needs_build():
            _build.build()
    if not os.path.exists(path):
        raise NebError(f"{path} is missing and could not be built; nebulae_amd has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in {**_SIGS, **_gi_sigs()}.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(lib, ctx, rc, what):
    if rc != NEB_OK:
        msg = lib.neb_last_error(ctx)
        raise NebError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
