"""Host-side mirror of the GI + SVGF portion of the reference's ``Neb::DeferredRenderer``.

Names follow /root/reference/src/DeferredRenderer.h:45-81 (snake_case): ``init``, ``begin_frame``
(with a ``RenderInfo``), ``submit_commands_gi_pathtrace``, ``submit_commands_svgf_denoising``,
``end_frame``.  The frame policy is the reference's: SVGF is skipped while the camera (or sun) moves
and history is reset on the first static frame (src/DeferredRenderer.cpp:133-146,593-614); the first
rendered frame index is 1 (src/Renderer.cpp:262-273).  The raster G-buffer, PBR and tonemap passes
of the reference are outside this path: the G-buffer comes from ``submit_commands_gbuffer`` (a
primary-visibility ray cast with the reference encodings) or is uploaded by the caller.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib, scene as S
from .svgf import PLANE_RADIANCE, SLOT_CURRENT, NebError, SVGFDenoiser  # noqa: F401

HIT_DTYPE = np.dtype([("t", np.float32), ("geometry", np.uint32), ("primitive", np.uint32), ("flags", np.uint32)])


@dataclass
class SceneSunUI:  # src/DeferredRenderer.h:111-116
    rough_diameter: float = 0.58
    direction: tuple = (0.5, -1.0, -0.2)
    radiance: tuple = (20.0, 20.0, 20.0)


@dataclass
class GlobalIlluminationUI:  # src/DeferredRenderer.h:118-125 (nrcMaxPathVertices: 2 = one bounce, the supported value)
    sky_color: tuple = (8.0, 8.0, 8.0)
    gi_samples_per_pixel: int = 1
    max_path_vertices: int = 2
    throughput_threshold: float = 0.01


@dataclass
class RenderInfo:  # src/DeferredRenderer.h:56-63
    scene: object
    camera: S.CameraDesc
    frame_index: int
    stream: int = 0
    timestep: float = 0.0


class DeferredRenderer:
    def __init__(self):
        self.svgf = SVGFDenoiser()
        self.sun = SceneSunUI()
        self.gi_ui = GlobalIlluminationUI()
        self._scene = None
        self._eye = None
        self._sun_key = None
        self.dynamic_scene_this_frame = False
        self.reset_history = False
        # Beyond the reference ("always-on"): keep denoising while the camera moves.  The
        # reference skips SVGF on those frames (DeferredRenderer.cpp:595) and resets the history when the camera stops;
        # with this flag the temporal pass runs every frame (no reprojection: it then exercises quirk 2).
        self.denoise_while_moving = False
        self.info = None

    # ---- DeferredRenderer::Init (src/DeferredRenderer.cpp:26-57) ----
    def init(self, width, height, atrous_levels=None, device=0, row_begin=0, row_end=0):
        self.svgf.init(width, height, atrous_levels=atrous_levels, device=device, row_begin=row_begin, row_end=row_end)
        self.width, self.height = width, height
        return True

    @property
    def _lib(self):
        return self.svgf._lib

    @property
    def _ctx(self):
        return self.svgf._ctx

    def _check(self, rc, what):
        _lib.check(self._lib, self._ctx, rc, what)

    # ---- DeferredRenderer::InitPathtracerScene + InitRTAccelerationStructures (:978-1030,1083-1086) ----
    def init_pathtracer_scene(self, scene, stream=0):
        G, ng, M, nm, T, nt = scene.descs()
        self._scene = None  # (neb_gi_set_scene lets go of the old scene first: if the new one is refused the context has none)
        self._check(self._lib.neb_gi_set_scene(self._ctx, G, ng, M, nm, T, nt), "neb_gi_set_scene")
        self._check(self._lib.neb_gi_build_bvh(self._ctx, C.c_void_p(stream)), "neb_gi_build_bvh")
        self._scene = scene

    def scene_info(self):
        t, n = C.c_uint32(), C.c_uint32()
        self._check(self._lib.neb_gi_scene_info(self._ctx, C.byref(t), C.byref(n)), "neb_gi_scene_info")
        return t.value, n.value

    def scene_bytes(self):
        """device bytes of {texture tables, triangles + shading records, BVH nodes}"""
        v = (C.c_uint64 * 3)()
        self._check(self._lib.neb_gi_scene_bytes(self._ctx, v), "neb_gi_scene_bytes")
        return dict(zip(("texture_tables", "triangles", "bvh_nodes"), [int(x) for x in v]))

    def bvh_depth(self):
        d = C.c_uint32()
        self._check(self._lib.neb_gi_bvh_depth(self._ctx, C.byref(d)), "neb_gi_bvh_depth")
        return d.value

    def build_ms(self):
        """wall time (ms) of the last BVH build"""
        d = C.c_float()
        self._check(self._lib.neb_gi_build_ms(self._ctx, C.byref(d)), "neb_gi_build_ms")
        return float(d.value)

    def build_passes(self):
        d = C.c_uint32()
        self._check(self._lib.neb_gi_build_passes(self._ctx, C.byref(d)), "neb_gi_build_passes")
        return d.value

    # ---- DeferredRenderer::BeginFrame (src/DeferredRenderer.cpp:89-149) ----
    def begin_frame(self, info):
        self.info = info
        if info.scene is not None and info.scene is not self._scene:
            self.init_pathtracer_scene(info.scene, info.stream)
        self.svgf.begin_frame(info.frame_index)
        eye = tuple(info.camera.eye)
        sun_key = (self.sun.rough_diameter, tuple(self.sun.direction), tuple(self.sun.radiance))
        moved = (eye != self._eye) or (self._sun_key is not None and sun_key != self._sun_key)  # :133-146,169-171
        self._sun_key = sun_key
        if moved:
            self._eye = eye
            self.dynamic_scene_this_frame = True
        elif self.dynamic_scene_this_frame:
            self.dynamic_scene_this_frame = False  # camera stopped: reset history and start denoising
            self.reset_history = not self.denoise_while_moving

    def end_frame(self):
        self.svgf.end_frame()

    def global_constants(self):
        """GlobalConstants upload of SubmitCommandsGIPathtrace (src/DeferredRenderer.cpp:403-421)."""
        import math
        c = S.GIConstants()
        c.frameIndex = self.info.frame_index & 0xFFFFFFFF
        c.samplesPerPixel = int(self.gi_ui.gi_samples_per_pixel)
        c.maxPathVertices = int(self.gi_ui.max_path_vertices)
        c.cameraWorldPos[:] = tuple(self.info.camera.eye)
        c.skyColor[:] = self.gi_ui.sky_color
        c.sunLightDirection[:] = self.sun.direction
        c.sunLightRadiance[:] = self.sun.radiance
        c.sunTanHalfAngle = math.tan(math.radians(self.sun.rough_diameter * 0.5))
        c.throughputThreshold = self.gi_ui.throughput_threshold
        return c

    # ---- passes ----
    def submit_commands_gbuffer(self):
        """Stand-in for SubmitCommandsGbuffer (:254-324): primary visibility by ray cast, reference encodings."""
        self._check(self._lib.neb_gbuffer_raycast(self._ctx, C.byref(self.info.camera), C.c_void_p(self.info.stream)),
                    "neb_gbuffer_raycast")

    def submit_commands_pbr_lighting(self):
        """SubmitCommandsPBRLighting (src/DeferredRenderer.cpp:326-394): direct sun light, overwrites radiance[cur]."""
        c = self.global_constants()
        self._check(self._lib.neb_pbr_direct(self._ctx, C.byref(c), C.c_void_p(self.info.stream)), "neb_pbr_direct")

    def submit_commands_hdr_tonemapping(self):
        """SubmitCommandsHDRTonemapping (src/DeferredRenderer.cpp:616-660): radiance[cur] -> LDR plane (RGBA8)."""
        self._check(self._lib.neb_tonemap(self._ctx, C.c_void_p(self.info.stream)), "neb_tonemap")

    def submit_commands_gi_pathtrace(self, rows=None, stream=None):
        c = self.global_constants()
        st = C.c_void_p(self.info.stream if stream is None else stream)
        if rows is None:
            rc = self._lib.neb_gi_trace(self._ctx, C.byref(c), st)
        else:
            rc = self._lib.neb_gi_trace_rows(self._ctx, C.byref(c), rows[0], rows[1], st)
        self._check(rc, "neb_gi_trace")

    def submit_commands_gi_pathtrace_begin(self, rows=None, stream=None):
        """The first half of SubmitCommandsGIPathtrace -- ray generation + the closest-hit walk (neb_gi_trace_begin): touches only the
        G-buffer and GI records, so the next frame's may run beside this frame's shadow pass and SVGF on another stream."""
        c = self.global_constants()
        st = C.c_void_p(self.info.stream if stream is None else stream)
        r0, r1 = (self.svgf.row_begin, self.svgf.row_end) if rows is None else rows
        self._check(self._lib.neb_gi_trace_begin(self._ctx, C.byref(c), r0, r1, st), "neb_gi_trace_begin")

    def submit_commands_gi_pathtrace_finish(self, stream=None, after_shade_event=None):
        """The second half: shading + shadow passes of the dispatch begun longest ago, adding into radiance[cur] (neb_gi_trace_finish).
        after_shade_event: a raw hipEvent_t (e.g. torch.cuda.Event(...).cuda_event) recorded between the two passes."""
        self._check(self._lib.neb_gi_trace_finish(self._ctx, C.c_void_p(self.info.stream if stream is None else stream),
                                                  C.c_void_p(after_shade_event or 0)), "neb_gi_trace_finish")

    def set_defer_resolve(self, on=True):
        """Split the GI dispatch as the reference does (QueryAndTrain ... then Resolve, DeferredRenderer.cpp:560,586)."""
        self.svgf.set_option("gi_defer_resolve", int(on))

    def submit_commands_gi_resolve(self, stream=None):
        self._check(self._lib.neb_gi_resolve(self._ctx, C.c_void_p(self.info.stream if stream is None else stream)), "neb_gi_resolve")

    def submit_commands_svgf_denoising(self):
        if self.dynamic_scene_this_frame and not self.denoise_while_moving:  # :595
            return False
        self._lib.neb_marker_push(b"SVGF Denoising")  # NEB_PIX_SCOPED_EVENT, src/DeferredRenderer.cpp:599
        try:
            if self.reset_history:
                self.reset_history = False
                self.svgf.reset_history(self.info.stream)
            self.svgf.submit_temporal_accumulation(self.info.stream)
            self.svgf.submit_atrous_compute_wavelet(self.info.stream)
        finally:
            self._lib.neb_marker_pop()
        return True

    # ---- introspection ----
    def ray_count(self, reset=False):
        v = C.c_uint64()
        self._check(self._lib.neb_gi_ray_count(self._ctx, C.byref(v), int(reset), C.c_void_p(self.info.stream if self.info else 0)),
                    "neb_gi_ray_count")
        return v.value

    def traversal_stats(self):
        """{rays, node visits and triangle tests of the bounce rays (+ shadow rays while debug hits are on)} as of the
        last ray_count() call."""
        v = (C.c_uint64 * 5)()
        self._check(self._lib.neb_gi_traversal_stats(self._ctx, v), "neb_gi_traversal_stats")
        return dict(zip(("rays", "bounce_nodes", "bounce_tris", "shadow_nodes", "shadow_tris"), [int(x) for x in v]))

    def wave_stats(self):
        """closest-hit pass, as of the last ray_count() with debug hits on: where its waves' loop iterations go"""
        v = (C.c_uint64 * 6)()
        self._check(self._lib.neb_gi_wave_stats(self._ctx, v), "neb_gi_wave_stats")
        return dict(zip(("waves", "iterations", "node_iterations", "node_lanes", "leaf_iterations", "leaf_lanes"), [int(x) for x in v]))

    def node_index_stats(self):
        """closest-hit pass, same collection: node visits by position in the breadth-first node array, and deep stacks"""
        v = (C.c_uint64 * 5)()
        self._check(self._lib.neb_gi_node_index_stats(self._ctx, v), "neb_gi_node_index_stats")
        return dict(zip(("below_64", "below_256", "below_1024", "below_4096", "deep_stack_phases"), [int(x) for x in v]))

    def sun_table_stats(self):
        """{sides proven lit (+normal side / -normal side), shadow rays the table answered as of the last ray_count(), builds}"""
        v = (C.c_uint64 * 4)()
        self._check(self._lib.neb_gi_sun_table_stats(self._ctx, v, C.c_void_p(self.info.stream if self.info else 0)), "neb_gi_sun_table_stats")
        return dict(zip(("lit_plus", "lit_minus", "rays_answered", "builds"), [int(x) for x in v]))

    def sun_table_build_ms(self):
        """device time of the last build of the sun table, or None when none has been built"""
        ms = C.c_float()
        if self._lib.neb_gi_sun_table_build_ms(self._ctx, C.byref(ms)) != 0:
            return None
        return float(ms.value)

    def shadow_tail_mode(self):
        """-> (mode, (us_lists, us_sorted)): 0 the compacted lists take the rays the sun table leaves, 1 the sorted pass, -1 not decided yet for this table"""
        mode, us = C.c_int(), (C.c_float * 2)()
        self._check(self._lib.neb_gi_shadow_tail_mode(self._ctx, C.byref(mode), us), "neb_gi_shadow_tail_mode")
        return int(mode.value), (float(us[0]), float(us[1]))

    def set_debug_hits(self, on=True):
        self.svgf.set_option("gi_debug_hits", int(on))

    def download_hits(self):
        rows = self.svgf.row_end - self.svgf.row_begin
        out = np.zeros((rows, self.width), HIT_DTYPE)
        self._check(self._lib.neb_gi_download_hits(self._ctx, out.ctypes.data_as(C.c_void_p),
                                                   C.c_void_p(self.info.stream if self.info else 0)), "neb_gi_download_hits")
        return out

    def destroy(self):
        self.svgf.destroy()
