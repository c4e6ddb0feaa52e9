"""Diagnostics: is one strip of an N-strip frame bound by the GPU or by the host that submits it?  One interior rank of an N = 8
partition of the 1080p frame runs alone on the device with the halo exchange replaced by nothing (the rows it would receive keep
their old contents: timing only); prints the host time to submit a frame (no synchronisation) and the wall time per frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nebulae_amd import scene as S, strips  # noqa: E402
from nebulae_amd.renderer import RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE  # noqa: E402

sc, cam = S.atrium_standin(), S.sponza_camera()
MODES = [int(m) for m in os.environ.get("NEB_SUN_TABLE_MODES", "1").split(",")]  # "gi_sun_table": 1 table + ray lists, 2 table + tiled / sorted pass, 0 off
for N in (8, 8, 4, 2):  # (the first configuration also pays the process's one-time costs: listed twice)
    for scheme, mode in [(sch, m) for sch in ("once", "per_level") for m in MODES]:
        part = strips.StripPartition(1920, 1080, N, 5, scheme=scheme)
        r = strips.StripRenderer(part, N // 2)
        r._swap_rows_begin = lambda planes, plan: (lambda: None)  # no peers here
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
        r.svgf.set_option("gi_sun_table", mode)
        r.submit_commands_gbuffer()
        torch.cuda.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):  # static camera: both slots hold the G-buffer
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        r.submit_commands_pbr_lighting()
        torch.cuda.synchronize()
        direct = r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).clone()

        def frame(f):
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f))
            r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
            r.submit_commands_gi_pathtrace()
            r.submit_commands_svgf_denoising()
            r.end_frame()
        for f in range(2, 60):
            frame(f)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for f in range(60, 60 + n):
            frame(f)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"N = {N} ({part.H // N} rows), scheme {scheme}, gi_sun_table {mode}: host submits a frame in {(t1 - t0) / n * 1e6:.0f} us; wall {(t2 - t0) / n * 1e6:.0f} us per frame", flush=True)
        r.destroy()
