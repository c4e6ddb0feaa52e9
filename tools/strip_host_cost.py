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
for N in ((8, 8, 4, 2) if not os.environ.get("NEB_HOST_COST_C_ONLY") else ()):  # (the first configuration also pays the process's one-time costs: listed twice)
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

# ---- round 5: the same strip frame as ONE library call per phase (neb_strip_frame_begin / _finish), all N strips of the frame on this one GPU
# with the local transport (rows pushed between the contexts): host time per STRIP frame, the library calls alone and with the Python around them ----
for N in (8, 4, 2):
    part = strips.StripPartition(1920, 1080, N, 5, scheme="once")
    rs = [strips.StripRenderer(part, k) for k in range(N)]
    directs = []
    for r in rs:
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1))
        r.submit_commands_gbuffer()
        torch.cuda.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        r.submit_commands_pbr_lighting()
        torch.cuda.synchronize()
        directs.append(r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).clone())
    nb = [(rs[k - 1] if k > 0 else None, rs[k + 1] if k + 1 < N else None) for k in range(N)]
    t_calls = 0.0

    def frame(f, timed=False):
        global t_calls
        info = RenderInfo(scene=sc, camera=cam, frame_index=f)
        for r, d in zip(rs, directs):
            r.begin_frame(info)
            r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(d, non_blocking=True)
        ta = time.perf_counter()
        for phase in ("begin", "finish"):
            for r, (up, down) in zip(rs, nb):
                r.submit_strip_frame_local(phase, up, down)
        t_calls += time.perf_counter() - ta
        for r in rs:
            r.end_frame()
    for f in range(2, 40):
        frame(f)
    torch.cuda.synchronize()
    # (N strips' worth of launches per frame on ONE device: the host is only "submitting" while the launch queue has room -- batches of three
    # frames, drained in between, so that what is timed is the host's own work and not the device's back-pressure)
    n, t_calls, t_host, t_wall, f = 0, 0.0, 0.0, 0.0, 40
    for batch in range(12):
        t0 = time.perf_counter()
        for _ in range(3):
            frame(f)
            f += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t_host += t1 - t0
        t_wall += time.perf_counter() - t0
        n += 3
    print(f"N = {N} ({part.H // N} rows), one library call per strip and phase (local transport, all {N} strips on this GPU): host submits a STRIP frame in "
          f"{t_host / n / N * 1e6:.0f} us, of which the two library calls {t_calls / n / N * 1e6:.0f} us; wall {t_wall / n / N * 1e6:.0f} us per strip frame", flush=True)
    for r in rs:
        r.destroy()
