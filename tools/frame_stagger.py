"""Timing + identity: the DENOISER of frame n beside the memory-bound half of frame n + 1's GI dispatch.  The frame's launches fall into two chains --
GI: ray generation + closest-hit walk W (issue- / latency-bound, fills every wave slot), shade pass S (memory-bound, VALU ~28 %), list pass L (latency-bound);
denoise: direct-term copy C, resolve R, fused temporal + level 0 F (bandwidth-bound), levels A (VALU / LDS-bound) -- coupled only by R(n) after L(n).
"split" (bench.py's default) runs W(n+1) beside L(n) + F(n) + A(n): two issue-bound halves mostly share the chip.  "stagger": with "gi_defer_resolve" the GI chain
runs back to back on a side stream and the WHOLE denoise chain of frame n is held until W(n+1) has finished, i.e. it runs beside S(n+1) + L(n+1).
Prints wall time per frame of serial / split / stagger and checks that all leave the same frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nebulae_amd import scene as S  # noqa: E402
from nebulae_amd.renderer import DeferredRenderer, RenderInfo  # noqa: E402
from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
L = 5
sc, cam = S.atrium_standin(), S.sponza_camera()


def make(main):
    r = DeferredRenderer()
    r.init(W, H, atrous_levels=L)
    with torch.cuda.stream(main):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=main.cuda_stream))
        r.submit_commands_gbuffer()
        main.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        r.submit_commands_pbr_lighting()
        main.synchronize()
        direct = r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).clone()
    return r, direct


def serial(r, direct, main, f):
    with torch.cuda.stream(main):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
        r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
        r.submit_commands_gi_pathtrace()
        r.submit_commands_svgf_denoising()
        r.end_frame()


def split(r, direct, main, side, f, state):
    with torch.cuda.stream(main):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
        if state.get("shaded") is not None:
            side.wait_event(state["shaded"])
        r.submit_commands_gi_pathtrace_begin(stream=side.cuda_stream)
        walked = torch.cuda.Event()
        walked.record(side)
        r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
        main.wait_event(walked)
        ev = torch.cuda.Event()
        ev.record(main)
        r.submit_commands_gi_pathtrace_finish(after_shade_event=ev.cuda_event)
        state["shaded"] = ev
        r.submit_commands_svgf_denoising()
        r.end_frame()


def stagger(r, direct, main, side, f, state, flush=False):
    """frame f's GI chain on `side`; frame f - 1's denoise chain on `main`, released when W(f) has finished"""
    with torch.cuda.stream(main):
        if not flush:
            # ---- W(f): needs nothing of frame f - 1 but its record planes, which the side stream's own order protects ----
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
            r.submit_commands_gi_pathtrace_begin(stream=side.cuda_stream)
            walked = torch.cuda.Event()
            walked.record(side)
        # ---- denoise chain of frame f - 1: C, R (after its GI chain), F + A -- held until W(f) is over ----
        if state.get("gi_done") is not None:
            g = f - 1
            r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=g, stream=main.cuda_stream))  # (the context's cur / hist of frame f - 1 again)
            r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
            main.wait_event(state["gi_done"])
            r.submit_commands_gi_resolve()
            resolved = torch.cuda.Event()
            resolved.record(main)
            state["resolved"] = resolved
            if not flush:
                main.wait_event(walked)
            r.submit_commands_svgf_denoising()
            r.end_frame()
            state["gi_done"] = None
        if flush:
            return
        # ---- S(f) + L(f) on the side stream, once the resolve of frame f - 1 has read the sums out of the record plane ----
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
        if state.get("resolved") is not None:
            side.wait_event(state["resolved"])
        r.submit_commands_gi_pathtrace_finish(stream=side.cuda_stream)
        done = torch.cuda.Event()
        done.record(side)
        state["gi_done"] = done


main, side = torch.cuda.Stream(), torch.cuda.Stream()
outs = {}
for mode in ("serial", "split", "stagger", "serial", "split", "stagger"):
    r, direct = make(main)
    state = {}
    if mode == "stagger":
        r.set_defer_resolve(1)
    step = {"serial": lambda f: serial(r, direct, main, f), "split": lambda f: split(r, direct, main, side, f, state),
            "stagger": lambda f: stagger(r, direct, main, side, f, state)}[mode]
    for f in range(2, 70):
        step(f)
    if mode == "stagger":
        stagger(r, direct, main, side, 70, state, flush=True)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for f in range(70, 70 + n):
        step(f)
    if mode == "stagger":
        stagger(r, direct, main, side, 70 + n, state, flush=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    outs[mode] = r.svgf.download(PLANE_RADIANCE)
    print(f"{W}x{H} {mode}: {(t1 - t0) / n * 1e6:.1f} us per frame = {n / (t1 - t0):.1f} frames/s", flush=True)
    r.destroy()
print("same frame:", bool(np.array_equal(outs["serial"], outs["split"])), bool(np.array_equal(outs["serial"], outs["stagger"])), float(np.abs(outs["serial"][..., :3]).max()))
