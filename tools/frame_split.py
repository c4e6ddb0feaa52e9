"""Timing + identity: the GI dispatch in two calls with two frames in flight.  neb_gi_trace_begin (ray generation + closest-hit walk) of frame f + 1
runs on a side stream from the moment frame f's shade pass has finished -- beside frame f's short, latency-bound shadow pass, its SVGF chain and the
next direct-term copy -- and neb_gi_trace_finish (shade + shadow passes) of frame f + 1 follows on the main stream.  Prints the wall time per frame of
the serial loop and of the split loop, and checks that both leave the same frame."""
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


def split(r, direct, main, side, f, state, early=False):
    with torch.cuda.stream(main):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
        if early:  # as soon as the record set is free: beside the previous frame's shade pass too
            if state.get(("done", f - 2)) is not None:
                side.wait_event(state.pop(("done", f - 2)))
        elif state.get("shaded") is not None:
            side.wait_event(state["shaded"])  # the previous frame's shade pass has finished (its set's last reader was two frames back)
        r.submit_commands_gi_pathtrace_begin(stream=side.cuda_stream)
        walked = torch.cuda.Event()
        walked.record(side)
        r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
        main.wait_event(walked)
        ev = torch.cuda.Event()
        ev.record(main)  # (creates the underlying hipEvent_t; the library records it again between its two passes)
        r.submit_commands_gi_pathtrace_finish(after_shade_event=ev.cuda_event)
        state["shaded"] = ev
        done = torch.cuda.Event()
        done.record(main)
        state[("done", f)] = done
        r.submit_commands_svgf_denoising()
        r.end_frame()


main, side = torch.cuda.Stream(), torch.cuda.Stream()
outs = {}
for mode in ("serial", "split", "early", "serial", "split", "early"):
    r, direct = make(main)
    state = {}
    step = (lambda f: serial(r, direct, main, f)) if mode == "serial" else (lambda f: split(r, direct, main, side, f, state, early=(mode == "early")))
    for f in range(2, 70):
        step(f)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for f in range(70, 70 + n):
        step(f)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    outs[mode] = r.svgf.download(PLANE_RADIANCE)
    print(f"{W}x{H} {mode}: {(t1 - t0) / n * 1e6:.1f} us per frame = {n / (t1 - t0):.1f} frames/s", flush=True)
    r.destroy()
print("same frame:", bool(np.array_equal(outs["serial"], outs["split"])), bool(np.array_equal(outs["serial"], outs["early"])), float(np.abs(outs["serial"][..., :3]).max()))
