"""Diagnostics: how much of an N = 8 strip's frame time is idle chip?  Two independent strip renderers (the same interior strip of the 1080p frame, each
with its own context and its own stream) are fed frames alternately on ONE device; the exchange is replaced by nothing (timing only).  If K renderers
in flight deliver K x the frames of one in less than K x the time, a strip's kernels leave that much of the chip idle -- the room that overlapping
frame n's SVGF with frame n + 1's GI would have."""
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


def make(N, stream):
    part = strips.StripPartition(1920, 1080, N, 5, scheme="once")
    r = strips.StripRenderer(part, N // 2)
    r._swap_rows_begin = lambda planes, plan: (lambda: None)  # no peers here
    with torch.cuda.stream(stream):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=stream.cuda_stream))
        r.submit_commands_gbuffer()
        stream.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        r.submit_commands_pbr_lighting()
        stream.synchronize()
        direct = r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).clone()
    return r, direct


def frame(r, direct, stream, f):
    with torch.cuda.stream(stream):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=stream.cuda_stream))
        r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
        r.submit_commands_gi_pathtrace()
        r.submit_commands_svgf_denoising()
        r.end_frame()


def frame_pipelined(r, direct, main, sides, f, state):
    """the GI stages of frame f on a side stream (deferred resolve), beside the SVGF passes of frame f - 1 on the main stream (bench.py --overlap);
    with two side streams (and the library's two record sets, "gi_defer_resolve" = 2) also beside the GI stages of frame f - 1"""
    side = sides[f % len(sides)]
    with torch.cuda.stream(main):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
        ev = state.get(("resolved", f % len(sides)))
        if ev is not None:
            side.wait_event(ev)  # the resolve that consumed this record set
        r.submit_commands_gi_pathtrace(stream=side.cuda_stream)
        r.svgf.plane_tensor(PLANE_RADIANCE, r.svgf.get_current_resource_index()).copy_(direct, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
        main.wait_event(done)
        r.submit_commands_gi_resolve()
        ev = torch.cuda.Event()
        ev.record(main)
        state[("resolved", f % len(sides))] = ev
        r.submit_commands_svgf_denoising()
        r.end_frame()


def frame_split(r, direct, main, side, f, state):
    """neb_gi_trace_begin / _finish (tools/frame_split.py): the closest-hit walk of frame f on the side stream from the end of frame f - 1's shade pass"""
    with torch.cuda.stream(main):
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=main.cuda_stream))
        if state.get("shaded") is not None:
            side.wait_event(state["shaded"])
        r.submit_commands_gi_pathtrace_begin(rows=r.part.gi_rows(r.rank), stream=side.cuda_stream)
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


for N in (8, 4, 2, 1):
    streams = [torch.cuda.Stream() for _ in range(3)]
    rs = [make(N, s) for s in streams]
    for K in ((1, 2, 3) if N > 1 else (1,)):
        for f in range(2, 60):
            for k in range(K):
                frame(rs[k][0], rs[k][1], streams[k], f)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for f in range(60, 60 + n):
            for k in range(K):
                frame(rs[k][0], rs[k][1], streams[k], f)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"N = {N}: {K} strip renderer(s) in flight: host {(t1 - t0) / (n * K) * 1e6:.0f} us per frame; wall {(t2 - t0) / (n * K) * 1e6:.0f} us per frame "
              f"({(t2 - t0) / n * 1e6:.0f} us per round of {K})", flush=True)
    r, direct = rs[0]
    extra = torch.cuda.Stream()
    for depth in (1, 2):
        torch.cuda.synchronize()
        r.set_defer_resolve(depth)
        main, sides, state = streams[0], [streams[1], extra][:depth], {}
        for f in range(200, 260):
            frame_pipelined(r, direct, main, sides, f, state)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for f in range(260, 260 + n):
            frame_pipelined(r, direct, main, sides, f, state)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"N = {N}: ONE renderer, GI of frame f + 1 on a side stream beside the SVGF passes of frame f, {depth} record set(s) / side stream(s): "
              f"host {(t1 - t0) / n * 1e6:.0f} us per frame; wall {(t2 - t0) / n * 1e6:.0f} us per frame", flush=True)
    torch.cuda.synchronize()
    r.set_defer_resolve(0)
    main, side, state = streams[0], streams[1], {}
    for f in range(500, 560):
        frame_split(r, direct, main, side, f, state)
    torch.cuda.synchronize()
    n = 100
    t0 = time.perf_counter()
    for f in range(560, 560 + n):
        frame_split(r, direct, main, side, f, state)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N = {N}: ONE renderer, neb_gi_trace_begin of frame f + 1 on a side stream from the end of frame f's shade pass: wall {(t2 - t0) / n * 1e6:.0f} us per frame", flush=True)
    for r, _ in rs:
        r.destroy()
