#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

One "step" = one frame of the hot path on device-resident inputs: refresh radiance[cur] with the
frame's noisy input (device-to-device, stands in for the PBR + GI writes that precede SVGF in
Renderer::RenderSceneDeferred, src/Renderer.cpp:123-133), SVGF temporal accumulation, and the
a-trous wavelet levels.  Workload at N=1: BASELINE.json configs[2] -- 1920x1080, 1 spp, 5 a-trous
levels ("sponza-gltf-pbr"; the Sponza geometry blobs are stripped from the reference checkout, so
the G-buffer is the synthetic stand-in of nebulae_amd/synth.py).

Prints ONE JSON line (rank 0).  Extra keys: "roofline" (dominant kernel: a-trous level, HBM bound,
algorithmic 46 B/px/level) and "cpu_baseline" (the scalar C oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TEMPORAL_BYTES_PX = 82  # 60 B read + 22 B written
ATROUS_BYTES_PX = 46    # per level: 30 B read + 16 B written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--atrous-variant", type=int, default=1)
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU oracle to time (0 = skip)")
    return ap.parse_args()


def cpu_baseline(W, H, L, frames, g, rads):
    """Times oracle/svgf_ref.c (kind "port") on all host cores over `frames` full frames."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import OracleSVGF
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)  # the 1-GPU box shares its host: 16 cores is this job's CPU share
    o = OracleSVGF(W, H, L, threads=cores)
    # one untimed frame so history is populated (frame 1 is all-history, quirk 2)
    times = []
    for f in range(1, frames + 2):
        o.begin_frame(f)
        c = o.cur
        o.depth[c][...] = g["depth"]
        o.normal[c][...] = g["normal"]
        t0 = time.perf_counter()
        o.radiance[c][...] = rads[f % len(rads)]
        o.temporal_pass()
        o.atrous_pass()
        t1 = time.perf_counter()
        if f > 1:
            times.append(t1 - t0)
    o.close()
    dt = sum(times) / len(times)
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full {W}x{H} frames (temporal + {L} a-trous levels) of oracle/svgf_ref.c, "
                      f"OpenMP row-parallel on {cores} threads, mean frame {dt * 1e3:.1f} ms"}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nebulae_amd import synth
    from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE, SLOT_CURRENT, SVGFDenoiser

    W, H, L = args.width, args.height, args.levels
    den = SVGFDenoiser()
    den.init(W, H, atrous_levels=L, device=local_rank)
    den.set_option("atrous_variant", args.atrous_variant)

    g = synth.synth_gbuffer(W, H)
    n_inputs = 4
    rads_host = [synth.synth_radiance(g["base"], f + 1) for f in range(n_inputs)]
    rads_dev = [torch.from_numpy(r).cuda() for r in rads_host]
    for slot in (0, 1):  # static camera: both G-buffer slots hold the same depth/normals
        den.upload(PLANE_DEPTH, slot, g["depth"])
        den.upload(PLANE_NORMAL, slot, g["normal"])
    rad_view = [den.plane_tensor(PLANE_RADIANCE, 0), den.plane_tensor(PLANE_RADIANCE, 1)]

    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    frame = [0]

    def step():
        frame[0] += 1
        f = frame[0]
        den.begin_frame(f)
        rad_view[den.get_current_resource_index()].copy_(rads_dev[f % n_inputs], non_blocking=True)
        den.submit_temporal_accumulation(stream=sh)
        den.submit_atrous_compute_wavelet(stream=sh)
        den.end_frame()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel: per-launch duration of the a-trous level kernel (HIP events on its stream) ----
    ev = []
    for _ in range(8):
        frame[0] += 1
        den.begin_frame(frame[0])
        rad_view[den.get_current_resource_index()].copy_(rads_dev[frame[0] % n_inputs], non_blocking=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        den.submit_temporal_accumulation(stream=sh)
        e1.record(stream)
        lv = []
        for i in range(L):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            den.submit_atrous_level(i, (0, H), stream=sh)
            b.record(stream)
            lv.append((a, b))
        ev.append(((e0, e1), lv))
    torch.cuda.synchronize()
    t_temporal = float(np.mean([a.elapsed_time(b) for (a, b), _ in ev])) * 1e-3
    per_level = [float(np.mean([lv[i][0].elapsed_time(lv[i][1]) for _, lv in ev])) * 1e-3 for i in range(L)]
    t_atrous = float(np.mean(per_level))

    if rank == 0:
        px = W * H
        fps = args.steps / dt * world  # every rank denoises its own W x H frame (replicas until strips land)
        achieved = ATROUS_BYTES_PX * px / t_atrous / 1e9
        out = {
            "metric": "denoised frames/s (SVGF temporal + a-trous, 1920x1080 1 spp)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"sponza-standin(synthetic G-buffer) {W}x{H}, 1 spp, SVGF temporal + {L} a-trous levels",
                       "width": W, "height": H, "atrous_levels": L, "atrous_variant": args.atrous_variant,
                       "frames_per_rank": args.steps, "parallelism": f"replicas x{world}" if world > 1 else "single"},
            "frame_algorithmic_GBps": (TEMPORAL_BYTES_PX + ATROUS_BYTES_PX * L) * px * (args.steps / dt) / 1e9,
            "kernel_us": {"temporal": t_temporal * 1e6, "atrous_levels": [t * 1e6 for t in per_level]},
            "roofline": {"bound": "hbm", "kernel": "svgf_atrous_lds_kernel (mean over the levels of a frame)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None},
            "temporal_roofline": {"achieved": TEMPORAL_BYTES_PX * px / t_temporal / 1e9, "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s"},
        }
        if args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(W, H, L, args.cpu_frames, g, rads_host)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    den.destroy()


if __name__ == "__main__":
    main()
