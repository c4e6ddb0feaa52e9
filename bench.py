#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

One "step" = one frame of the hot path on device-resident inputs, in the order of
Renderer::RenderSceneDeferred (/root/reference/src/Renderer.cpp:123-133): radiance[cur] <- the
direct-light term (device-to-device copy of the plane neb_pbr_direct produced once for the static
view; stands in for the per-frame PBR pass that overwrites it), the GI dispatch (one-bounce indirect diffuse, adds into radiance[cur]), SVGF temporal
accumulation and the a-trous wavelet levels.  Workload at N=1: BASELINE.json configs[2] --
1920x1080, 1 spp, 5 a-trous levels on "sponza-standin" (the Sponza geometry blobs are stripped from
the reference checkout; nebulae_amd/scene.py:atrium_standin matches Sponza.gltf's statistics).
The G-buffer is produced once, outside the timed region, by neb_gbuffer_raycast (static camera).

N > 1 (weak scaling): the frame grows with N -- rank r owns a 1080p-equivalent row strip of a
(1920*a) x (1080*b) image, a*b = N (N=4 is BASELINE.json configs[3], 3840x2160) -- and the strips
exchange a-trous halo rows over RCCL (nebulae_amd/strips.py).  `value` is in 1080p-frame
equivalents per second: N x (global frames/s).

Prints ONE JSON line (rank 0) with "roofline" (dominant kernel = a-trous level, HBM bound,
algorithmic 46 B/px/level) and "cpu_baseline" (the scalar C/C++ oracle on the host cores).
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
    ap.add_argument("--width", type=int, default=1920, help="per-GPU-equivalent frame width")
    ap.add_argument("--height", type=int, default=1080, help="per-GPU-equivalent frame height")
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--atrous-variant", type=int, default=1)
    ap.add_argument("--triangles", type=int, default=262267)
    ap.add_argument("--cpu-frames", type=int, default=3, help="SVGF frames of the CPU oracle to time (0 = skip the CPU leg)")
    ap.add_argument("--svgf-only", action="store_true", help="skip the GI dispatch (synthetic noisy radiance instead)")
    ap.add_argument("--gather", action="store_true", help="N > 1: also time the loop with the final gather of all strips to rank 0 after every frame")
    ap.add_argument("--sort-rays", type=int, default=-1, help="GI ray sorting mask: bit 0 shadow rays, bit 1 bounce rays (-1 = library default)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the GI stages of frame f+1 on a side stream while frame f is denoised (measured: +1 %%, off by default)")
    return ap.parse_args()


def measured_traffic(width, height, levels):
    """HBM bytes per a-trous launch from the newest committed PMC summary (tools/traffic_from_pmc.py): PMC counters need
    their own rocprofv3 passes, so bench.py reports the value measured on this workload, or None if there is none."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("width") == width and d.get("height") == height:
            per = [v["total"] for k, v in d["kernels"].items() if "svgf_atrous_lds_kernel" in k]
            if len(per) == levels:
                best = (sum(per) / len(per), os.path.basename(f))
    return best


def measured_valu(width, height, levels):
    """f32 VALU occupancy of the a-trous launches from the newest committed SQ counter summary (profiles/*sq_counters.json):
    {"lane_instructions_per_pixel", "issue_us_per_launch"} averaged over the levels, or None.  The kernel is bound by
    vector-instruction issue, not by HBM; the HBM roofline above is the one the contract asks for."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        per = [v for k, v in d.get("kernels", {}).items() if "svgf_atrous_lds_kernel" in k]
        if len(per) == levels:
            insts = sum(v["SQ_INSTS_VALU"] for v in per) / len(per)
            quad = sum(v["SQ_ACTIVE_INST_VALU"] for v in per) / len(per)
            best = {"lane_instructions_per_pixel": insts * 64.0 / (width * height),
                    "issue_us_per_launch": quad / 1024.0 * 4.0 / 2.4e3,  # quad-cycles per SIMD at the 2.4 GHz peak clock
                    "source": os.path.basename(f)}
    return best


def host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))  # a 1-GPU box shares its host: 16 cores is this job's CPU share


def cpu_baseline(W, H, L, frames, gb, consts, scene, noisy, do_gi):
    """Times the oracle (kind "port": oracle/svgf_ref.c + oracle/trace_ref.cpp) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle_lib import OracleSVGF, OracleTracer
    cores = host_cores()
    t_gi, rays = 0.0, 0
    if do_gi:
        tr = OracleTracer(scene, threads=cores)
        t0 = time.perf_counter()
        noisy, _, rays = tr.gi(gb, consts, radiance=np.zeros((H, W, 4), np.float32), want_hits=False)
        t_gi = time.perf_counter() - t0
        tr.close()
    o = OracleSVGF(W, H, L, threads=cores)
    times = []
    for f in range(1, frames + 2):  # frame 1 (all-history, quirk 2) is warm-up
        o.begin_frame(f)
        c = o.cur
        o.depth[c][...] = gb["depth"]
        o.normal[c][...] = gb["normal"]
        o.radiance[c][...] = noisy
        t0 = time.perf_counter()
        o.temporal_pass()
        o.atrous_pass()
        if f > 1:
            times.append(time.perf_counter() - t0)
    o.close()
    t_svgf = sum(times) / len(times)
    dt = t_gi + t_svgf
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "gi_mrays_per_s": (rays / t_gi / 1e6) if do_gi else None,
            "sample": (f"1 full {W}x{H} GI frame ({rays} rays, {t_gi * 1e3:.0f} ms) of oracle/trace_ref.cpp + " if do_gi else "")
                      + f"{len(times)} full {W}x{H} SVGF frames (temporal + {L} a-trous levels, mean {t_svgf * 1e3:.0f} ms) of "
                        f"oracle/svgf_ref.c; OpenMP row-parallel on {cores} threads"}


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
    if os.environ.get("NEB_BENCH_SHARE_DEVICE"):  # rehearsal of the N > 1 path on a 1-GPU box: every rank on cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        backend = os.environ.get("NEB_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))

    from nebulae_amd import scene as S
    from nebulae_amd import strips, synth
    from nebulae_amd.renderer import RenderInfo
    from nebulae_amd.svgf import PLANE_ALBEDO, PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE, PLANE_ROUGH_METAL, PLANE_WORLDPOS

    L = args.levels
    # weak scaling: the global frame is (W*a) x (H*b) with a*b = world; every rank owns W*H pixels of it
    a, b = strips.frame_factors(world)
    GW, GH = args.width * a, args.height * b
    part = strips.StripPartition(GW, GH, world, L)
    r = strips.StripRenderer(part, rank, device=local_rank, group=dist.group.WORLD if world > 1 else None)
    r.svgf.set_option("atrous_variant", args.atrous_variant)
    r.gi_ui.gi_samples_per_pixel = args.spp
    own0, own1 = part.owned(rank)
    res0, res1 = part.resident(rank)

    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    do_gi = not args.svgf_only
    sc = S.atrium_standin(target_triangles=args.triangles)
    cam = S.sponza_camera()
    r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=sh))
    r.submit_commands_gbuffer()          # G-buffer of this rank's resident rows, slot "current" of frame 1
    torch.cuda.synchronize()
    for pl in (PLANE_NORMAL, PLANE_DEPTH):  # static camera: the other slot holds the same G-buffer
        r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
    rad_view = [r.svgf.plane_tensor(PLANE_RADIANCE, 0), r.svgf.plane_tensor(PLANE_RADIANCE, 1)]
    if args.sort_rays >= 0:
        r.svgf.set_option("gi_sort_rays", args.sort_rays)
    r.submit_commands_pbr_lighting()     # direct sun term of the static view (row f1), computed once, outside the timed region
    torch.cuda.synchronize()
    direct = rad_view[r.svgf.get_current_resource_index()].clone()
    if do_gi:
        r.ray_count(reset=True)
    noisy_dev = None
    if not do_gi:
        g = synth.synth_gbuffer(GW, GH)
        noisy_dev = [torch.from_numpy(synth.synth_radiance(g["base"][res0:res1], f + 1)).cuda() for f in range(4)]
    frame = [1]
    ran_svgf = []
    # Optional frames in flight (the reference keeps 3, src/nri/Swapchain.h:15): the GI stages of frame f+1 touch only
    # the G-buffer and the GI records, so they can run on a side stream while frame f is resolved and denoised on the
    # main stream; the two meet at neb_gi_resolve (the reference's separate nrc Resolve step, DeferredRenderer.cpp:586).
    # Measured on MI355X: +1 % (the GI kernels already occupy every wave slot), so it is off by default.
    overlap = do_gi and args.overlap
    side = torch.cuda.Stream() if overlap else None
    resolved = [None]
    if overlap:
        r.set_defer_resolve(True)

    def step(timed_events=None):
        frame[0] += 1
        f = frame[0]
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=f, stream=sh))
        cur = r.svgf.get_current_resource_index()
        pipelined = overlap and timed_events is None
        gi_stream = side if pipelined else stream
        if not overlap:  # PBR pass stand-in (overwrites radiance[cur]); the fused GI dispatch then adds into it
            rad_view[cur].copy_(direct if do_gi else noisy_dev[f % 4], non_blocking=True)
        if timed_events is not None:
            timed_events["gi0"].record(stream)
        if do_gi:
            if pipelined and resolved[0] is not None:
                side.wait_event(resolved[0])  # the previous frame's resolve has consumed the GI records
            r.submit_commands_gi_pathtrace(stream=gi_stream.cuda_stream)
        if timed_events is not None:
            timed_events["gi1"].record(stream)
        if overlap:
            rad_view[cur].copy_(direct, non_blocking=True)
        if overlap:
            if pipelined:
                done = torch.cuda.Event()
                done.record(side)
                stream.wait_event(done)
            r.submit_commands_gi_resolve()
            if pipelined:
                resolved[0] = torch.cuda.Event()
                resolved[0].record(stream)
        ran_svgf.append(r.submit_commands_svgf_denoising(timed_events))
        r.end_frame()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):  # >= 2: frame 2 is "camera moved", frame 3 resets history
        step()
    barrier()
    del ran_svgf[:]
    if do_gi:
        r.ray_count(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    rays_timed = r.ray_count(reset=True) if do_gi else 0
    assert all(ran_svgf), "SVGF was skipped inside the timed region"
    stats = torch.tensor([dt, float(rays_timed)], dtype=torch.float64,
                         device="cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dt, rays_total = float(mx[0].item()), float(stats[1].item())
    else:
        rays_total = float(rays_timed)

    # ---- optional: the same loop with SURVEY.md 8e's final gather of the strips to rank 0 after every frame ----
    fps_with_gather = None
    if args.gather and world > 1:
        barrier()
        tg = time.perf_counter()
        for _ in range(args.steps):
            step()
            r.gather_frame(dst=0)
        barrier()
        dtg = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=stats.device)
        dist.all_reduce(dtg, op=dist.ReduceOp.MAX)
        fps_with_gather = args.steps / float(dtg[0].item()) * world

    # ---- per-kernel durations: HIP events on the launch stream, 8 extra frames ----
    ev = []
    for _ in range(8):
        e = {k: torch.cuda.Event(enable_timing=True) for k in ("gi0", "gi1", "t0", "t1")}
        e["levels"] = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(L)]
        step(e)
        ev.append(e)
    torch.cuda.synchronize()
    rays_ev = r.ray_count(reset=True) if do_gi else 0
    t_gi = float(np.mean([e["gi0"].elapsed_time(e["gi1"]) for e in ev])) * 1e-3
    t_temporal = float(np.mean([e["t0"].elapsed_time(e["t1"]) for e in ev])) * 1e-3
    per_level = [float(np.mean([e["levels"][i][0].elapsed_time(e["levels"][i][1]) for e in ev])) * 1e-3 for i in range(L)]
    # (with N > 1 the first level's interval also holds the halo exchange it overlaps with, and every level but the last
    # filters a few extra rows: the roofline line is then taken over the levels after the first)
    t_atrous = float(np.mean(per_level if (world == 1 or L == 1) else per_level[1:]))

    if rank == 0:
        own_px = (own1 - own0) * GW                       # pixels a rank owns (= one 1080p frame)
        fps_equiv = args.steps / dt * world               # 1080p-frame equivalents per second, whole job
        achieved = ATROUS_BYTES_PX * own_px / t_atrous / 1e9
        traffic = measured_traffic(GW, GH, L) if world == 1 else None
        valu = measured_valu(GW, GH, L) if world == 1 else None
        if valu:
            valu["busy_frac"] = valu["issue_us_per_launch"] / (sum(per_level) / len(per_level) * 1e6)
        out = {
            "metric": "denoised frames/s (1920x1080-frame equivalents: GI 1 spp + SVGF temporal + a-trous)" if do_gi else
                      "denoised frames/s (1920x1080-frame equivalents: SVGF temporal + a-trous only)",
            "value": fps_equiv, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"sponza-standin {GW}x{GH} ({sc.num_triangles} triangles, {len(sc.geometries)} submeshes), "
                                   f"{args.spp} spp one-bounce GI + SVGF temporal + {L} a-trous levels"
                                   + ("" if do_gi else " [GI skipped: --svgf-only]"),
                       "global_width": GW, "global_height": GH, "atrous_levels": L, "spp": args.spp,
                       "parallelism": (f"row-strips x{world} + RCCL halo exchange ({'one per frame' if part.scheme == 'once' else 'one per a-trous level'}, "
                                       f"{part.exchanged_bytes_per_frame() / 1e6:.1f} MB sent per rank and frame)") if world > 1 else "single GPU",
                       "frames_in_flight": 2 if overlap else 1},
            "frames_per_s_with_final_gather": fps_with_gather,
            "mrays_per_s": (rays_total / dt / 1e6) if do_gi else None,
            "gi_kernel_mrays_per_s": (rays_ev / 8 / t_gi / 1e6) if do_gi else None,
            "frame_algorithmic_GBps": (TEMPORAL_BYTES_PX + ATROUS_BYTES_PX * L) * own_px * world * (args.steps / dt) / 1e9,
            "kernel_us": {"gi_trace": t_gi * 1e6, "temporal": t_temporal * 1e6, "atrous_levels": [t * 1e6 for t in per_level]},
            "roofline": {"bound": "hbm", "kernel": "svgf_atrous_lds_kernel (mean over the levels of a frame)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": ATROUS_BYTES_PX * own_px,
                         "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None,
                         "valu": valu},
            "temporal_roofline": {"achieved": TEMPORAL_BYTES_PX * own_px / t_temporal / 1e9, "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s"},
        }
        if args.cpu_frames > 0 and world == 1:
            gb = {"albedo": r.svgf.download(PLANE_ALBEDO, 0), "rough_metal": r.svgf.download(PLANE_ROUGH_METAL, 0),
                  "world_pos": r.svgf.download(PLANE_WORLDPOS, 0), "normal": r.svgf.download(PLANE_NORMAL, 0),
                  "depth": r.svgf.download(PLANE_DEPTH, 0)}
            noisy = noisy_dev[0].cpu().numpy() if not do_gi else None
            out["cpu_baseline"] = cpu_baseline(GW, GH, L, args.cpu_frames, gb, r.global_constants(), sc, noisy, do_gi)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.destroy()


if __name__ == "__main__":
    main()
