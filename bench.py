#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

One "step" = one frame of the hot path on device-resident inputs, in the order of
Renderer::RenderSceneDeferred (/root/reference/src/Renderer.cpp:123-133): radiance[cur] <- the
direct-light term (device-to-device copy of the plane neb_pbr_direct produced once for the static
view; stands in for the per-frame PBR pass that overwrites it), the GI dispatch (one-bounce indirect
diffuse, adds into radiance[cur]), SVGF temporal accumulation and the a-trous wavelet levels.
Workload at N=1: BASELINE.json configs[2] -- 1920x1080, 1 spp, 5 a-trous levels on "sponza-standin"
(the Sponza geometry blobs are stripped from the reference checkout; nebulae_amd/scene.py:atrium_standin
matches Sponza.gltf's statistics: 103 submeshes, 262 k triangles, 25 materials, 69 textures of 1024^2).
The G-buffer is produced once, outside the timed region, by neb_gbuffer_raycast (static camera).

N > 1 (`value`, strong scaling = the metric's own curve): the SAME 1920x1080 frame cut into N row strips, one per GPU,
which exchange a-trous halo rows over RCCL (nebulae_amd/strips.py); `value` = frames/s of that frame.  Two more legs ride in
the same JSON line for N > 1:
  "weak_scaling"  the frame grows with N -- rank r owns a 1080p-equivalent row strip of a (1920*a) x (1080*b) image,
                  a*b = N (N=4 is BASELINE.json configs[3], 3840x2160) -- in 1080p-frame equivalents per second;
  "config5"       (N = 8, or --config5) BASELINE.json configs[4]: 3840x2160 in N strips, 4 spp, the camera
                  orbiting for half the frames and still for the rest -- the reference's policy (SVGF
                  skipped while moving, history reset on the first still frame) and "always-on".

Frames in flight (`--overlap auto`, the default; `off` = every stage of a frame on one stream): the K timed steps are K whole frames between two
barriers + device synchronisations, with the next frame's ray generation + closest-hit walk running on a side stream beside this frame's shadow pass
and SVGF chain (neb_gi_trace_begin / _finish; strips of at most 0.3 M pixels: the whole GI dispatch of the next two frames on two record sets).  Same
frames bit for bit; `config.frames_in_flight` says which, `value_one_frame_in_flight` is the rate of the same K steps run serially.
(NEB_BENCH_PIPELINE=split|defer forces a form: experiments.)

Prints ONE JSON line (rank 0) with "roofline" (dominant SVGF kernel = a-trous level, HBM bound,
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
GI_STREAM_BYTES_PX = 56  # 24 B G-buffer read + 32 B radiance read-modify-write
PROFILE_ROUND = "r05"   # only PMC summaries of this round's kernels are quoted (profiles/r05*_*.json), and only of this very build


def parse(argv=None):
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
    ap.add_argument("--tex-size", type=int, default=1024, help="edge of the stand-in's 68 large textures (Sponza: 1024)")
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames of the CPU oracle to time (0 = skip the CPU leg)")
    ap.add_argument("--svgf-only", action="store_true", help="skip the GI dispatch (synthetic noisy radiance instead)")
    ap.add_argument("--gather", action="store_true", help="N > 1: also time the loop with the final gather of all strips to rank 0 after every frame")
    ap.add_argument("--sort-rays", type=int, default=-1, help="GI ray sorting mask: bit 0 shadow rays, bit 1 bounce rays (-1 = library default)")
    ap.add_argument("--sun-table", type=int, default=int(os.environ.get("NEB_BENCH_SUN_TABLE", "-1")),
                    help="GI sun-visibility table: 1 on, 0 off (every shadow ray traced; same results), -1 = library default (on)")
    ap.add_argument("--overlap", nargs="?", const="on", default="auto", choices=("auto", "on", "off"),
                    help="run the GI stages of frame f+1 on a side stream (deferred resolve) while frame f is denoised.  auto (default): only "
                         "where a rank's strip is at most 0.6 M pixels -- N >= 4 strips of the 1080p frame -- whose short launches leave the chip "
                         "idle (tools/strip_overlap.py: 135-row strip 207 -> 183 us per frame, 270 rows 270 -> 244, 540 rows +-0, whole frame 4 %% slower)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the weak-scaling leg (the frame that grows with N)")
    ap.add_argument("--scene", default=None, help="a .gltf / .glb file to render instead of the procedural Sponza stand-in "
                    "(e.g. a real sponza-gltf-pbr/Sponza.glb, the reference's default scene: src/Nebulae.cpp:36)")
    ap.add_argument("--exchange", choices=("torch", "rccl"), default=None,
                    help="N > 1: halo transport -- torch.distributed P2P on the planes, or the library's own neb_strips_exchange (RCCL); "
                         "default: NEB_STRIPS_EXCHANGE or torch")
    ap.add_argument("--scheme", choices=("once", "per_level", "overlap", "auto"), default="auto",
                    help="N > 1: halo exchange scheme (auto: the cheaper one by strips.choose_scheme's cost table)")
    ap.add_argument("--config5", action="store_true", help="also run BASELINE.json configs[4] (default: only when N = 8)")
    ap.add_argument("--config5-frames", type=int, default=32, help="frames of the config-5 sequence (half moving, half still)")
    ap.add_argument("--config5-size", type=int, nargs=2, default=(3840, 2160), metavar=("W", "H"),
                    help="frame of the config-5 leg (BASELINE.json configs[4]: 3840 2160; the CPU dry run shrinks it)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, child=None, emit=None, timeout=None):
    """`python3 bench.py --gpus N` typed without a launcher: this process -- which has touched no GPU and never will -- starts N
    ranks of `child` (default: this very file) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one per GPU,
    relays rank 0's stdout (the ONE JSON line) and every rank's stderr, and returns 0 only if every rank exited 0.  The first
    rank that fails ends the job: the others are terminated (their own process groups, exact PIDs), nothing is retried and no
    process that initialised a GPU is ever replaced by exec."""
    import signal
    import subprocess
    # Under a profiler this process is not what it seems: rocprofv3's preloaded tool library initialises the GPU before main() runs, and every
    # rank started from here would be an exec out of a fork of a process that holds a GPU -- what this pool forbids (it takes the machine
    # down).  Profile the ranks themselves instead: put rocprofv3 inside each rank's command, the rank program right after `--`.
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY", "HSA_TOOLS_LIB"))
    if child is None and any(t in preload.lower() for t in ("rocprof", "roctracer", "rocprofiler")):
        print("bench.py --gpus N refuses to start ranks from under a profiler (its preloaded library has initialised the GPU in this process): "
              "launch the ranks with torch.distributed.run and profile each rank's own command", file=sys.stderr)
        return 2
    child = child or [sys.executable, os.path.abspath(__file__)]
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(child + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True, start_new_session=True))

    def stop_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        t_end = time.time() + 10
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass

    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    rc, t0 = 0, time.time()
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    print(f"bench.py: rank {r} of {n} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    rc = code if code > 0 else 1
                    live.clear()
                    break
            if timeout is not None and time.time() - t0 > timeout:
                print(f"bench.py: ranks still running after {timeout} s; stopping them", file=sys.stderr, flush=True)
                rc = 124
                break
            time.sleep(0.05)
    finally:
        stop_all()
    reader.join(timeout=5)
    out = [l.rstrip("\n") for l in lines if l.strip()]
    if rc == 0:
        for l in out:
            (emit or (lambda line: print(line, flush=True)))(l)
        if not any(l.startswith("{") for l in out):
            print("bench.py: rank 0 printed no JSON line", file=sys.stderr, flush=True)
            rc = 1
    else:
        for l in out:
            print(l, file=sys.stderr, flush=True)
    return rc


def library_build_id():
    """sha1 over the sources the HIP library is built from: a committed counter summary is only quoted for the build it was
    taken on (tools/profile_round.sh stamps it into the summaries)."""
    import hashlib
    h = hashlib.sha1()
    for d in (os.path.join(ROOT, "nebulae_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):  # (what libnebulae_hip.so is compiled from: not the header-only C++ mirror)
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_profile(pattern):
    """Newest committed profile summary of THIS round AND THIS BUILD matching profiles/<round>*<pattern>, or None: PMC
    counters need their own rocprofv3 passes (one counter group per pass), so bench.py quotes the committed summary of the
    same workload -- never one taken on other kernels."""
    import glob
    best, bid = None, library_build_id()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}*{pattern}"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("build_id") == bid:
            best = (d, os.path.basename(f))
    return best


def _pure_level(name):
    """a kernel-stats name of a pure a-trous level (not the launch that carries the temporal pass: template argument IN = 2)"""
    import re
    m = re.search(r"svgf_atrous_lds_kernel<\d+, \d+, (\d+)", name)
    return "svgf_atrous" in name and (m is None or m.group(1) != "2")


def measured_traffic(width, height, levels):
    """HBM bytes per a-trous launch (FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections, tools/traffic_from_pmc.py),
    mean over the pure levels; the fused temporal + level-0 launch apart."""
    got = committed_profile("hbm_traffic.json")
    if not got:
        return None
    d, name = got
    if d.get("width") != width or d.get("height") != height:
        return None
    per = [v["total"] for k, v in d["kernels"].items() if _pure_level(k)]
    fused = [v["total"] for k, v in d["kernels"].items() if "svgf_atrous" in k and not _pure_level(k)]
    if not per:
        return None
    return sum(per) / len(per), name, (fused[0] if fused else None)


def measured_valu(width, height, levels):
    """f32 VALU instructions of the pure a-trous launches (SQ_INSTS_VALU), or None."""
    got = committed_profile("sq_counters.json")
    if not got:
        return None
    d, name = got
    per = [v for k, v in d.get("kernels", {}).items() if _pure_level(k)]
    if not per:
        return None
    insts = sum(v["SQ_INSTS_VALU"] for v in per) / len(per)
    # priced at the measured issue cost of this kernel's own instruction mix with three waves per SIMD (tools/ubench_bank.hip:
    # 1.25 ns per three-operand wave-instruction, 2.09 per packed pair (round 4: 100 v_pk_* per pixel -- four of a pixel's five tap
    # rows are evaluated for two output rows at once), + 3.4 ns for each v_exp / v_log) -- NOT at SQ_ACTIVE_INST_VALU, which
    # charges 4 cycles per instruction
    transcendental = 2 * 25 * width * height / 64.0
    packed = 100 * width * height / 64.0
    return {"lane_instructions_per_pixel": insts * 64.0 / (width * height),
            "issue_us_per_launch": ((insts - packed) * 1.25e-3 + packed * 2.09e-3 + transcendental * 3.4e-3) / 1024.0,
            "ns_per_wave_instruction": 1.25, "ns_per_packed_pair": 2.09, "ns_extra_per_transcendental": 3.4, "source": name}


def host_cores():
    """-> (threads the CPU leg uses, cores this process may run on, why): every core in the affinity mask, unless a cgroup CPU
    quota says this job may use fewer core-seconds per second (more runnable threads than that are only throttled), or
    NEB_BENCH_CPU_THREADS overrides."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    use, why = n, "every core of the affinity mask"
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            quota = int(txt[0]) if txt[0] != "max" else -1
            period = int(txt[1]) if len(txt) > 1 else int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0 and quota / period < use:
                use, why = max(1, int(quota // period)), f"cgroup CPU quota {quota}/{period} us = {quota / period:.1f} cores"
            break
        except (OSError, ValueError, IndexError):
            continue
    if os.environ.get("NEB_BENCH_CPU_THREADS"):
        use, why = max(1, int(os.environ["NEB_BENCH_CPU_THREADS"])), "NEB_BENCH_CPU_THREADS"
    return use, n, why


def cpu_baseline(W, H, L, frames, gb, consts, scene, noisy, do_gi):
    """Times the oracle (kind "port": oracle/svgf_ref.c + oracle/trace_ref.cpp) on the host cores: all of this job's cores
    (whole frames, median) and ONE thread (a 128-row band of the same frame, scaled to the frame -- a scalar 1080p frame
    takes about a minute)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import numpy as np
    from oracle_lib import OracleSVGF, OracleTracer, SvgfParams, lib
    cores, cores_available, cores_why = host_cores()
    med = lambda v: float(np.median(v))  # noqa: E731
    # ---- all cores: `frames` GI frames + `frames` SVGF frames, median of each ----
    t_gi, rays = [], 0
    if do_gi:
        tr = OracleTracer(scene, threads=cores)
        for _ in range(frames):
            t0 = time.perf_counter()
            noisy, _, rays = tr.gi(gb, consts, radiance=np.zeros((H, W, 4), np.float32), want_hits=False)
            t_gi.append(time.perf_counter() - t0)
    o = OracleSVGF(W, H, L, threads=cores)
    t_svgf = []
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
            t_svgf.append(time.perf_counter() - t0)
    dt_all = (med(t_gi) if do_gi else 0.0) + med(t_svgf)
    # ---- one thread: rows [b0, b1) of the same frame ----
    b0 = max(0, (H // 2 - 64) // 8 * 8)
    b1 = min(H, b0 + 128)
    scale = H / float(b1 - b0)
    t1_gi = 0.0
    if do_gi:
        tr1 = OracleTracer(scene, threads=1)
        t0 = time.perf_counter()
        tr1.gi(gb, consts, radiance=np.zeros((H, W, 4), np.float32), rows=(b0, b1), want_hits=False)
        t1_gi = time.perf_counter() - t0
        tr1.close()
        tr.close()
    Lb = lib()
    Lb.svgf_ref_temporal.argtypes = [C.c_int] * 4 + [C.c_void_p] * 9 + [C.POINTER(SvgfParams)]
    Lb.svgf_ref_atrous.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int, C.POINTER(SvgfParams)]
    prm = SvgfParams(0.002, 0.9, 1e-4, 4.0 / 255.0, 128.0, 0.002)
    c, h = o.cur, o.hist
    src, dst = o.radiance[c].copy(), np.zeros((H, W, 4), np.float32)
    t0 = time.perf_counter()
    Lb.svgf_ref_temporal(W, H, b0, b1, src.ctypes.data, o.radiance[h].ctypes.data, o.depth[c].ctypes.data, o.depth[h].ctypes.data,
                         o.normal[c].ctypes.data, o.normal[h].ctypes.data, o.moments[h].ctypes.data, o.moments[c].ctypes.data,
                         o.variance.ctypes.data, C.byref(prm))
    for lvl in range(L):
        Lb.svgf_ref_atrous(W, H, b0, b1, src.ctypes.data, dst.ctypes.data, o.variance.ctypes.data, o.depth[c].ctypes.data,
                           o.normal[c].ctypes.data, 1 << lvl, C.byref(prm))
    t1_svgf = time.perf_counter() - t0
    o.close()
    dt_one = (t1_gi + t1_svgf) * scale
    return {"value": 1.0 / dt_all, "unit": "frames/s", "cores": cores, "cores_available": cores_available,
            "cores_note": f"{cores} of the {cores_available} cores this process may run on (host has {os.cpu_count()}): {cores_why}", "kind": "port",
            "gi_mrays_per_s": (rays / med(t_gi) / 1e6) if do_gi else None,
            "sample": (f"median of {frames} full {W}x{H} GI frames ({rays} rays, {med(t_gi) * 1e3:.0f} ms) of oracle/trace_ref.cpp + " if do_gi else "")
                      + f"median of {len(t_svgf)} full {W}x{H} SVGF frames (temporal + {L} a-trous levels, {med(t_svgf) * 1e3:.0f} ms) of "
                        f"oracle/svgf_ref.c; OpenMP row-parallel on {cores} threads",
            "single_thread": {"value": 1.0 / dt_one, "unit": "frames/s", "cores": 1,
                              "sample": f"rows [{b0},{b1}) of the same {W}x{H} frame on one thread (GI {t1_gi:.1f} s + SVGF {t1_svgf:.1f} s), "
                                        f"scaled by {scale:.2f} to the whole frame"}}


class GpuRuntime:
    """What the control flow below needs from the device side.  bench.py always runs with this one; tests/test_bench_dryrun.py
    substitutes a CPU stand-in (no-op events, an oracle-backed strip renderer) to rehearse the N = 8 control flow under gloo."""
    name = "hip"

    def __init__(self):
        import torch
        self.torch = torch

    def available(self):
        return self.torch.cuda.is_available()

    def set_device(self, local_rank):
        self.torch.cuda.set_device(local_rank)

    def init_process_group(self, dist, local_rank):
        backend = os.environ.get("NEB_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": self.torch.device("cuda", local_rank)} if backend == "nccl" else {}))

    def current_stream(self):
        return self.torch.cuda.current_stream()

    def stream_handle(self, stream):
        return stream.cuda_stream

    def new_stream(self):
        # (experiment: NEB_BENCH_SIDE_PRIORITY = the HIP priority of the side stream(s) that carry the next frame's closest-hit walk: lower = more urgent)
        prio = os.environ.get("NEB_BENCH_SIDE_PRIORITY")
        return self.torch.cuda.Stream(priority=int(prio)) if prio is not None else self.torch.cuda.Stream()

    def synchronize(self):
        self.torch.cuda.synchronize()

    def event(self, timing=True):
        return self.torch.cuda.Event(enable_timing=timing)

    def event_handle(self, event):
        """the raw hipEvent_t of a torch event that has been recorded at least once"""
        return event.cuda_event

    def to_device(self, t):
        return t.cuda()

    def make_renderer(self, part, rank, local_rank, group, exchange):
        from nebulae_amd import strips
        return strips.StripRenderer(part, rank, device=local_rank, group=group, exchange=exchange)


class Workload:
    """One strip renderer of a GW x GH frame cut into `world` row strips, its static G-buffer and direct-light term."""

    def __init__(self, rt, args, GW, GH, L, spp, sc, cam, rank, world, local_rank, group, do_gi=True, scheme=None, link=None):
        import torch
        from nebulae_amd import strips, synth
        self.rt = rt
        from nebulae_amd.renderer import RenderInfo
        from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE
        self.torch, self.RenderInfo = torch, RenderInfo
        self.args, self.GW, self.GH, self.L, self.sc, self.cam, self.rank, self.world, self.do_gi = args, GW, GH, L, sc, cam, rank, world, do_gi
        self.part = strips.StripPartition(GW, GH, world, L, scheme=scheme, link=link)
        self.r = r = rt.make_renderer(self.part, rank, local_rank, group, args.exchange)
        # one strip = the whole frame: temporal + a-trous run as the library's fused chain when every pixel is covered by both
        self.fused_chain = (world == 1 and GW % 8 == 0 and GH % 8 == 0 and 1 <= L <= 6 and args.atrous_variant == 1
                            and not os.environ.get("NEB_BENCH_NO_FUSE"))
        if os.environ.get("NEB_BENCH_NO_FUSE"):
            r.svgf.set_option("svgf_fuse", 0)
        r.svgf.set_option("atrous_variant", args.atrous_variant)
        r.gi_ui.gi_samples_per_pixel = spp
        self.own = self.part.owned(rank)
        self.res = self.part.resident(rank)
        self.stream = rt.current_stream()
        if os.environ.get("NEB_BENCH_MAIN_PRIORITY") is not None and hasattr(rt, "torch"):  # (experiment: the frame's own stream with a HIP priority)
            self.stream = rt.torch.cuda.Stream(priority=int(os.environ["NEB_BENCH_MAIN_PRIORITY"]))
            rt.torch.cuda.set_stream(self.stream)
        self.sh = rt.stream_handle(self.stream)
        r.begin_frame(RenderInfo(scene=sc, camera=cam, frame_index=1, stream=self.sh))
        r.submit_commands_gbuffer()          # G-buffer of this rank's resident rows, slot "current" of frame 1
        rt.synchronize()
        for pl in (PLANE_NORMAL, PLANE_DEPTH):  # static camera: the other slot holds the same G-buffer
            r.svgf.plane_tensor(pl, 0).copy_(r.svgf.plane_tensor(pl, 1))
        self.rad_view = [r.svgf.plane_tensor(PLANE_RADIANCE, 0), r.svgf.plane_tensor(PLANE_RADIANCE, 1)]
        if args.sort_rays >= 0:
            r.svgf.set_option("gi_sort_rays", args.sort_rays)
        if args.sun_table >= 0:
            r.svgf.set_option("gi_sun_table", args.sun_table)
        if os.environ.get("NEB_BENCH_SUN_HINTS"):  # tuning: occluder hints tried per hit in the shade pass (0, 2, 4)
            r.svgf.set_option("gi_sun_hints", int(os.environ["NEB_BENCH_SUN_HINTS"]))
        r.submit_commands_pbr_lighting()     # direct sun term of the static view (row f1), computed once, outside the timed region
        rt.synchronize()
        self.direct = self.rad_view[r.svgf.get_current_resource_index()].clone()
        self.noisy_dev = None
        if do_gi:
            r.ray_count(reset=True)
        else:
            g = synth.synth_gbuffer(GW, GH)
            self.noisy_dev = [rt.to_device(torch.from_numpy(synth.synth_radiance(g["base"][self.res[0]:self.res[1]], f + 1))) for f in range(4)]
        self.frame = 1
        self.ran_svgf = []
        # Frames in flight (the reference keeps 3, src/nri/Swapchain.h:15).  Kernels that are bound by their own dependent chains leave the chip idle;
        # work of the NEXT frame that depends on nothing of this one can run beside them on a side stream.  Two forms, same frames bit for bit
        # (tools/strip_overlap.py, tools/frame_split.py; --overlap off = one frame in flight, reported beside as value_one_frame_in_flight):
        #  "split":  neb_gi_trace_begin (ray generation + closest-hit walk) of frame f+1 starts when frame f's shade pass has finished and runs beside
        #            frame f's short, latency-bound shadow pass, its SVGF chain and the direct-term copy; neb_gi_trace_finish (shade + shadow passes)
        #            follows on the main stream.  Whole 1080p frame 649 -> 614-622 us, 540-row strip 402 -> 373, 270 rows 271 -> 235.
        #  "defer":  the whole GI dispatch of frames f+1 and f+2 on two side streams and two record sets ("gi_defer_resolve" = 2), meeting the SVGF
        #            passes at neb_gi_resolve: best on the smallest strips (135 rows: 208 -> 149 us, split: 166), no gain on big ones.
        own_px = ((GH + world - 1) // world) * GW  # (the largest strip's: every rank must take the same form -- the timed regions hold collectives)
        splittable = do_gi and spp == 1 and int(r.gi_ui.max_path_vertices) <= 2  # (one sample, one bounce per pixel: what neb_gi_trace_begin / _finish take)
        forced = os.environ.get("NEB_BENCH_PIPELINE")  # experiments: "split" | "defer"
        self.mode = None
        if do_gi and args.overlap != "off":
            self.mode = (forced if forced != "split" or splittable else None) or ("defer" if (world > 1 and own_px <= 300_000) or not splittable else "split")
            if self.mode == "defer" and not forced and args.overlap == "auto" and not (world > 1 and own_px <= 300_000):
                self.mode = None  # (several samples per pixel on a big strip: nothing to gain)
            if self.mode == "split" and not forced and args.overlap == "auto" and own_px > 6_000_000:
                self.mode = None  # (measured, tools/size_sweep.sh: 2560x1440 gains 1.3 %, 3840x2160 LOSES 2.5 % -- its launches fill the chip for long enough as they are)
        self.overlap = self.mode is not None
        self.depth = 2 if self.mode == "defer" else 0
        self.sides = [rt.new_stream() for _ in range(2 if self.mode == "defer" else 1 if self.mode == "split" else 0)]
        self.resolved = [None, None]
        self.shaded = None
        self.force_serial = False
        if self.mode == "defer":
            r.set_defer_resolve(2)

    def frames_in_flight(self):
        return {None: 1, "split": 2, "defer": 3}[self.mode]

    def step(self, timed_events=None, cam=None, regen_gbuffer=False):
        torch, r, stream = self.torch, self.r, self.stream
        self.frame += 1
        f = self.frame
        r.begin_frame(self.RenderInfo(scene=self.sc, camera=cam or self.cam, frame_index=f, stream=self.sh))
        cur = r.svgf.get_current_resource_index()
        if regen_gbuffer:  # a camera that moved: the raster pass re-renders the G-buffer and the PBR pass its direct term
            r.submit_commands_gbuffer()
            r.submit_commands_pbr_lighting()
            self.direct.copy_(self.rad_view[cur], non_blocking=True)  # (kept for the still frames that follow)
        pipelined = self.overlap and timed_events is None and not self.force_serial
        defer = self.mode == "defer"
        slot = f % 2
        side = self.sides[slot if defer else 0] if self.overlap else None
        if not defer and not regen_gbuffer and not (self.mode == "split" and pipelined):
            # PBR pass stand-in (overwrites radiance[cur]); the GI dispatch then adds into it
            self.rad_view[cur].copy_(self.direct if self.do_gi else self.noisy_dev[f % 4], non_blocking=True)
        if timed_events is not None:
            timed_events["gi0"].record(stream)
        if self.do_gi and self.mode == "split" and pipelined:
            if self.shaded is not None:
                side.wait_event(self.shaded)  # the previous frame's shade pass is over (what read this record set finished a frame before that)
            if regen_gbuffer:  # the G-buffer of this frame was just rendered on the main stream
                drawn = self.rt.event(timing=False)
                drawn.record(stream)
                side.wait_event(drawn)
            r.submit_commands_gi_pathtrace_begin(rows=self.part.gi_rows(self.rank), stream=self.rt.stream_handle(side))
            walked = self.rt.event(timing=False)
            walked.record(side)
            if not regen_gbuffer:
                self.rad_view[cur].copy_(self.direct, non_blocking=True)
            stream.wait_event(walked)
            self.shaded = self.rt.event(timing=False)
            self.shaded.record(stream)  # (creates the underlying event; the library records it again between its shade and shadow passes)
            r.submit_commands_gi_pathtrace_finish(after_shade_event=self.rt.event_handle(self.shaded))
        elif self.do_gi:
            gi_stream = side if (pipelined and defer) else stream
            if pipelined and defer and self.resolved[slot] is not None:
                side.wait_event(self.resolved[slot])  # the resolve that consumed this record set
            if pipelined and defer and regen_gbuffer:
                drawn = self.rt.event(timing=False)
                drawn.record(stream)
                side.wait_event(drawn)
            r.submit_commands_gi_pathtrace(stream=self.rt.stream_handle(gi_stream))
        if self.mode == "split" and not pipelined and self.do_gi:
            self.shaded = self.rt.event(timing=False)  # (a serial frame in between: the next split frame's walk waits for all of its GI)
            self.shaded.record(stream)
        if timed_events is not None:
            timed_events["gi1"].record(stream)
        if defer:
            self.rad_view[cur].copy_(self.direct, non_blocking=True)
            if pipelined:
                done = self.rt.event(timing=False)
                done.record(side)
                stream.wait_event(done)
            r.submit_commands_gi_resolve()
            self.resolved[slot] = self.rt.event(timing=False)  # (recorded on the serial, event-timed frames too: the next pipelined frame of this
            self.resolved[slot].record(stream)                 # parity must not start tracing into the set before this resolve has read it)
        self.ran_svgf.append(r.submit_commands_svgf_denoising(timed_events))
        r.end_frame()

    def barrier(self):
        import torch.distributed as dist
        self.rt.synchronize()
        if self.world > 1:
            dist.barrier()
        self.rt.synchronize()

    def reduce_max_sum(self, dt, rays):
        """-> (max over ranks of dt, sum over ranks of rays)"""
        import torch.distributed as dist
        torch = self.torch
        if self.world == 1:
            return dt, float(rays)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"  # (the MAX / SUM over ranks travels on the job's own backend)
        mx = torch.tensor([dt], dtype=torch.float64, device=dev)
        sm = torch.tensor([float(rays)], dtype=torch.float64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        return float(mx[0].item()), float(sm[0].item())

    # A freshly created context reaches its steady frame time only after ~50 frames (clocks, TLBs over the 0.8 GB of scene tables):
    # 20 timed steps behind 5 warm-ups read ~1290 frames/s, behind 64 warm-ups ~1352 on the same box.  `value` is what the contract
    # defines -- exactly W warm-up frames, then K timed steps; the settled rate is measured in a SECOND timed region of K steps once
    # SETTLE_FRAMES frames have run, and reported beside it (`value_settled`).
    SETTLE_FRAMES = 48

    def timed(self, steps, warmup, step_fn=None):
        """`warmup` untimed steps, barrier, EXACTLY `steps` steps, barrier -> (seconds: max over ranks, rays: sum over ranks)"""
        step_fn = step_fn or (lambda k: self.step())
        for k in range(warmup):
            step_fn(k - warmup)
        self.barrier()
        del self.ran_svgf[:]
        if self.do_gi:
            self.r.ray_count(reset=True)
        t0 = time.perf_counter()
        for k in range(steps):
            step_fn(k)
        self.barrier()
        dt = time.perf_counter() - t0
        rays = self.r.ray_count(reset=True) if self.do_gi else 0
        # of those, the sun-visibility queries the table answered inside the shade pass (no walk): `rays` counts QUERIES, SURVEY 8d's formula
        answered = self.r.sun_table_stats()["rays_answered"] if self.do_gi else 0
        _, self.last_answered = self.reduce_max_sum(dt, answered)
        return self.reduce_max_sum(dt, rays)

    def parallelism_label(self):
        p = self.part
        if self.world == 1:
            return "single GPU" + self.pipeline_label()
        return (f"row-strips x{self.world} of {p.H // p.N} rows + halo exchange over RCCL: scheme '{p.scheme}' ({p.scheme_reason}; "
                f"{ {'once': 'one exchange per frame', 'per_level': 'one exchange per a-trous level', 'overlap': 'two exchanges per frame, GI recomputed on the overlap rows'}[p.scheme]}, {p.exchanged_bytes_per_frame() / 1e6:.1f} MB sent "
                f"per rank and frame), transport '{self.r.exchange}' ({'neb_strips_exchange: grouped ncclSend / ncclRecv' if self.r.exchange == 'rccl' else 'torch.distributed batch_isend_irecv on the planes'})"
                + self.pipeline_label())

    def pipeline_label(self):
        if self.mode == "split":
            return ("; 2 frames in flight: ray generation + closest-hit walk of frame f+1 (neb_gi_trace_begin) on a side stream from the end of frame f's "
                    "shade pass, beside its shadow pass, SVGF chain and the direct-term copy")
        if self.mode == "defer":
            return ("; 3 frames in flight: the GI stages of frames f+1 and f+2 on two side streams and two record sets beside the SVGF passes of frame f, "
                    "meeting at neb_gi_resolve")
        return ""

    def destroy(self):
        self.r.destroy()


def run_config5(rt, args, sc, rank, world, local_rank, group, scheme=None, link=None):
    """BASELINE.json configs[4]: 3840x2160 in `world` strips, 4 spp, 5 levels; the camera orbits
    (yaw += 0.5 deg per frame) for the first half of the sequence and stands still for the second.  Two runs: the
    reference's policy (SVGF skipped while moving, history reset on the first still frame) and always-on (beyond the
    reference: the temporal pass runs every frame, no reprojection)."""
    from nebulae_amd import scene as S
    n = max(2, args.config5_frames)
    half = n // 2
    c5w, c5h = args.config5_size

    def cam_at(k):
        return S.orbit_camera(origin=(0.0, 2.0, 0.0), yaw_deg=12.0 + 0.5 * min(k + 1, half), pitch_deg=60.0, distance=9.0)
    out = {"workload": f"{c5w}x{c5h}, 4 spp one-bounce GI + SVGF temporal + 5 a-trous levels, {world} row strips of {c5h // world} rows, "
                       f"{half} frames with the camera orbiting (yaw += 0.5 deg per frame; G-buffer and direct term re-rendered) then {n - half} still frames"}
    for mode in ("reference_policy", "always_on"):
        w = Workload(rt, args, c5w, c5h, 5, 4, sc, cam_at(-1), rank, world, local_rank, group, scheme=scheme, link=link)
        w.r.denoise_while_moving = mode == "always_on"
        w.step()  # one still frame first, so that the sequence starts from a settled state
        w.step()

        def fn(k, w=w):
            # the G-buffer slot of the first still frame (k == half) still holds an older view: it is rendered once more
            w.step(cam=cam_at(k) if k >= 0 else None, regen_gbuffer=0 <= k <= half)
        dt, rays = w.timed(n, 0, fn)
        denoised = sum(1 for x in w.ran_svgf if x)
        out[mode] = {"frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3, "mrays_per_s": rays / dt / 1e6, "frames": n,
                     "frames_denoised": denoised}
        w.destroy()
    return out


def main(argv=None, rt=None, emit=None):
    """argv / rt / emit: None for the real run (sys.argv, the HIP runtime, print); the CPU dry run passes its own."""
    args = parse(argv)
    if rt is None and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed as `python3 bench.py --gpus N` with no launcher around it: be the launcher (before anything touches a GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    import numpy as np
    import torch
    import torch.distributed as dist
    rt = rt or GpuRuntime()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not rt.available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("NEB_BENCH_SHARE_DEVICE"):  # rehearsal of the N > 1 path on a 1-GPU box: every rank on cuda:0
        local_rank = 0
    rt.set_device(local_rank)
    if world > 1:
        rt.init_process_group(dist, local_rank)
    group = dist.group.WORLD if world > 1 else None

    from nebulae_amd import scene as S
    from nebulae_amd import strips
    # N > 1: what one halo exchange with a neighbour costs on THIS node -- 20 exchanges of 4 KB and of 2 MB, the product's own
    # batch_isend_irecv pattern, outside every timed region -- feeds the scheme chooser in place of guessed xGMI constants
    link = None
    if world > 1:
        link = strips.measure_link(rank, world, group, lambda n: rt.to_device(torch.zeros(n, dtype=torch.uint8)), rt.synchronize)
    from nebulae_amd.svgf import PLANE_ALBEDO, PLANE_DEPTH, PLANE_NORMAL, PLANE_ROUGH_METAL, PLANE_WORLDPOS

    L = args.levels
    do_gi = not args.svgf_only
    if args.scene:
        # a real scene file (e.g. the reference's default sponza-gltf-pbr/Sponza.glb, stripped from its checkout) through the
        # same loader the tests use; the camera stays the reference's Sponza view
        sc = S.load_gltf(args.scene)
        scene_label = f"{os.path.basename(args.scene)}"
        # the reference's Sponza view for a Sponza file (Nebulae's default scene), otherwise a camera that frames the scene's box
        cam = S.sponza_camera() if "sponza" in scene_label.lower() else S.framing_camera(sc)
    else:
        sc = S.atrium_standin(target_triangles=args.triangles, tex_size=args.tex_size)
        scene_label = "sponza-standin"
        cam = S.sponza_camera()
    scheme = None if args.scheme == "auto" else args.scheme

    def workload_label(gw, gh, spp, levels, gi=True):
        tex = sorted({t.shape[0] for t in sc.textures}) if sc.textures else []
        return (f"{scene_label} {gw}x{gh} ({sc.num_triangles} triangles, {len(sc.geometries)} submeshes, {len(sc.materials)} materials, "
                f"{len(sc.textures)} textures" + (f" of {tex[-1]}^2" if tex else "") + f"), {spp} spp one-bounce GI + SVGF temporal + {levels} a-trous levels"
                + ("" if gi else " [GI skipped: --svgf-only]"))

    # ---- the primary workload: ONE width x height frame (BASELINE.json configs[2]) on `world` GPUs = `world` row strips ----
    GW, GH = args.width, args.height
    w = Workload(rt, args, GW, GH, L, args.spp, sc, cam, rank, world, local_rank, group, do_gi=do_gi, scheme=scheme, link=link)
    frames_in_flight = w.frames_in_flight()
    r, part = w.r, w.part
    scene_bytes = r.scene_bytes() if do_gi else None
    bvh = {"triangles": r.scene_info()[0], "bvh4_nodes": r.scene_info()[1], "bvh4_depth": r.bvh_depth(), "build_ms": round(r.build_ms(), 2)} if do_gi else None

    # (frame 1 -- "camera moved": SVGF skipped -- ran in the set-up; frame 2 resets the history.  At least one warm-up frame runs so
    # that every timed frame is a steady-state denoised frame; `warmup_run` on the line says what ran)
    warmup_run = max(args.warmup, 1)
    dt, rays_total = w.timed(args.steps, warmup_run)
    answered_total = w.last_answered
    assert all(w.ran_svgf), "SVGF was skipped inside the timed region"
    if do_gi:  # (as of the ray count that ended the timed region: sides proven lit, shadow rays the table answered in those K frames)
        bvh["sun_table"] = r.sun_table_stats()
        ms = r.sun_table_build_ms()
        bvh["sun_table"]["build_ms"] = None if ms is None else round(ms, 2)
        mode, (us_lists, us_sorted) = r.shadow_tail_mode()  # which pass takes the rays the table leaves: measured by the library once per table build
        bvh["sun_table"]["tail"] = {"mode": {0: "lists", 1: "sorted pass", -1: "undecided"}[mode], "timed_us": {"lists": round(us_lists, 1), "sorted_pass": round(us_sorted, 1)}}
    # the settled rate: the same K steps, timed the same way, once the context has run SETTLE_FRAMES frames in all
    settle_run = max(Workload.SETTLE_FRAMES - warmup_run - args.steps, 0)
    dt_settled, _ = w.timed(args.steps, settle_run)
    dt_serial = None
    if w.overlap:  # the same K steps with ONE frame in flight (every stage of a frame on the main stream, in order)
        w.force_serial = True
        dt_serial, _ = w.timed(args.steps, 2)
        w.force_serial = False

    # ---- optional: the same loop with SURVEY.md 8e's final gather of the strips to rank 0 after every frame ----
    fps_with_gather = None
    if args.gather and world > 1:
        dtg, _ = w.timed(args.steps, 0, lambda k: (w.step(), r.gather_frame(dst=0)))
        fps_with_gather = args.steps / dtg

    # ---- per-kernel durations: HIP events on the launch stream, 8 extra frames with an event pair per pass, and 8 more in
    # which the SVGF chain is bracketed by ONE pair (every event is a packet of its own between two launches).  One GPU: the
    # chain is the library's own (temporal pass fused into level 0) and the per-kernel marks are its own events on the
    # launch stream (option svgf_profile -> neb_svgf_level_times); N > 1: the strip renderer's per-level events. ----
    def avg(v):
        """average over the profiled frames, without the one largest and the one smallest value (a frame in ~100 runs 150 us long
        on these boxes, whatever it is made of: among 16 samples one such frame moved a plain mean by 10 us)"""
        v = sorted(float(x) for x in v)
        return float(np.mean(v[1:-1] if len(v) > 4 else v))

    NPROF = 16
    ev, ev2, lt = [], [], []
    if world == 1:
        r.svgf.set_option("svgf_profile", 1)
    for _ in range(NPROF):
        e = {k: rt.event() for k in ("gi0", "gi1", "t0", "t1", "a1")}
        e["levels"] = [(rt.event(), rt.event()) for _ in range(L)]
        w.step(e)
        if world == 1:
            lt.append(r.svgf.level_times())
        ev.append(e)
    if world == 1:
        r.svgf.set_option("svgf_profile", 0)
    for _ in range(NPROF):
        e = {k: rt.event() for k in ("gi0", "gi1", "t0", "t1", "a1")}
        w.step(e)
        ev2.append(e)
    # the levels behind the first as ONE interval of the library's events (svgf_profile = 2: no event between them)
    lt2 = []
    if world == 1:
        r.svgf.set_option("svgf_profile", 2)
        for _ in range(NPROF):
            w.step()
            lt2.append(r.svgf.level_times())
        r.svgf.set_option("svgf_profile", 0)
    rt.synchronize()
    rays_ev = (r.ray_count(reset=True) if do_gi else 0) // (3 if world == 1 else 2)  # (batches of NPROF frames)
    answered_ev = (r.sun_table_stats()["rays_answered"] if do_gi else 0) // (3 if world == 1 else 2)
    t_gi = avg([e["gi0"].elapsed_time(e["gi1"]) for e in ev]) * 1e-3
    fused = bracketed = False
    atrous_samples = None
    if world == 1:
        per_level = [avg([x[i] for x in lt]) * 1e-6 for i in range(L)]
        t_temporal = avg([e["t0"].elapsed_time(e["t1"]) for e in ev2]) * 1e-3
        fused = w.fused_chain
        if fused:
            t_temporal = 0.0  # (the held-back call: the pass runs inside level 0's kernel)
        t_svgf_chain = avg([e["t0"].elapsed_time(e["a1"]) for e in ev2]) * 1e-3  # temporal + all levels, two events
        # the roofline's kernel = a pure a-trous level (levels 1.. of the fused chain; every level otherwise)
        pure = per_level[1:] if (fused and L > 1) else per_level
        t_atrous = float(np.mean(pure)) if pure else 0.0
        bracketed = fused and L > 1 and all(len(x) == 2 for x in lt2)
        if bracketed:  # average launch duration over the L - 1 pure levels, timed as one interval
            t_atrous = avg([x[1] for x in lt2]) * 1e-6 / (L - 1)
            atrous_samples = [round(float(x[1]) / (L - 1), 2) for x in lt2]
    else:
        t_temporal = avg([e["t0"].elapsed_time(e["t1"]) for e in ev]) * 1e-3
        per_level = [avg([e["levels"][i][0].elapsed_time(e["levels"][i][1]) for e in ev]) * 1e-3 for i in range(L)]
        # (with N > 1 the first level's interval also holds the halo exchange it overlaps with, and every level but the last
        # filters a few extra rows: the roofline line is then taken over the levels after the first)
        t_atrous = float(np.mean(per_level if L == 1 else per_level[1:]))
        t_svgf_chain = t_temporal + sum(per_level)
    own_px = (w.own[1] - w.own[0]) * GW                      # pixels a rank owns
    gb = noisy = consts = None
    if rank == 0 and args.cpu_frames > 0 and world == 1:
        gb = {"albedo": r.svgf.download(PLANE_ALBEDO, 0), "rough_metal": r.svgf.download(PLANE_ROUGH_METAL, 0),
              "world_pos": r.svgf.download(PLANE_WORLDPOS, 0), "normal": r.svgf.download(PLANE_NORMAL, 0),
              "depth": r.svgf.download(PLANE_DEPTH, 0)}
        noisy = w.noisy_dev[0].cpu().numpy() if not do_gi else None
        consts = r.global_constants()
    parallelism = w.parallelism_label()
    halo_rows = part.halo
    w.destroy()

    # ---- N > 1: the weak-scaling frame (per-GPU work fixed) and BASELINE.json configs[4] ----
    weak = None
    if world > 1 and not args.no_weak and do_gi:
        a, b = strips.frame_factors(world)
        ww = Workload(rt, args, args.width * a, args.height * b, L, args.spp, sc, cam, rank, world, local_rank, group, scheme=scheme, link=link)
        dtw, rays_w = ww.timed(args.steps, max(args.warmup, 1))
        weak = {"frames_per_s_1080p_equivalents": args.steps / dtw * world, "global_frames_per_s": args.steps / dtw, "ms_per_frame": dtw / args.steps * 1e3,
                "mrays_per_s": rays_w / dtw / 1e6, "global_width": args.width * a, "global_height": args.height * b,
                "rows_per_strip": args.height * b // world, "parallelism": ww.parallelism_label(), "frames_in_flight": ww.frames_in_flight()}
        ww.destroy()
    config5 = None
    if do_gi and (args.config5 or world == 8):
        config5 = run_config5(rt, args, sc, rank, world, local_rank, group, scheme, link)

    if rank == 0:
        fps = args.steps / dt                                 # whole-job frames per second of the ONE frame
        achieved = ATROUS_BYTES_PX * own_px / t_atrous / 1e9 if t_atrous > 0 else 0.0
        traffic = measured_traffic(GW, GH, L) if world == 1 else None
        valu = measured_valu(GW, GH, L) if world == 1 else None
        if valu and t_atrous > 0:
            valu["busy_frac"] = valu["issue_us_per_launch"] / (t_atrous * 1e6)
        svgf_bytes = (TEMPORAL_BYTES_PX + ATROUS_BYTES_PX * L) * own_px
        # the fused chain's own byte model: temporal reads 60 B + moments / variance writes 6 B + level 0's output 16 B = 82 B/px for
        # temporal + level 0 together (the accumulated radiance is neither written nor re-read: 46 B/px less), then 46 B/px per level
        fused_bytes = (TEMPORAL_BYTES_PX + ATROUS_BYTES_PX * max(L - 1, 0)) * own_px
        frame_gbps = svgf_bytes * world * fps / 1e9
        out = {
            "metric": ("denoised frames/s (1920x1080: GI 1 spp + SVGF temporal + a-trous)" if do_gi else
                       "denoised frames/s (1920x1080: SVGF temporal + a-trous only)"),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_run": warmup_run,
            # a second timed region of the same K steps after `settled_after_frames` frames of this context (value: after W only)
            "value_settled": args.steps / dt_settled, "ms_per_step_settled": dt_settled / args.steps * 1e3,
            # ... and with one frame in flight (None when `value` already is that): value's frames overlap the next frame's closest-hit walk
            "value_one_frame_in_flight": (args.steps / dt_serial) if dt_serial else None,
            "settled_after_frames": warmup_run + args.steps + settle_run,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not args.scene else f"scene file {os.path.basename(args.scene)}; synthetic camera",
            "config": {"workload": workload_label(GW, GH, args.spp, L, do_gi),
                       "global_width": GW, "global_height": GH, "atrous_levels": L, "spp": args.spp,
                       "parallelism": parallelism, "link": (link.label() if link is not None else None), "halo_rows": halo_rows, "rows_per_strip": GH // world,
                       "frames_in_flight": frames_in_flight,
                       # what the metric's own strong-scaling curve cannot exceed at N strips, before a byte is exchanged: one strip's frame alone on
                       # a GPU (measured with the exchange stubbed, strips.STRIP_FRAME_US_1080P) -- to read a SCALE_rNN.json against
                       "strong_scaling_ceiling": (strips.strong_scaling_ceilings() if (GW, GH) == (1920, 1080) else None),
                       "scene_device_bytes": scene_bytes, "bvh": bvh, "library_build_id": library_build_id()},
            "frames_per_s_with_final_gather": fps_with_gather,
            # rays = W H spp (1 + hit fraction) per frame: one bounce ray per pixel + one sun-visibility QUERY per hit.  Most of the
            # queries are answered by the sun table inside the shade pass; `traced_*` counts only the rays a traverser walked (bounce rays + the
            # shadow rays left in the lists) -- the figure to read as traverser throughput
            "mrays_per_s": (rays_total / dt / 1e6) if do_gi else None,
            "traced_mrays_per_s": ((rays_total - answered_total) / dt / 1e6) if do_gi else None,
            "gi_kernel_mrays_per_s": (rays_ev / NPROF / t_gi / 1e6) if do_gi else None,
            "gi_kernel_traced_mrays_per_s": ((rays_ev - answered_ev) / NPROF / t_gi / 1e6) if do_gi else None,
            "rays_per_frame": {"queries": rays_total / args.steps, "traced": (rays_total - answered_total) / args.steps,
                               "answered_by_sun_table": answered_total / args.steps} if do_gi else None,
            "weak_scaling": weak,
            "config5": config5,
            "kernel_us": {"gi_trace": t_gi * 1e6, "temporal": None if fused else t_temporal * 1e6,
                          "fused_temporal_level0": per_level[0] * 1e6 if (fused and L > 0) else None,
                          "atrous_levels": [t * 1e6 for t in per_level], "svgf_chain": t_svgf_chain * 1e6},
            "roofline": {"bound": "hbm",
                         "kernel": (("svgf_atrous_lds_kernel, a pure level (levels 1.. of the fused chain timed as ONE interval of the library's own HIP events on the "
                                     "launch stream, divided by their number; kernel_us.atrous_levels has an event pair per level, ~2 us of packets each)"
                                     if bracketed else
                                     "svgf_atrous_lds_kernel, a pure level (mean over levels 1.. of the fused chain: the library's own HIP events on the launch stream)")
                                    if fused else "svgf_atrous_lds_kernel (mean over the levels of a frame)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": ATROUS_BYTES_PX * own_px,
                         # the per-frame samples behind `achieved` (us per pure level; the average leaves out the largest and the smallest)
                         "launch_us_samples": atrous_samples,
                         "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None,
                         # the contract prices this kernel against HBM; what binds it in fact is instruction issue
                         "limiter": "VALU issue: 25 taps x (12 fp32 operations -- ten of them as packed pairs in four taps of five -- + v_log + v_exp) per pixel; the 30 ds_read_b128 per pixel keep the LDS pipe ~75 % busy beside it" if valu else None, "valu": valu},
            "temporal_roofline": (None if (fused or t_temporal <= 0) else
                                  {"achieved": TEMPORAL_BYTES_PX * own_px / t_temporal / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": TEMPORAL_BYTES_PX * own_px / t_temporal / 1e9 / HBM_PEAK_GBPS}),
            # the same byte model over longer intervals: all of SVGF (temporal + L levels: SURVEY 8d's (82 + 46 L) B/px over the
            # chain's time), the chain's own fused byte model beside it, and the whole frame (GI included in the time, SVGF's
            # algorithmic bytes only -- the GI stage is not HBM-priced)
            "svgf_roofline": {"achieved": svgf_bytes / t_svgf_chain / 1e9, "frac": svgf_bytes / t_svgf_chain / 1e9 / HBM_PEAK_GBPS, "unit": "GB/s",
                              "bytes_per_px": TEMPORAL_BYTES_PX + ATROUS_BYTES_PX * L, "us": t_svgf_chain * 1e6},
            "svgf_fused_model": ({"bytes_per_px": TEMPORAL_BYTES_PX + ATROUS_BYTES_PX * max(L - 1, 0), "achieved": fused_bytes / t_svgf_chain / 1e9,
                                  "frac": fused_bytes / t_svgf_chain / 1e9 / HBM_PEAK_GBPS, "unit": "GB/s",
                                  "fused_launch": {"bytes_per_px": TEMPORAL_BYTES_PX, "us": per_level[0] * 1e6,
                                                   "frac": TEMPORAL_BYTES_PX * own_px / per_level[0] / 1e9 / HBM_PEAK_GBPS,
                                                   # the same launch under SURVEY 8d's unfused model: it does the work of the temporal
                                                   # pass (82 B/px) and of level 0 (46 B/px)
                                                   "bytes_per_px_8d": TEMPORAL_BYTES_PX + ATROUS_BYTES_PX,
                                                   "frac_8d": (TEMPORAL_BYTES_PX + ATROUS_BYTES_PX) * own_px / per_level[0] / 1e9 / HBM_PEAK_GBPS,
                                                   "traffic": traffic[2] if traffic else None},
                                  "note": "temporal + level 0 as one launch: 60 B read + 6 B moments / variance + 16 B level-0 output = 82 B/px; "
                                          "the accumulated radiance is never written or re-read"} if (fused and L > 0) else None),
            "frame_roofline": {"achieved": frame_gbps, "frac": frame_gbps / HBM_PEAK_GBPS, "unit": "GB/s",
                               "note": "SVGF algorithmic bytes / whole-frame time (GI included)"},
            "frame_algorithmic_GBps": frame_gbps,
        }
        if gb is not None:
            out["cpu_baseline"] = cpu_baseline(GW, GH, L, args.cpu_frames, gb, consts, sc, noisy, do_gi)
        (emit or (lambda line: print(line, flush=True)))(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
