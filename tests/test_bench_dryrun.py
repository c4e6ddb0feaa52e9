"""CPU dry run of bench.py's N > 1 control flow, rank for rank: `python bench.py --gpus 8` as the driver launches it, with
the device side replaced -- gloo instead of RCCL, an oracle-backed strip renderer instead of the HIP library, no-op events --
and the frames shrunk.  What it pins is everything the first multi-GPU run can trip over that is NOT a kernel: the partition
of every leg (the metric's own 1080p-style frame in N strips, the weak-scaling frame, BASELINE.json configs[4] with its
moving-then-still camera), the collectives every rank must enter in the same order (barriers, MAX / SUM reductions, the halo
exchanges of both schemes, the final gather), and the shape of the one JSON line.  Test infrastructure: the stand-ins live
here, bench.py itself never imports them."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Event:
    def record(self, stream=None):
        pass

    def elapsed_time(self, other):
        return 0.05  # ms


class _Stream:
    cuda_stream = 0

    def wait_event(self, e):
        pass


class DryRuntime:
    """bench.GpuRuntime's interface on the CPU."""
    name = "dry"

    def available(self):
        return True

    def set_device(self, local_rank):
        pass

    def init_process_group(self, dist, local_rank):
        dist.init_process_group("gloo")

    def current_stream(self):
        return _Stream()

    def stream_handle(self, stream):
        return 0

    def new_stream(self):
        return _Stream()

    def synchronize(self):
        pass

    def event(self, timing=True):
        return _Event()

    def event_handle(self, event):
        return 0

    def to_device(self, t):
        return t

    def make_renderer(self, part, rank, local_rank, group, exchange):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_backend import OracleDenoiser

        from nebulae_amd import strips, synth
        from nebulae_amd.svgf import PLANE_DEPTH, PLANE_NORMAL, PLANE_RADIANCE

        class DryStripRenderer(strips.StripRenderer):
            """The strip renderer with every GI / G-buffer entry point of the HIP library replaced by cheap host fills; the
            SVGF side (what the strips exchange and filter) is the oracle-backed denoiser of the gloo strip tests."""

            def __init__(self):
                super().__init__(part, rank, device=0, group=group, denoiser_factory=OracleDenoiser, exchange="torch")
                self._rays = 0
                self._g = synth.synth_gbuffer(part.W, part.H)

            def init_pathtracer_scene(self, scene, stream=0):
                self._scene = scene

            def scene_info(self):
                return 12, 3

            def scene_bytes(self):
                return {"texture_tables": 0, "triangles": 0, "bvh_nodes": 0}

            def bvh_depth(self):
                return 2

            def build_ms(self):
                return 0.0

            def sun_table_stats(self):
                return {"lit_plus": 0, "lit_minus": 0, "rays_answered": 0, "builds": 0}

            def sun_table_build_ms(self):
                return None

            def shadow_tail_mode(self):
                return 0, (0.0, 0.0)

            def _rows(self):
                return self.svgf.row_begin, self.svgf.row_end

            def submit_commands_gbuffer(self):
                r0, r1 = self._rows()
                cur = self.svgf.get_current_resource_index()
                self.svgf.plane_tensor(PLANE_DEPTH, cur).copy_(torch.from_numpy(self._g["depth"][r0:r1].view(np.int32)))
                self.svgf.plane_tensor(PLANE_NORMAL, cur).copy_(torch.from_numpy(self._g["normal"][r0:r1]))

            def submit_commands_pbr_lighting(self):
                r0, r1 = self._rows()
                cur = self.svgf.get_current_resource_index()
                self.svgf.plane_tensor(PLANE_RADIANCE, cur).copy_(torch.from_numpy(synth.synth_radiance(self._g["base"][r0:r1], 1)))

            def set_defer_resolve(self, on=True):  # (bench.py --overlap: the indirect term stays in the GI records until neb_gi_resolve)
                assert int(on) in (0, 1, 2)
                self._defer, self._pending = bool(on), None

            def submit_commands_gi_pathtrace(self, rows=None, stream=None):
                own = self.part.gi_rows(self.rank) if rows is None else rows
                self._rays += 2 * (own[1] - own[0]) * self.part.W * int(self.gi_ui.gi_samples_per_pixel)
                add = (own, 0.01 * float(self.info.frame_index % 7))
                if getattr(self, "_defer", False):
                    self._pending = add
                else:
                    self._add(add)

            def submit_commands_gi_pathtrace_begin(self, rows=None, stream=None):  # (bench.py's "split" form: the walk now, shading + adding later)
                assert getattr(self, "_begun", None) is None, "neb_gi_trace_begin: the dry run keeps one dispatch begun"
                own = self.part.gi_rows(self.rank) if rows is None else rows
                self._rays += 2 * (own[1] - own[0]) * self.part.W * int(self.gi_ui.gi_samples_per_pixel)
                self._begun = (own, 0.01 * float(self.info.frame_index % 7))

            def submit_commands_gi_pathtrace_finish(self, stream=None, after_shade_event=None):
                assert self._begun is not None, "neb_gi_trace_finish: nothing begun"
                self._add(self._begun)
                self._begun = None

            def submit_commands_gi_resolve(self, stream=None):
                assert self._defer and self._pending is not None, "neb_gi_resolve: nothing pending"
                self._add(self._pending)
                self._pending = None

            def _add(self, add):
                own, v = add
                r0, _ = self._rows()
                t = self.svgf.plane_tensor(PLANE_RADIANCE, self.svgf.get_current_resource_index())
                t[own[0] - r0:own[1] - r0, :, :3] += v  # "adds into radiance[cur]"

            def ray_count(self, reset=False):
                v = self._rays
                if reset:
                    self._rays = 0
                return v

        return DryStripRenderer()


def _rank_main(rank, world, port, argv, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    bench.Workload.SETTLE_FRAMES = 4  # (the dry run has no clocks to settle: keep the CPU suite short)
    lines = []
    bench.main(argv, rt=DryRuntime(), emit=lines.append)
    if rank == 0:
        assert len(lines) == 1
        open(out_path, "w").write(lines[0])
    else:
        assert not lines


@pytest.mark.parametrize("world,scheme", [(8, "auto"), (2, "per_level"), (4, "overlap")])
def test_multi_gpu_control_flow_on_cpu(tmp_path, world, scheme):
    # 8 strips need >= 62 rows each for the one-exchange scheme at 5 levels: 64 x 512 frames; the config-5 leg runs at its own
    # (shrunk) size with its moving-then-still camera; --gather adds the final gather after every frame
    out = tmp_path / "line.json"
    argv = ["--gpus", str(world), "--steps", "2", "--warmup", "2", "--width", "64", "--height", "512", "--cpu-frames", "0", "--triangles", "2000",
            "--tex-size", "16", "--gather", "--scheme", scheme, "--config5", "--config5-frames", "4", "--config5-size", "64", "512"]
    port = 29900 + (os.getpid() % 1500) + world
    # strips this small take the "defer" form of frames in flight (three: two record sets); world 2 is made to take the "split" form a
    # whole frame or a big strip takes (two: neb_gi_trace_begin of the next frame on a side stream)
    form = "split" if world == 2 else "defer"
    os.environ["NEB_BENCH_PIPELINE"] = form
    try:
        mp.spawn(_rank_main, args=(world, port, argv, str(out)), nprocs=world, join=True)
    finally:
        del os.environ["NEB_BENCH_PIPELINE"]
    d = json.loads(out.read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "weak_scaling", "config5", "kernel_us", "svgf_roofline", "frame_roofline"):
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["vs_baseline"] is None
    cfg = d["config"]
    assert cfg["global_width"] == 64 and cfg["global_height"] == 512 and cfg["rows_per_strip"] == 512 // world
    want_scheme = "once" if scheme == "auto" else scheme
    assert f"scheme '{want_scheme}'" in cfg["parallelism"] and f"row-strips x{world}" in cfg["parallelism"]
    a, b = {8: (2, 4), 4: (2, 2), 2: (1, 2)}[world]
    ws = d["weak_scaling"]
    assert ws["global_width"] == 64 * a and ws["global_height"] == 512 * b and ws["frames_per_s_1080p_equivalents"] > 0
    c5 = d["config5"]
    assert c5["reference_policy"]["frames_denoised"] == 2 and c5["always_on"]["frames_denoised"] == 4  # skip while moving, then reset
    assert c5["reference_policy"]["mrays_per_s"] > 0
    assert d["frames_per_s_with_final_gather"] > 0
    assert d["mrays_per_s"] > 0 and len(d["kernel_us"]["atrous_levels"]) == 5
    # the contract's warm-up count is what ran in front of `value`; the settled rate rides beside it
    assert d["warmup"] == 2 and d["warmup_run"] == 2 and d["value_settled"] > 0 and d["settled_after_frames"] >= 4
    # the scheme chooser was fed a link measured at start-up (20 exchanges of 4 KB and of 2 MB with a neighbour), not guesses
    assert cfg["link"].startswith("link measured: ") and "20 exchanges of 4096 B" in cfg["link"] and "of 2097152 B" in cfg["link"]
    if scheme == "auto":
        assert "cost table: once +" in cfg["parallelism"] and "link measured" in cfg["parallelism"]
    # strips this small run two frames in flight (--overlap auto): GI of frame f + 1 on a side stream, deferred resolve
    if form == "defer":
        assert cfg["frames_in_flight"] == 3 and "3 frames in flight" in cfg["parallelism"] and "two record sets" in cfg["parallelism"]
    else:
        assert cfg["frames_in_flight"] == 2 and "neb_gi_trace_begin" in cfg["parallelism"]
    assert d["value_one_frame_in_flight"] > 0
    assert d["weak_scaling"]["frames_in_flight"] == cfg["frames_in_flight"]


_ARGV = ["--steps", "2", "--warmup", "1", "--width", "64", "--height", "256", "--cpu-frames", "0", "--triangles", "2000", "--tex-size", "16",
         "--no-weak", "--levels", "3"]


def test_launcher_starts_the_ranks_and_relays_rank0s_line():
    """`python3 bench.py --gpus N` with no launcher around it: bench.launch_ranks is what main() calls then (WORLD_SIZE unset).
    Here with the CPU stand-in as the child program: two ranks under gloo, ONE JSON line relayed, exit code 0."""
    sys.path.insert(0, ROOT)
    import bench
    lines = []
    child = [sys.executable, os.path.join(ROOT, "tests", "bench_dry_child.py")]
    rc = bench.launch_ranks(2, ["--gpus", "2"] + _ARGV, child=child, emit=lines.append, timeout=600)
    assert rc == 0
    js = [l for l in lines if l.startswith("{")]
    assert len(js) == 1
    d = json.loads(js[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "row-strips x2" in d["config"]["parallelism"]


def test_launcher_stops_every_rank_when_one_fails():
    sys.path.insert(0, ROOT)
    import time

    import bench
    lines = []
    child = [sys.executable, os.path.join(ROOT, "tests", "bench_dry_child.py")]
    os.environ["NEB_DRY_CHILD_FAIL_RANK"] = "1"
    try:
        t0 = time.time()
        rc = bench.launch_ranks(2, ["--gpus", "2"] + _ARGV, child=child, emit=lines.append, timeout=600)
    finally:
        del os.environ["NEB_DRY_CHILD_FAIL_RANK"]
    assert rc == 3 and not lines and time.time() - t0 < 120  # rank 0 was waiting in the rendezvous: terminated, not waited for


def test_bare_bench_gpus_2_is_its_own_launcher_and_fails_loudly_without_a_gpu():
    """The driver's form, `python3 bench.py --gpus 2 ...`, typed in a container without a GPU: the parent must start two ranks
    (no "launch with torch.distributed.run" refusal), each rank must refuse to run without a device (no CPU fallback), and the
    parent must exit non-zero."""
    import subprocess
    if torch.cuda.is_available():  # (before anything is started: on a GPU box this would be a real two-rank 1080p bench)
        pytest.skip("a GPU is present: covered by tests/test_bench_gpu.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-frames", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "exited with code" in p.stderr and "torch.distributed.run" not in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_the_launcher_refuses_to_start_ranks_from_under_a_profiler():
    """rocprofv3 preloads a tool library that initialises the GPU before bench.py's main() runs: starting the ranks from such a process would
    be an exec out of a fork of a process that holds a GPU.  `bench.py --gpus N` then exits with 2 and says what to do instead."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/librocprofiler-sdk-tool.so"  # (the variable alone: nothing is preloaded here)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-frames", "0"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert p.returncode == 2 and "refuses to start ranks from under a profiler" in p.stderr
    for script in ("profile_round.sh", "ab_env.sh", "ab_svgf.sh"):
        q = subprocess.run(["bash", os.path.join(ROOT, "tools", script), "tag", "--gpus", "2"], capture_output=True, text=True, timeout=60, cwd=ROOT,
                           env=dict(os.environ, AB_BENCH_FLAGS="--gpus 2"))
        assert q.returncode == 2 and "refuses --gpus" in q.stdout, (script, q.stdout, q.stderr)
