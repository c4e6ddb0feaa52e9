"""bench.py's contract, exercised on the GPU box: the one-line JSON of a short single-GPU run, and a two-rank rehearsal
of the N > 1 path on ONE GPU (both ranks on cuda:0, gloo instead of RCCL, rows staged through the host)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--cpu-frames", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "svgf_roofline", "svgf_fused_model", "frame_roofline", "kernel_us"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    # two frames in flight by default: the next frame's closest-hit walk (neb_gi_trace_begin) beside this frame's shadow pass and SVGF chain;
    # the rate with one frame in flight rides beside
    assert d["config"]["frames_in_flight"] == 2 and "neb_gi_trace_begin" in d["config"]["parallelism"] and d["value_one_frame_in_flight"] > 0
    assert d["warmup"] == 2 and d["warmup_run"] == 2 and d["value_settled"] > 0 and d["settled_after_frames"] >= 48
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 8000.0
    # one GPU: temporal + level 0 ran as one launch, the pure levels are timed by the library's own events
    ku = d["kernel_us"]
    assert ku["temporal"] is None and ku["fused_temporal_level0"] > 0 and len(ku["atrous_levels"]) == 5 and all(t > 0 for t in ku["atrous_levels"])
    assert abs(ku["svgf_chain"] - sum(ku["atrous_levels"])) < 0.25 * ku["svgf_chain"]
    assert d["svgf_fused_model"]["bytes_per_px"] == 82 + 46 * 4 and d["svgf_roofline"]["bytes_per_px"] == 82 + 46 * 5
    assert len(d["config"]["library_build_id"]) == 16
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["cpu_baseline"]["cores_available"] >= d["cpu_baseline"]["cores"]
    one = d["cpu_baseline"]["single_thread"]
    assert one["cores"] == 1 and 0 < one["value"] <= d["cpu_baseline"]["value"] * 1.5
    assert "workload" in d["config"] and "69 textures of 1024^2" in d["config"]["workload"]
    assert d["config"]["scene_device_bytes"]["texture_tables"] > 7.5e8  # 24 material bundles of 1024^2 x 32 B
    assert 0 < d["frame_roofline"]["frac"] < d["svgf_roofline"]["frac"] < 1


def test_single_gpu_line_with_the_separate_kernels():
    """NEB_BENCH_NO_FUSE=1: option svgf_fuse = 0 -- the temporal pass as its own kernel, level times still from the library's events."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-frames", "0", "--tex-size", "256"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, NEB_BENCH_NO_FUSE="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    ku = d["kernel_us"]
    assert ku["temporal"] > 0 and ku["fused_temporal_level0"] is None and len(ku["atrous_levels"]) == 5 and all(t > 0 for t in ku["atrous_levels"])
    assert d["svgf_fused_model"] is None and 0 < d["temporal_roofline"]["frac"] < 1 and 0 < d["roofline"]["frac"] < 1


def test_scene_file_replaces_the_stand_in():
    """bench.py --scene <file>: a real glTF file through the same loader (what a dropped-in Sponza.glb would take)."""
    scene = os.path.join(ROOT, "tests", "golden", "DamagedHelmet_jpeg.glb")
    if not os.path.exists(scene):
        pytest.skip("tests/golden/DamagedHelmet_jpeg.glb is not present (an optional third-party asset: tests/golden/README.md)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-frames", "0", "--scene", scene,
                        "--width", "1280", "--height", "720", "--levels", "3"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert "DamagedHelmet_jpeg.glb 1280x720 (15452 triangles, 1 submeshes, 1 materials, 3 textures of 2048^2)" in d["config"]["workload"]
    assert d["config"]["bvh"]["triangles"] == 15452 and d["value"] > 0 and d["mrays_per_s"] > 0
    assert "scene file DamagedHelmet_jpeg.glb" in d["data"]


@pytest.mark.parametrize("scheme", ["once", "per_level"])
def test_two_rank_rehearsal(scheme):
    env = dict(os.environ, NEB_BENCH_SHARE_DEVICE="1", NEB_BENCH_BACKEND="gloo", NEB_STRIPS_SCHEME=scheme)
    if scheme == "per_level":
        env["NEB_BENCH_PIPELINE"] = "defer"  # (the form 135-row strips take; 540-row strips take "split" by themselves)
    # "once": the driver's own form, `python bench.py --gpus 2 ...` with no launcher around it (bench.py starts its ranks itself);
    # "per_level": under torch.distributed.run, as the contract also allows
    launcher = [] if scheme == "once" else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                            "--master-port", str(29650 + (os.getpid() % 200))]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--cpu-frames", "0", "--gather", "--tex-size", "256"] + (["--config5", "--config5-frames", "4"] if scheme == "once" else [])
    env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    # `value` = the metric's own curve: ONE 1920x1080 frame in two strips; the weak-scaling frame rides beside it
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong" and d["config"]["global_height"] == 1080 and d["config"]["rows_per_strip"] == 540
    assert ("one exchange per frame" if scheme == "once" else "one exchange per a-trous level") in d["config"]["parallelism"]
    assert f"scheme '{scheme}'" in d["config"]["parallelism"] and "transport 'torch'" in d["config"]["parallelism"]
    assert d["frames_per_s_with_final_gather"] > 0
    # 540-row strips take the "split" form of two frames in flight; "per_level" is made to take the "defer" form of the smallest strips
    assert d["config"]["frames_in_flight"] == (2 if scheme == "once" else 3)
    assert ("neb_gi_trace_begin" if scheme == "once" else "two record sets") in d["config"]["parallelism"]
    assert d["config"]["link"].startswith("link measured: ") and d["value_settled"] > 0 and d["warmup_run"] == 2
    assert d["weak_scaling"]["global_height"] == 2160 and d["weak_scaling"]["rows_per_strip"] == 1080 and d["weak_scaling"]["frames_per_s_1080p_equivalents"] > 0
    if scheme == "once":
        c5 = d["config5"]
        assert c5["reference_policy"]["frames_denoised"] == 2 and c5["always_on"]["frames_denoised"] == 4
        assert c5["reference_policy"]["mrays_per_s"] > 0
