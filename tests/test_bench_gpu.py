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
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 8000.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"]


@pytest.mark.parametrize("scheme", ["once", "per_level"])
def test_two_rank_rehearsal(scheme):
    env = dict(os.environ, NEB_BENCH_SHARE_DEVICE="1", NEB_BENCH_BACKEND="gloo", NEB_STRIPS_SCHEME=scheme)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29650 + (os.getpid() % 200)), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--cpu-frames", "0", "--gather"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_height"] == 2160
    assert ("one per frame" if scheme == "once" else "one per a-trous level") in d["config"]["parallelism"]
    assert d["frames_per_s_with_final_gather"] > 0
