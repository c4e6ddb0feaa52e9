"""One rank of the CPU rehearsal of `python bench.py --gpus N` as bench.launch_ranks starts it: bench.main with the device side
replaced by tests/test_bench_dryrun.DryRuntime (gloo, an oracle-backed strip renderer).  Test infrastructure only.
NEB_DRY_CHILD_FAIL_RANK=<r>: that rank exits with code 3 at once while the others would wait in the rendezvous."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    if os.environ.get("NEB_DRY_CHILD_FAIL_RANK") == os.environ.get("RANK"):
        print("dry child: failing on purpose", file=sys.stderr)
        sys.exit(3)
    import bench
    from test_bench_dryrun import DryRuntime
    bench.Workload.SETTLE_FRAMES = 4
    bench.main(sys.argv[1:], rt=DryRuntime())
