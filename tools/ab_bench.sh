#!/bin/bash
# tuning only: the bench line (no profiler) under different environment settings, same box.  usage (GPU box): bash tools/ab_bench.sh "" "NEB_BENCH_SIDE_PRIORITY=1" ...
case " $* $AB_BENCH_FLAGS " in *" --gpus "*) echo "$0 refuses --gpus"; exit 2;; esac
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
k=0
for setting in "$@"; do
  k=$((k + 1))
  ( for kv in $setting; do export "$kv"; done
    timeout -k 10 300 python bench.py --steps ${AB_STEPS:-64} --warmup ${AB_WARMUP:-16} --cpu-frames 0 $AB_BENCH_FLAGS > gpurun_out/abbench_$k.json 2> gpurun_out/abbench_$k.err ) || { tail -5 gpurun_out/abbench_$k.err; exit 1; }
  python - "$setting" gpurun_out/abbench_$k.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("%-60s value %.1f settled %s one-in-flight %s | %.4f ms" % (sys.argv[1] or "(default)", j["value"], j.get("value_settled"), j.get("value_one_frame_in_flight"), j["ms_per_step"]))
PY
done
