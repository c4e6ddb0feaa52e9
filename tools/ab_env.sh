#!/bin/bash
# tuning only: per-kernel mean durations (rocprofv3 --kernel-trace --stats) of `python bench.py` under different environment
# settings.  usage (GPU box):  bash tools/ab_env.sh "NEB_PLOC_BUDGET=0" "NEB_PLOC_BUDGET=8388608 NEB_PLOC_RADIUS=32" ...
case " $* $AB_BENCH_FLAGS " in *" --gpus "*) echo "$0 refuses --gpus: under rocprofv3 bench.py would start its ranks from a process the profiler has given a GPU (profile each rank's own command instead)"; exit 2;; esac
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
k=0
for setting in "$@"; do
  k=$((k + 1))
  d=gpurun_out/abenv_$k
  rm -rf "$d"
  ( for kv in $setting; do export "$kv"; done
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python bench.py --steps 24 --warmup 8 --cpu-frames 0 --overlap off $AB_BENCH_FLAGS > "$d.log" 2>&1 ) || { tail -5 "$d.log"; exit 1; }
  python - "$setting" "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + "/*/*kernel_stats.csv")[0]
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
tot = {r["Name"]: float(r["TotalDurationNs"]) / 1e3 for r in csv.DictReader(open(f))}
pick = lambda s: sum(v for k, v in rows.items() if s in k)
build = sum(v for k, v in tot.items() if "ploc_" in k or "collapse_" in k)
import json
line = [l for l in open(sys.argv[2] + ".log") if l.startswith("{")]
j = json.loads(line[-1]) if line else {"value": 0, "kernel_us": {"gi_trace": 0}}
print("%-44s raygen_trace %.1f resume %.1f shade %.1f shadow %.1f | GI (events) %.1f us, %.1f fps" % (sys.argv[1], pick("gi_raygen_trace"), pick("gi_resume"), pick("gi_shade"), pick("gi_shadow_"), j["kernel_us"]["gi_trace"], j["value"]))
PY
done
