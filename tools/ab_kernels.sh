#!/bin/bash
# tuning only: per-kernel mean durations (rocprofv3 --kernel-trace --stats) of each build_variants/lib_<name>.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export NEB_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/lib_$v.so
  rm -rf gpurun_out/abk_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$v -- python bench.py --steps 24 --warmup 8 --cpu-frames 0 --overlap off > gpurun_out/abk_$v.log 2>&1 || exit 1
  python - "$v" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/abk_{sys.argv[1]}/*/*kernel_stats.csv")[0]
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
pick = lambda s: sum(v for k, v in rows.items() if s in k)
print(sys.argv[1], "raygen_trace %.1f shade %.1f shadow %.1f" % (pick("gi_raygen_trace"), pick("gi_shade"), pick("gi_shadow_")))
PY
done
