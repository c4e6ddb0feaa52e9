#!/bin/bash
# tuning only: bench each build_variants/lib_<name>.so given on the command line (run on the GPU box)
for v in "$@"; do
  NEB_LIB_PATH=$PWD/build_variants/lib_$v.so timeout -k 10 120 python bench.py --cpu-frames 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python - "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"], 1), "fps  gi_us", round(d["kernel_us"]["gi_trace"], 1))
PY
done
