#!/bin/bash
# context numbers, not bench lines: the frame at other sizes on one GPU (event-timed per-kernel times beside the frame time).  usage (GPU box): bash tools/size_sweep.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
for cfg in "1280 720 1" "1920 1080 1" "2560 1440 1" "3840 2160 1" "1920 1080 4" "3840 2160 4"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --width $1 --height $2 --spp $3 --steps 24 --warmup 8 --cpu-frames 0 > gpurun_out/size_$1x$2_$3.json 2> gpurun_out/size_$1x$2_$3.err || { tail -5 gpurun_out/size_$1x$2_$3.err; exit 1; }
  python - "$1x$2, $3 spp" gpurun_out/size_$1x$2_$3.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
k = j["kernel_us"]
lv = k["atrous_levels"][1:]
print("| %s | %.3f (one frame in flight %.3f) | %.0f µs | %.1f | %.1f µs | %.1f µs, %.2f | %.0f µs |" % (
    sys.argv[1], j["ms_per_step"], 1e3 / j["value_one_frame_in_flight"] if j.get("value_one_frame_in_flight") else float("nan"), k["gi_trace"],
    j.get("gi_kernel_mrays_per_s", 0) / 1e3, k["fused_temporal_level0"] or 0, sum(lv) / len(lv), j["roofline"]["frac"], k["svgf_chain"]))
PY
done
