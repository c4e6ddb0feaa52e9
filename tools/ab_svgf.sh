#!/bin/bash
# tuning only: per-kernel mean durations of the SVGF kernels (rocprofv3 --kernel-trace --stats over `bench.py --svgf-only`)
# for each build_variants/lib_<name>.so given ("product" = the in-tree library).  AB_TEST=1 also runs the SVGF parity tests.
# AB_FULL=1 profiles the whole frame (GI + SVGF) instead: the fused level-0 kernel then follows the GI kernels, as in the product.
# usage (GPU box): bash tools/ab_svgf.sh product pipe1 ...
case " $* $AB_BENCH_FLAGS " in *" --gpus "*) echo "$0 refuses --gpus: under rocprofv3 bench.py would start its ranks from a process the profiler has given a GPU (profile each rank's own command instead)"; exit 2;; esac
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
mode=--svgf-only; [ -n "$AB_FULL" ] && mode=
for v in "$@"; do
  if [ "$v" = product ]; then unset NEB_LIB_PATH; else export NEB_LIB_PATH=$root/build_variants/lib_$v.so; fi
  if [ -n "$AB_TEST" ]; then
    timeout -k 10 400 python -m pytest tests/test_svgf_gpu.py -x -q > gpurun_out/absvgf_$v.test.log 2>&1; echo "$v tests: $(tail -1 gpurun_out/absvgf_$v.test.log)"
  fi
  rm -rf gpurun_out/absvgf_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/absvgf_$v -- python bench.py $mode --steps 24 --warmup 8 --cpu-frames 0 --overlap off $AB_BENCH_FLAGS > gpurun_out/absvgf_$v.log 2>&1 || { tail -5 gpurun_out/absvgf_$v.log; exit 1; }
  python - "$v" <<'PY'
import csv, glob, re, sys
f = glob.glob(f"gpurun_out/absvgf_{sys.argv[1]}/*/*kernel_stats.csv")[0]
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
lv = sorted((int(re.search(r"<(\d+)", k).group(1)), v) for k, v in rows.items() if "atrous" in k and "<" in k)
other = {k.split("(")[0].split("::")[-1]: v for k, v in rows.items() if ("svgf" in k and "<" not in k)}
tot = sum(v for _, v in lv) + sum(other.values())
print(sys.argv[1], "levels", " ".join(f"S{s}:{v:.1f}" for s, v in lv), "mean %.2f" % (sum(v for _, v in lv) / max(len(lv), 1)), "|", " ".join(f"{k}:{v:.1f}" for k, v in other.items()), "| svgf total %.1f us" % tot)
PY
done
