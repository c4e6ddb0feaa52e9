#!/bin/bash
# tuning only: durations of the sun-table build kernels (rocprofv3 --kernel-trace --stats over one short bench run) and the sides proven lit, for the in-tree
# library ("product") or build_variants/lib_<name>.so; environment settings may be given as name:VAR=value.  usage (GPU box): bash tools/ab_sun_table.sh product sw2
case " $* " in *" --gpus "*) echo "$0 refuses --gpus"; exit 2;; esac
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
k=0
for spec in "$@"; do
  k=$((k + 1))
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  d=gpurun_out/absun_$k
  rm -rf "$d"
  ( if [ "$v" != product ]; then export NEB_LIB_PATH=$root/build_variants/lib_$v.so; fi
    for kv in $envs; do export "$kv"; done
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python bench.py --steps 8 --warmup 2 --cpu-frames 0 --overlap off > "$d.log" 2>&1 ) || { tail -5 "$d.log"; exit 1; }
  python - "$spec" "$d" <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[2] + "/*/*kernel_stats.csv")[0]
rows = [(r["Name"].split("(")[0].replace("void neb::", "").replace("neb::", ""), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f)) if "sun_table" in r["Name"]]
line = [l for l in open(sys.argv[2] + ".log") if l.startswith("{")]
st = json.loads(line[-1])["config"]["bvh"]["sun_table"] if line else {}
print("%-40s %s total %.2f ms | lit %s + %s, answered %s" % (sys.argv[1], " ".join(f"{n}:{v / 1e3:.2f}" for n, v in sorted(rows)), sum(v for _, v in rows) / 1e3,
                                                               st.get("lit_plus"), st.get("lit_minus"), st.get("rays_answered")))
PY
done
