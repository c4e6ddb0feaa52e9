#!/bin/bash
# One GPU-box call that produces a round's evidence under gpurun_out/<tag>/ and the reduced summaries under profiles/<tag>_*:
#   bench line, rocprofv3 --kernel-trace --stats, and three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ counters).
#   The profiled runs keep ONE frame in flight (--overlap off): a kernel's duration and counters are then its own, not those of a kernel that ran beside
#   the next frame's closest-hit walk (the bench line itself is the default run; its per-kernel times come from serial, event-timed frames anyway).
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02a [extra bench.py flags]
case " $* $AB_BENCH_FLAGS " in *" --gpus "*) echo "$0 refuses --gpus: under rocprofv3 bench.py would start its ranks from a process the profiler has given a GPU (profile each rank's own command instead)"; exit 2;; esac
set -o pipefail
tag=$1
[ -n "$tag" ] || { echo "usage: $0 <tag> [bench.py flags]"; exit 2; }
shift
extra="$@"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
rm -rf "$out" && mkdir -p "$out" profiles
python bench.py $extra > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
echo "[profile_round] bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python bench.py --steps 32 --warmup 8 --cpu-frames 0 --overlap off $extra > "$out/stats.log" 2>&1 || { tail -20 "$out/stats.log"; exit 1; }
echo "[profile_round] kernel trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python bench.py --steps 4 --warmup 2 --cpu-frames 0 --overlap off $extra > "$out/pmc_$c.log" 2>&1 || { tail -20 "$out/pmc_$c.log"; exit 1; }
  echo "[profile_round] pmc $c done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$out/pmc_sq" -- python bench.py --steps 4 --warmup 2 --cpu-frames 0 --overlap off $extra > "$out/pmc_sq.log" 2>&1 || { tail -20 "$out/pmc_sq.log"; exit 1; }
echo "[profile_round] pmc SQ done"
# memory-pipeline counters (own passes): texture-address unit busy cycles, L1 (TCP) accesses / misses to L2, L2 (TCC) hits / misses
k=0
for c in "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  k=$((k + 1))
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_mem$k" -- python bench.py --steps 4 --warmup 2 --cpu-frames 0 --overlap off $extra > "$out/pmc_mem$k.log" 2>&1 || { tail -20 "$out/pmc_mem$k.log"; exit 1; }
done
echo "[profile_round] pmc memory pipeline done"
cp "$out/bench.json" "profiles/${tag}_bench.json"
cp $(ls "$out"/stats/*/*kernel_stats.csv | head -1) "profiles/${tag}_kernel_stats.csv"
python tools/traffic_from_pmc.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "profiles/${tag}_hbm_traffic.json" 1920 1080 > /dev/null
python tools/sq_from_pmc.py "$out/pmc_sq" "profiles/${tag}_sq_counters.json" > /dev/null
python tools/mem_from_pmc.py "$out/pmc_mem1" "$out/pmc_mem2" "$out/pmc_mem3" "profiles/${tag}_kernel_stats.csv" "profiles/${tag}_mem_counters.json" > /dev/null
# stamp the library build the counters were taken on into every summary (bench.py quotes a summary only for its own build)
python - "$tag" <<'PY'
import glob, json, sys
tag = sys.argv[1]
bid = json.load(open(f"profiles/{tag}_bench.json"))["config"]["library_build_id"]
for f in glob.glob(f"profiles/{tag}_*.json"):
    d = json.load(open(f))
    if isinstance(d, dict) and "build_id" not in d:
        d["build_id"] = bid
        json.dump(d, open(f, "w"), indent=1)
PY
# the bench line once more, now that this build's counter summaries exist: it quotes them (roofline.traffic / .valu)
python bench.py $extra > "$out/bench2.json" 2>> "$out/bench.err" && cp "$out/bench2.json" "profiles/${tag}_bench.json"
mkdir -p "$out/profiles" && cp profiles/${tag}_* "$out/profiles/"
python - "$tag" <<'PY'
import csv, sys
tag = sys.argv[1]
rows = sorted(csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{float(r["AverageNs"]) / 1e3:9.1f} us x{r["Calls"]:>5}  {r["Name"][:110]}')
PY
