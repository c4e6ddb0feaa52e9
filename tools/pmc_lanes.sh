#!/bin/bash
# evidence: instruction mix and VALU lane use of the frame's kernels from hardware counters (own --pmc passes, never with a trace domain):
#   SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) = the share of lanes live in the VALU instructions issued -- what tools/gi_wave_stamps.py derives from the
#   walk's own wave stamps.   usage (GPU box): bash tools/pmc_lanes.sh <tag>   -> gpurun_out/<tag>_lanes/profiles/<tag>_sq_mix.json (copy it to profiles/)
case " $* " in *" --gpus "*) echo "$0 refuses --gpus"; exit 2;; esac
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/${tag}_lanes
rm -rf "$out" && mkdir -p "$out"
k=0
for c in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
  k=$((k + 1))
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$out/p$k" -- python bench.py --steps 4 --warmup 2 --cpu-frames 0 --overlap off > "$out/p$k.log" 2>&1 || { tail -5 "$out/p$k.log"; exit 1; }
  python tools/sq_from_pmc.py "$out/p$k" "$out/p$k.json" > /dev/null || exit 1
done
mkdir -p "$out/profiles"; python - "$out" "$out/profiles/${tag}_sq_mix.json" <<'PY'
import json, sys
merged = {}
for k in (1, 2, 3):
    for name, v in json.load(open(f"{sys.argv[1]}/p{k}.json"))["kernels"].items():
        merged.setdefault(name, {}).update(v)
out = {"method": "rocprofv3 --pmc, three separate passes of SQ counters over `python bench.py --steps 4 --warmup 2 --overlap off`; per-kernel means over the launches; "
                 "valu_lane_share = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)", "kernels": {}}
for name, v in merged.items():
    if not any(s in name for s in ("gi_raygen_trace", "gi_shade", "gi_shadow_list", "svgf_atrous_lds")):
        continue
    if v.get("SQ_ACTIVE_INST_VALU"):
        v["valu_lane_share"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (v["SQ_ACTIVE_INST_VALU"] * 64.0)
    out["kernels"][name] = v
    w = v.get("SQ_WAVES") or 1
    print(name.split("(")[0][-60:], {c: round(v[c] / 1e6, 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM") if c in v},
          "M wave-instructions; VALU lane share", round(v.get("valu_lane_share", 0.0), 3), "; LDS bank-conflict cycles / LDS active", round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_ACTIVE_INST_LDS", 1.0), 1.0), 3))
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
