"""Per-kernel means of one rocprofv3 --pmc pass of SQ counters -> JSON (bench.py reads SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU).
usage: python tools/sq_from_pmc.py <pmc_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])):
    if "neb::" not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta[r["Kernel_Name"]] = {"VGPR_Count": float(r["VGPR_Count"]), "LDS_Block_Size": float(r["LDS_Block_Size"]),
                              "Grid_Size": float(r["Grid_Size"]), "Workgroup_Size": float(r["Workgroup_Size"])}
out = {"method": "rocprofv3 --pmc <SQ counters> (own pass), per-kernel means over the launches", "kernels": {}}
for k in sorted(acc):
    out["kernels"][k] = {c: sum(v) / len(v) for c, v in acc[k].items()}
    out["kernels"][k].update(meta[k])
    out["kernels"][k]["launches"] = len(next(iter(acc[k].values())))
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
