"""Per-kernel means of the memory-pipeline counter passes of tools/profile_round.sh -> JSON.
TA_BUSY_avr = cycles the texture-address units (the vector-memory issue path) were busy, averaged over the units; set against
the kernel's duration it says how close the kernel is to the rate at which a CU can issue vector loads.
usage: python tools/mem_from_pmc.py <pass1> <pass2> <pass3> <kernel_stats.csv> <out.json>"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:4]:
    for r in csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])):
        if "neb::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[4]))}
out = {"method": "rocprofv3 --pmc, three separate passes: {TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum}, {TCP_TCC_READ_REQ_sum "
                 "TCP_TOTAL_CACHE_ACCESSES_sum}, {TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum}; per-kernel means over the launches; "
                 "ta_busy_frac = TA_BUSY_avr / (kernel duration x 2.4 GHz peak clock)", "kernels": {}}
for k in sorted(acc):
    v = {c: sum(x) / len(x) for c, x in acc[k].items()}
    if k in dur and "TA_BUSY_avr" in v:
        v["duration_us"] = dur[k] / 1e3
        v["ta_busy_frac"] = v["TA_BUSY_avr"] / (dur[k] * 2.4)
    if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        v["l1_hit_frac"] = 1.0 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if v.get("TCC_REQ_sum"):
        v["l2_hit_frac"] = v["TCC_HIT_sum"] / v["TCC_REQ_sum"]
    out["kernels"][k] = v
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
