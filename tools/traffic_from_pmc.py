"""Turns two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of `python bench.py ...` into per-kernel HBM traffic.

Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB of fabric requests; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE reads exactly.
(Check on this workload: the temporal kernel gives 124.5 MB read / 45.6 MB written = its algorithmic 60 / 22 B per pixel.)
usage: python tools/traffic_from_pmc.py <fetch_dir> <write_dir> <out.json> <width> <height>
"""
import collections
import csv
import glob
import json
import sys


def means(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
out = {"width": int(sys.argv[4]), "height": int(sys.argv[5]), "unit": "bytes per launch",
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024",
       "kernels": {}}
for k in sorted(fetch):
    if "neb::" in k:
        out["kernels"][k] = {"read": 2 * fetch[k] * 1024, "written": write.get(k, 0.0) * 1024,
                             "total": 2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
