"""Condense the rocprofv3 outputs of tools/profile_round.sh into small tracked files:
   <out>/profiles/<tag>_kernel_stats.csv      (copy of rocprofv3's kernel_stats)
   <out>/profiles/<tag>_pmc_fetch_write.json  (per kernel: launches, avg KB per launch, avg ns)
   <out>/profiles/pmc_traffic.json            (bytes per launch of the probe-search kernels; bench.py reads it)
Usage: python tools/summarize_prof.py TAG OUTDIR [WORKLOAD]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag, out = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg4"
dst = os.path.join(out, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "")


for f in glob.glob(os.path.join(out, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_{workload}_kernel_stats.csv"))

pmc = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    agg = defaultdict(lambda: [0, 0.0, 0])
    for f in glob.glob(os.path.join(out, f"{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            a = agg[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    pmc[ctr] = {k: {"launches": v[0], "sum_KB": v[1], "avg_KB_per_launch": v[1] / v[0], "avg_ns": v[2] / v[0]}
                for k, v in agg.items() if k.startswith("asgart::")}
json.dump(pmc, open(os.path.join(dst, f"{tag}_{workload}_pmc_fetch_write.json"), "w"), indent=1)


def per_launch(ctr, prefix):
    return sum(v["avg_KB_per_launch"] for k, v in pmc.get(ctr, {}).items() if k.startswith(prefix)) * 1024


probe = "asgart::probe_count_kernel"
big = "asgart::big_count_kernel"
search = sum(per_launch(c, p) for c in ("FETCH_SIZE", "WRITE_SIZE") for p in (probe, big))
if search > 0:
    json.dump({workload: int(search),
               "_note": "HBM-side bytes per launch of probe_count_kernel+big_count_kernel = (FETCH_SIZE+WRITE_SIZE) KB*1024 "
                        f"from separate rocprofv3 --pmc passes (profiles/{tag}_{workload}_pmc_fetch_write.json); narrow "
                        "(4-8 byte) gathers: not the calibrated 16 B/lane stream of MI355X_MICROARCH.md (HBM), so no x2 is "
                        "applied to FETCH_SIZE; Infinity-Cache hits are included in the counter"},
              open(os.path.join(dst, "pmc_traffic.json"), "w"))
print("summarised into", dst, "search bytes/launch", int(search))
