"""Sums the rocprofv3 --pmc counter rows of the arm kernel's long launches (tools/pole_pmc.sh)."""
import csv
import glob
import json
import sys

out_dir, conf = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
tot, names, launches = {}, set(), 0
for f in glob.glob(out_dir + "/g*/**/*counter_collection.csv", recursive=True):
    per_dispatch = {}
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if not any(t in kn for t in ("extend_fast_kernel", "extend_arms_kernel", "extend_k7_kernel")):
            continue
        key = row.get("Dispatch_Id")
        per_dispatch.setdefault(key, {"k": kn})[row["Counter_Name"]] = float(row["Counter_Value"])
    # the long launches only (the pole; the flanks give tiny launches): top counter value per group
    for d in per_dispatch.values():
        names.add(d["k"])
        for c, v in d.items():
            if c != "k":
                tot[c] = tot.get(c, 0.0) + v
probes = 29898 * 2  # hit-probes of the pole x the launches that carry it (best of 3 repeats => 3 launches; see note)
res = {"conf": conf, "kernels": sorted(names), "counters_sum_over_launches": tot,
       "note": "tools/pole_synth.py runs the direct pass three times (best of 3) and an empty -RC pass; divide by 3 x 29 898 "
               "hit-probes x waves per workgroup for per-wave, per-probe figures"}
print(json.dumps(res, indent=1))
