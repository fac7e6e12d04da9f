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
        if not any(t in kn for t in ("extend_fast_kernel", "extend_k8_kernel")):
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
# per wave and hit-probe (16 waves per workgroup; the pole's 29 898 hit-probes x the three timed launches)
den = 3.0 * 29898.0 * 16.0
c = tot
if c.get("SQ_WAVE_CYCLES"):
    res["per_wave_and_hit_probe"] = {
        "instructions_valu": round(c.get("SQ_INSTS_VALU", 0) / den, 1), "instructions_salu": round(c.get("SQ_INSTS_SALU", 0) / den, 1),
        "instructions_lds": round(c.get("SQ_INSTS_LDS", 0) / den, 1), "wave_cycles": round(c["SQ_WAVE_CYCLES"] / den, 0),
        "branches": round(c.get("SQ_INSTS_BRANCH", 0) / den, 1)}
    res["share_of_wave_cycles"] = {
        "parked_at_waitcnt_or_barrier": round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
        "issue_stalls": round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
        "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
        "waiting_for_lds_issue": round(c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"], 4)}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        res["lds_bank_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
print(json.dumps(res, indent=1))
