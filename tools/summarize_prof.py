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


for f in glob.glob(os.path.join(out, f"{tag}_{workload}_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_{workload}_kernel_stats.csv"))

pmc = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    agg = defaultdict(lambda: [0, 0.0, 0])
    for f in glob.glob(os.path.join(out, f"{tag}_{workload}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            a = agg[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    pmc[ctr] = {k: {"launches": v[0], "sum_KB": v[1], "avg_KB_per_launch": v[1] / v[0], "avg_ns": v[2] / v[0]}
                for k, v in agg.items() if k.startswith("asgart::")}
pj = os.path.join(dst, f"{tag}_{workload}_pmc_fetch_write.json")
if any(pmc.values()):
    json.dump(pmc, open(pj, "w"), indent=1)
elif os.path.exists(pj):          # raw traces already deleted: re-summarise from the per-kernel file
    pmc = json.load(open(pj))


def per_launch(ctr, prefix, field="avg_KB_per_launch"):
    return sum(v[field] for k, v in pmc.get(ctr, {}).items() if k.startswith(prefix))


# the product kernels only: <..., true> are the byte-accounting variants bench.py runs once per pass
probe = "asgart::probe_count_kernel<unsigned int, false>"
big = "asgart::big_count_kernel<unsigned int, false>"
rank = "asgart::rank_count_kernel<unsigned int, false>"   # (absent from older profiles: contributes 0 then)
coll = "asgart::collect_pending_kernel"                   # (round 4: builds the two work lists from the marks)
if not any(k.startswith(probe) for k in pmc.get("FETCH_SIZE", {})):   # 64-bit index
    probe, big, rank = (x.replace("unsigned int", "unsigned long") for x in (probe, big, rank))
group = (probe, coll, big, rank)
search = sum(per_launch(c, p) for c in ("FETCH_SIZE", "WRITE_SIZE") for p in group) * 1024
# the kernels' own durations in the (serialising) PMC passes, averaged over the two passes
pmc_ms = sum(per_launch(c, p, "avg_ns") for c in ("FETCH_SIZE", "WRITE_SIZE") for p in group) / 2 / 1e6
stats_ms = None
sfile = os.path.join(dst, f"{tag}_{workload}_kernel_stats.csv")
if os.path.exists(sfile):
    tot = 0.0
    for row in csv.DictReader(open(sfile)):
        nm = short(row["Name"])
        if any(nm.startswith(g_) for g_ in group):  # (short() keeps the template arguments)
            tot += float(row["AverageNs"])
    stats_ms = tot / 1e6
if search > 0:
    tfile = os.path.join(dst, "pmc_traffic.json")
    allw = {}
    # keep the other workloads' entries (repo copy first, then this run's scratch copy)
    for prev in (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"), tfile):
        if os.path.exists(prev):
            try:
                allw.update({k: v for k, v in json.load(open(prev)).items() if isinstance(v, dict)})
            except Exception:
                pass
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get("ASGART_LIB") or os.path.join(root, "asgart_amd", "libasgart_hip.so")
    build = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:12] if os.path.exists(lib) else None
    ppl = 1
    try:   # passes per launch of the profiled runs (the passes call fuses the passes of a step into one job)
        sb = open(os.path.join(out, f"{tag}_{workload}_stats_bench.json")).read().strip().splitlines()[-1]
        ppl = int(json.loads(sb)["roofline"].get("passes_per_launch", 1))
    except Exception:
        pass
    allw[workload] = {
        "build": build,
        "passes_per_launch": ppl,   # sha256[:12] of the library the passes ran (bench.py compares it with the one it loads)
        "traffic_bytes_per_launch": int(search),
        "kernel_ms_per_launch": round(pmc_ms, 4),
        "stats_kernel_ms_per_launch": None if stats_ms is None else round(stats_ms, 4),
        "source": f"profiles/{tag}_{workload}_pmc_fetch_write.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                  f"passes, kernels serialised) and profiles/{tag}_{workload}_kernel_stats.csv (--kernel-trace --stats of "
                  "the default bench run)",
    }
    allw["_note"] = ("per workload: HBM-side bytes per launch (= passes_per_launch passes: the passes call runs the passes of a step as one job) of probe_count_kernel + collect_pending_kernel + big_count_kernel + rank_count_kernel = "
                     "(FETCH_SIZE + WRITE_SIZE) KB * 1024; kernel_ms_per_launch = the two kernels' durations in those "
                     "PMC passes; stats_kernel_ms_per_launch = their rocprofv3 --stats averages in the un-instrumented "
                     "bench run (passes overlapped).  Narrow 4-8 byte gathers: FETCH_SIZE is used as reported (no x2); bench.py adds half the bytes of the "
                     "wide coalesced loads (text windows, filter bitmaps), which gfx950 counts at one half; "
                     "tools/ubench_gather.hip calibrates bytes per random gather; Infinity-Cache hits are included in "
                     "the counter")
    json.dump(allw, open(tfile, "w"), indent=1)
print("summarised into", dst, "search bytes/launch", int(search), "pmc ms", round(pmc_ms, 3), "stats ms", stats_ms)
