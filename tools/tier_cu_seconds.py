"""Compute-unit time the extension tiers of a step hold, from the library's own tallies (option debug: every persistent
workgroup adds its lifetime to its tier's counter; placement tallies hit-probes and hits per tier).

    python tools/tier_cu_seconds.py [cfgK=cfg4] [--out FILE]

Runs the passes of the workload as one job with debug = 1 (in a child process, whose stderr carries the tallies), parses
the last step's lines and writes FILE (default gpurun_out/tier_cu_seconds_<cfg>.json; commit it as
profiles/rNN_<cfg>_tier_cu_seconds.json).  The step measured this way is ~25 ms slower than a plain one (two contended
atomics per segment in the placement walk): the tallies, not the step time, are what this is for."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
out_path = None
if "--out" in args:
    i = args.index("--out")
    out_path = args[i + 1]
    del args[i:i + 2]
wl = args[0] if args else "cfg4"
out_path = out_path or os.path.join(ROOT, "gpurun_out", f"tier_cu_seconds_{wl}.json")
env = dict(os.environ, TUNE_REPS="4")
p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tune_tiers.py"), wl, "--fused", "", "debug=1"], env=env,
                   capture_output=True, text=True)
if p.returncode != 0:
    sys.stderr.write(p.stderr[-4000:])
    sys.exit(p.returncode)
err = p.stderr.splitlines()


def last(prefix):
    for line in reversed(err):
        if prefix in line:
            return line
    return ""


import asgart_amd  # noqa: E402

res = {"workload": wl, "library_build": hashlib.sha256(open(asgart_amd.library_path(), "rb").read()).hexdigest()[:12],
       "method": "asgart_search_duplications_passes (both passes as one job), option debug = 1: per tier the sum of its workgroups' "
                 "lifetimes divided by the workgroups of that shape a compute unit holds (tiers 1..7: 11, 8, 1, 4, 2, 1, 1); the "
                 "runs over ranges of cut segments count with tier 3",
       "steps": [l.strip() for l in p.stdout.splitlines() if l.startswith("[")]}
m = last("compute-unit time held per tier")
cu = re.findall(r"(\d): (\d+) \((\d+)\)", m)
tot = re.search(r"total (\d+) = ([\d.]+) ms of the whole chip", m)
hp = re.findall(r"(\d): (\d+)K / ([\d.]+) / ([\d.]+)", last("per tier: hit-probes"))
lg = re.findall(r"(\d): ([\d.]+)", last("longest single segment per tier").split("(ms):")[-1])
tiers = {}
for t, ms, wgs in cu:
    tiers[t] = {"cu_ms": int(ms), "workgroups": int(wgs)}
for t, k_probes, hits, us in hp:
    tiers.setdefault(t, {}).update({"hit_probes": int(k_probes) * 1000, "hits_per_hit_probe": float(hits), "cu_us_per_hit_probe": float(us)})
for t, ms in lg:
    tiers.setdefault(t, {})["longest_segment_ms"] = float(ms)
res["tiers"] = tiers
if tot:
    res["total_cu_ms"] = int(tot.group(1))
    res["ms_of_the_whole_chip"] = float(tot.group(2))
res["placement"] = last("segments,").split("] ", 1)[-1]
res["ranges"] = last("cut into ranges").split("] ", 1)[-1]
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res["tiers"], indent=1))
print("wrote", out_path)
