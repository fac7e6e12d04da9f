"""What every rank of an N-GPU run costs, measured on ONE GPU: the shards of a step (shard r = the r-th slice of every pass,
all passes as one job: what rank r of N runs) one after the other, each alone on the chip -- exactly what a rank's own
GPU sees, minus the gather of its result lists.  The shards' families merged by key must equal the unsharded result.

    python tools/shard_check.py [N[,N...]=1,2,4,8] [cfgK=cfg4] [--out FILE] [--reps 3] [--settle 5] [--only R] [option=value ...]

Writes FILE (default gpurun_out/shards_<cfg>.json; commit it as profiles/rNN_<cfg>_shards.json): per N the per-shard
call times (median of --reps steady-state calls, ms) with the library's phase times, max_r = what an N-GPU step takes
(bench.py's scaling model reads it), and whether the merged result equals the unsharded one.  No 1 -> N curve is
measured by this: one GPU, one shard at a time."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

args = [a for a in sys.argv[1:]]
out_path, reps, settle = None, 3, 5
if "--out" in args:
    i = args.index("--out")
    out_path = args[i + 1]
    del args[i:i + 2]
if "--reps" in args:
    i = args.index("--reps")
    reps = int(args[i + 1])
    del args[i:i + 2]
if "--settle" in args:
    i = args.index("--settle")
    settle = int(args[i + 1])
    del args[i:i + 2]
only = None   # (--only R: just shard R of every N, no merge check -- for a look at one shard with option debug)
if "--only" in args:
    i = args.index("--only")
    only = int(args[i + 1])
    del args[i:i + 2]
opts = [a for a in args if "=" in a]
args = [a for a in args if "=" not in a and a != "--fused"]
ns = [int(x) for x in (args[0] if args else "1,2,4,8").split(",")]
wl = args[1] if len(args) > 1 else "cfg4"
WORK = {"cfg1": (1, 1.0, False, ((False, False),)), "cfg2": (2, 1.0, False, ((False, False), (True, True))),
        "cfg3": (3, 1.0, True, ((False, False), (True, True))), "cfg4": (4, 1.0, False, ((False, False), (True, True))),
        "cfg5": (5, 1.0, False, ((False, False), (True, True)))}
cfg, scale, skip_masked, modes = WORK[wl]
out_path = out_path or os.path.join(ROOT, "gpurun_out", f"shards_{wl}.json")
pr = prep.prepare_records(synth.config_genome(cfg, scale), skip_masked=skip_masked)
total_bp = sum(l for _, l in pr.chunks)
PHASES = ("ms_search", "ms_scan", "ms_fill", "ms_extend", "ms_longest_segment", "ms_total")
res = {"workload": wl, "bp_per_pass": total_bp, "passes": len(modes), "reps": reps, "settling_calls": settle,
       "library_build": hashlib.sha256(open(asgart_amd.library_path(), "rb").read()).hexdigest()[:12],
       "method": "one GPU; shard r of N = the r-th slice of every pass as ONE job (asgart_search_duplications_passes_shard), "
                 "the shards one after the other, each alone on the chip; per shard the median wall time of the call over "
                 "`reps` steady-state calls (results on the host), phases from asgart_stats",
       "n": {}}
with asgart_amd.Index(pr.data, None) as idx:
    idx.prepare(20)
    for a in opts:
        k_, v_ = a.split("=")
        idx.set_option(k_, int(v_))
    sts = [asgart_amd.RunSettings.from_cli(reverse=r, complement=c, skip_masked=skip_masked) for r, c in modes]
    call = (lambda r, n: idx.search_duplications_passes(pr.chunks, sts, shard=r, n_shards=n, with_keys=True)) if len(sts) > 1 else \
        (lambda r, n: [idx.search_duplications_raw(pr.chunks, sts[0], shard=r, n_shards=n, with_keys=True)])
    for _ in range(2):   # (second use: presence filters, position-sorted lists; cuts that held)
        whole = call(0, 1)
    for n in ns:
        shards, parts = [], []
        for r in (range(n) if only is None else [only]):
            # (the first sharded call of a shape plans its cuts, and a segment with a cut that did not hold gets twice the
            # warm-up in the next one, up to split_warm_max: steady state after at most five calls with the defaults)
            first_ms = []
            for _ in range(settle):
                t0 = time.perf_counter()
                call(r, n)
                first_ms.append(round((time.perf_counter() - t0) * 1e3, 1))
            times, stats = [], None
            for _ in range(reps):
                t0 = time.perf_counter()
                part = call(r, n)
                times.append((time.perf_counter() - t0) * 1e3)
                stats = idx.stats().as_dict()
            parts.append(part)
            shards.append({"ms": round(float(np.median(times)), 3), "ms_all": [round(t, 3) for t in times], "ms_settling_calls": first_ms,
                           "passes_as_one_job": stats["passes"] == len(sts),
                           **{ph[3:]: round(stats[ph], 3) for ph in PHASES},
                           "front": round(stats["ms_search"] + stats["ms_scan"] + stats["ms_fill"], 3),
                           "segments": stats["segments"], "split_segments": stats["split_segments"],
                           "split_refused": stats["split_refused"], "proto_sds": stats["proto_sds"]})
        same = True
        for j in range(len(sts) if only is None else 0):
            mo, ms = asgart_amd.merge_shards([p_[j] for p_ in parts])
            same = same and np.array_equal(mo, whole[j][0]) and np.array_equal(ms, whole[j][1])
        mx = max(s_["ms"] for s_ in shards)
        res["n"][str(n)] = {"max_ms": mx, "mean_ms": round(sum(s_["ms"] for s_ in shards) / len(shards), 3),
                            "mbp_per_s_if_ranks_ran_side_by_side": round(total_bp * len(sts) / mx / 1e3, 1),
                            "merged_equals_unsharded": bool(same), "shards": shards}
        print(f"N={n}: per-shard ms {' '.join('%.1f' % s_['ms'] for s_ in shards)}; max {mx:.1f}, identical after the merge: {same}",
              flush=True)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as fh:
    json.dump(res, fh, indent=1)
print("wrote", out_path)
