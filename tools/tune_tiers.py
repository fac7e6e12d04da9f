"""Tuning aid: build the GRCh38-shaped index once, then time the search call under a list of
option configurations (launch order / grid sizes of the extension tiers are read per call).
Usage: python tools/tune_tiers.py [cfg] 'A=1 B=2' 'A=3' ...   (one quoted config per argument)"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

# option defaults (asgart_amd/csrc/index.hpp: struct Options); grid<t> = 0 means "default grid"
DEFAULTS = {"shard_lookback": 4096, "arms_kernel": 1, "kfilter_bits": 30, "long3": 16384, "cap1": 256,
            "test_cap_limit": -1, "test_genbits": 22, "tier_order": 3654217, "cap6_pct": 140, "cap3_pct": 160, "posbits": 1,
            "cap45_pct": 100, "solo": 1, "cap6w_pct": 160, "dense3": 16, "dense6": 32, "fuse_passes": 1, "barren": 2, "split": 1,
            "split_len": 0, "split_warm": 6144, "split_min": 0, "split_runs": 224}


def opt_name(key):  # "ASGART_GRID3" or "grid3" -> "grid3"
    key = key.lower()
    return key[7:] if key.startswith("asgart_") else key

args = sys.argv[1:]
PIPELINED = "--pipelined" in args   # time whole steps (RC pass first, direct pass issued at its progress signal)
FUSED = "--fused" in args           # time whole steps through the passes call (both passes as one job)
args = [a for a in args if a not in ("--pipelined", "--fused")]
cfg, scale = 4, 1.0
if args and args[0].startswith("cfg"):
    cfg = int(args[0][3:]); args = args[1:]
configs = args or [""]
recs = synth.config_genome(cfg, scale)
pr = prep.prepare_records(recs)
idx = asgart_amd.Index(pr.data, None)
idx.prepare(20)
settings = [asgart_amd.RunSettings.from_cli(reverse=r, complement=r) for r in (False, True)]
ref = None
for conf in configs:
    kv = dict(x.split("=", 1) for x in conf.split()) if conf.strip() else {}
    for k_, v in kv.items():
        idx.set_option(opt_name(k_), int(v))
    line = []
    sig = hashlib.sha1()
    if PIPELINED:
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)
        times = []
        for rep in range(int(os.environ.get("TUNE_REPS", "5"))):
            t0 = time.perf_counter()
            prog = np.zeros(len(pr.chunks), dtype=np.uint64)
            fut = pool.submit(idx.search_duplications_raw, pr.chunks, settings[1], 0, 1, prog)
            while not fut.done() and not prog.any():
                time.sleep(0.0005)
            r0 = idx.search_duplications_raw(pr.chunks, settings[0])
            r1 = fut.result()
            times.append((time.perf_counter() - t0) * 1e3)
        for r in (r0, r1):
            sig.update(r[0].tobytes()); sig.update(r[1].tobytes())
        line.append("step min %.1f median %.1f ms" % (min(times[1:]), sorted(times[1:])[len(times[1:]) // 2]))
    if FUSED:
        times = []
        for rep in range(int(os.environ.get("TUNE_REPS", "5"))):
            t0 = time.perf_counter()
            r0, r1 = idx.search_duplications_passes(pr.chunks, settings)
            times.append((time.perf_counter() - t0) * 1e3)
        s = idx.stats(0)
        for r in (r0, r1):
            sig.update(r[0].tobytes()); sig.update(r[1].tobytes())
        line.append("step min %.1f median %.1f ms | front %.1f extend %.1f longest tier %.1f ovf %d" % (
            min(times[1:]), sorted(times[1:])[len(times[1:]) // 2], s.ms_search + s.ms_scan + s.ms_fill, s.ms_extend,
            s.ms_longest_tier, s.overflow_segments))
    for st in ([] if PIPELINED or FUSED else settings):
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            dt = (time.perf_counter() - t0) * 1e3
            s = idx.stats(0)
            if best is None or dt < best[0]:
                best = (dt, s.ms_extend, s.ms_search, s.overflow_segments)
        sig.update(offs.tobytes()); sig.update(sds.tobytes())
        line.append("call %.1f ms extend %.1f search %.1f ovf %d" % best)
    ok = ""
    if ref is None:
        ref = sig.hexdigest()
    elif ref != sig.hexdigest():
        ok = "  RESULT DIFFERS"
    print(f"[{conf or 'default'}] " + " | ".join(line) + ok, flush=True)
    for k_ in kv:
        idx.set_option(opt_name(k_), DEFAULTS.get(opt_name(k_), 0))
