"""Tuning aid: wall time of the index build alone (upload + GPU suffix array, keys / tables / position-sorted lists,
presence filters of the direct and the -RC orientation).  Usage: python tools/index_build.py [cfgK] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

cfg = int((sys.argv[1] if len(sys.argv) > 1 else "cfg4")[3:])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pr = prep.prepare_records(synth.config_genome(cfg, 1.0))
print(f"text {len(pr.data)} bytes", flush=True)
for rep in range(reps):
    t0 = time.perf_counter()
    idx = asgart_amd.Index(pr.data, None)
    t1 = time.perf_counter()
    idx.prepare(20)
    t2 = time.perf_counter()
    small = [(pr.chunks[0][0], min(pr.chunks[0][1], 200_000))]
    for rc in (False, True):
        idx.search_duplications_raw(small, asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc))
    t3 = time.perf_counter()
    print(f"rep {rep}: upload + suffix array {t1 - t0:.2f} s, keys / tables / lists {t2 - t1:.2f} s, "
          f"two presence filters {t3 - t2:.2f} s, total {t3 - t0:.2f} s", flush=True)
    idx.close()
