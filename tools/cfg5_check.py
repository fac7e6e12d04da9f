"""Full-size run of BASELINE.json configs[4] on one GPU: GRCh38-shaped genome + its diverged copy (two files,
6.1 Gb, 64-bit suffix array).  Builds the index on the GPU, verifies the suffix array with the GPU checker,
runs the direct and the RC pass, and checks size-independent properties of the result (the CPU oracle cannot
hold this input): every ProtoSD inside the text, arms at least min_length long, families non-empty, identical
output from a second call and from 3 shards concatenated.
Usage: python tools/cfg5_check.py [scale]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
t0 = time.time()
pr = prep.prepare_records(synth.config_genome(5, scale))
n = len(pr.data)
print(f"text {n} bytes, {len(pr.chunks)} chunks, {len(pr.map)} records, gen {time.time() - t0:.0f}s", flush=True)
t0 = time.time()
idx = asgart_amd.Index(pr.data, None)
print(f"upload + suffix array {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
bad = idx.check_sa()
print(f"GPU suffix-array verifier: {bad} violations ({time.time() - t0:.1f}s)", flush=True)
assert bad == 0
t0 = time.time()
idx.prepare(20)
print(f"keys + tables {time.time() - t0:.1f}s", flush=True)
for rc in (False, True):
    st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
    idx.search_duplications_raw(pr.chunks, st)  # builds the presence filter of this orientation
    t0 = time.time()
    offs, sds = idx.search_duplications_raw(pr.chunks, st)
    dt = time.time() - t0
    s = idx.stats(1)
    print(f"{'RC' if rc else 'direct'}: {len(offs) - 1} families, {len(sds)} ProtoSDs, call {dt * 1e3:.0f} ms "
          f"(search {s.ms_search:.0f} scan {s.ms_scan:.0f} fill {s.ms_fill:.0f} extend {s.ms_extend:.0f}), "
          f"{s.probes_total} probes, {s.filtered_hits} hits, filter rejected {s.probes_filter_rejected}", flush=True)
    assert len(sds) and int(offs[-1]) == len(sds) and np.all(np.diff(offs.astype(np.int64)) > 0)
    assert np.all(sds[:, 0] + sds[:, 2] <= n) and np.all(sds[:, 1] + sds[:, 3] <= n)
    assert np.all(sds[:, 3] >= 1000)
    parts = [idx.search_duplications_raw(pr.chunks, st, shard=r, n_shards=3) for r in range(3)]
    assert np.array_equal(np.concatenate([p[1] for p in parts]), sds), "shards != unsharded"
    cross = int(np.sum((sds[:, 0] < n // 2) != (sds[:, 1] < n // 2)))
    print(f"   3 shards concatenate to the same result; {cross} ProtoSDs pair the two genomes", flush=True)
idx.close()
print("ok")
