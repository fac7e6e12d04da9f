"""Full-size check of the multi-GPU logic on ONE GPU: the families of shards 0..N-1 (run one after the other),
merged by their keys, must equal the unsharded result; prints every shard's call time and max / mean (what an
N-GPU pass would take / its balance).  Usage: python tools/shard_check.py [N] [cfgK] [shard_lpt=0|1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

n_shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = int((sys.argv[2] if len(sys.argv) > 2 else "cfg4")[3:])
pr = prep.prepare_records(synth.config_genome(cfg, 1.0))
with asgart_amd.Index(pr.data, None) as idx:
    idx.prepare(20)
    for a in sys.argv[3:]:
        k_, v_ = a.split("=")
        idx.set_option(k_, int(v_))
    for rc in (False, True):
        st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        parts, times = [], []
        for r in range(n_shards):
            t0 = time.perf_counter()
            part = idx.search_duplications_raw(pr.chunks, st, shard=r, n_shards=n_shards, with_keys=True)
            times.append((time.perf_counter() - t0) * 1e3)
            parts.append(part)
        mo, ms = asgart_amd.merge_shards(parts)
        same = np.array_equal(mo, offs) and np.array_equal(ms, sds)
        print(f"rc={rc}: {len(offs) - 1} families, {len(sds)} SDs; {n_shards} shards identical: {same}; "
              f"per-shard call ms: {' '.join(f'{t:.0f}' for t in times)}; max {max(times):.0f} mean {sum(times) / len(times):.0f}",
              flush=True)
