"""Full-size check of the multi-GPU logic on ONE GPU: the families of shards 0..N-1 (run one after the other),
concatenated in rank order, must equal the unsharded result.  Usage: python tools/shard_check.py [N] [cfgK]"""
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
    for rc in (False, True):
        st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        parts, times = [], []
        for r in range(n_shards):
            t0 = time.perf_counter()
            o, s = idx.search_duplications_raw(pr.chunks, st, shard=r, n_shards=n_shards)
            times.append((time.perf_counter() - t0) * 1e3)
            parts.append((o, s))
        cat_sds = np.concatenate([s for _, s in parts]) if parts else sds[:0]
        cat_offs = [0]
        for o, _ in parts:
            cat_offs.extend((o[1:] + cat_offs[-1]).tolist())
        same = np.array_equal(np.array(cat_offs, dtype=np.uint64), offs) and np.array_equal(cat_sds, sds)
        print(f"rc={rc}: {len(offs) - 1} families, {len(sds)} SDs; {n_shards} shards identical: {same}; "
              f"per-shard call ms: {' '.join(f'{t:.0f}' for t in times)}", flush=True)
