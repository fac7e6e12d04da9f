"""Timing aid for --compute-score: search a synthetic genome (direct pass), run the host steps, then the GPU
ComputeScore over all surviving duplications; a sample of them is also scored by the CPU oracle.
Lives under tests/ because it loads the CPU oracle (test infrastructure) for the comparison leg.
Usage: python tests/score_bench.py [cfg3|cfg2]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
import oracle  # noqa: E402  (CPU comparison only)
from asgart_amd import postprocess, prep, synth  # noqa: E402

cfg = int((sys.argv[1] if len(sys.argv) > 1 else "cfg3")[3:])
pr = prep.prepare_records(synth.config_genome(cfg, 1.0))
strand = asgart_amd.Strand("synthetic", pr.data, pr.map)
st = asgart_amd.RunSettings.from_cli()
with asgart_amd.Index(pr.data, None) as idx:
    fams = asgart_amd.SearchDuplications(pr.chunks, None, st, index=idx).run([], strand)
    t0 = time.perf_counter()
    for step in (postprocess.FilterNs(), postprocess.ReOrder(), postprocess.ReduceOverlap()):
        fams = step.run(fams, strand)
    t_host = time.perf_counter() - t0
    flat = [sd for f in fams for sd in f]
    arr = np.array([sd.as_tuple() for sd in flat], dtype=np.uint64).reshape(-1, 4)
    cells = float(((arr[:, 2] + 1).astype(np.float64) * (arr[:, 3] + 1)).sum())
    idx.compute_scores(arr[:8])  # warm-up
    t0 = time.perf_counter()
    ident = idx.compute_scores(arr)
    t_gpu = time.perf_counter() - t0
print(f"{len(flat)} duplications after FilterNs/ReOrder/ReduceOverlap ({t_host:.2f} s host), longest arm "
      f"{int(arr[:, 2:4].max())} bp, {cells:.3e} DP cells")
print(f"GPU ComputeScore: {t_gpu:.3f} s = {cells / t_gpu / 1e9:.1f} G cells/s; identity min/median/max "
      f"{ident.min():.2f} / {np.median(ident):.2f} / {ident.max():.2f}")
order = np.argsort(arr[:, 2] * arr[:, 3])
sample = order[np.linspace(0, len(order) - 1, 24).astype(int)]
t0 = time.perf_counter()
want = np.array([oracle.levenshtein_identity(pr.data, arr[j], False, False) for j in sample], dtype=np.float32)
t_cpu = time.perf_counter() - t0
sc = float(((arr[sample, 2] + 1).astype(np.float64) * (arr[sample, 3] + 1)).sum())
print(f"CPU oracle on {len(sample)} of them: {t_cpu:.2f} s = {sc / t_cpu / 1e9:.2f} G cells/s (1 thread); "
      f"equal: {bool(np.array_equal(want, ident[sample]))}")
