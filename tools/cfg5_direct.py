"""cfg5 (two genomes, 6.2 Gb): index + two direct-pass calls; used under rocprofv3 --kernel-trace to see which
extension kernels carry the pass (tools/timeline.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

pr = prep.prepare_records(synth.config_genome(5, 1.0))
idx = asgart_amd.Index(pr.data, None)
idx.prepare(20)
st = asgart_amd.RunSettings.from_cli()
for rep in range(2):
    t0 = time.time()
    offs, sds = idx.search_duplications_raw(pr.chunks, st)
    s = idx.stats(0)
    print(f"direct call {rep}: {(time.time() - t0) * 1e3:.0f} ms, extend {s.ms_extend:.0f} (cascade {s.ms_extend_tier2:.0f}), "
          f"segments {s.segments}, heavy {s.heavy_segments}, overflow {s.overflow_segments}", flush=True)
idx.close()
