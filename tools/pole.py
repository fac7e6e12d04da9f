"""Tuning aid: time the extension of ONE long tandem-array segment of the GRCh38-shaped workload.
The segment is given by its global start probe g0 (from the diagnostic build's "longest" line), the
pass (0 direct / 1 RC) and its length in probes; the script carves a sub-chunk around it and runs the
search on that sub-chunk only, under each option configuration given.
Usage: python tools/pole.py PASS G0 NPROBES ['grid3=64 long3=2048' ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

# option defaults (asgart_amd/csrc/index.hpp: struct Options); grid<t> = 0 means "default grid"
DEFAULTS = {"shard_lookback": 4096, "arms_kernel": 1, "kfilter_bits": 30, "long3": 16384, "cap1": 256,
            "test_cap_limit": -1, "test_genbits": 22, "tier_order": 3654217}


def opt_name(key):  # "ASGART_GRID3" or "grid3" -> "grid3"
    key = key.lower()
    return key[7:] if key.startswith("asgart_") else key

rc, g0, npr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
configs = sys.argv[4:] or [""]
k, step, M = 20, 10, 1000
pr = prep.prepare_records(synth.config_genome(4, 1.0))


def probes_in_chunk(L):
    if L < M or L < k + step or L - k - step == 0:
        return 0
    return (L - k - step + step - 1) // step


base = 0
for cs, cl in pr.chunks:
    n = probes_in_chunk(cl)
    if base <= g0 < base + n:
        break
    base += n
i0 = (g0 - base + 1) * step
span = npr * step
margin = 20000
if rc:
    lo = cl - i0 - span - k
else:
    lo = i0
lo = max(0, lo - margin)
hi = min(cl, lo + span + 2 * margin + k)
sub = [(cs + lo, hi - lo)]
print(f"chunk ({cs},{cl}) local probe {g0 - base}: sub-chunk {sub}", flush=True)
idx = asgart_amd.Index(pr.data, None)
idx.prepare(k)
st = asgart_amd.RunSettings.from_cli(reverse=bool(rc), complement=bool(rc))
ref = None
for conf in configs:
    kv = dict(x.split("=", 1) for x in conf.split()) if conf.strip() else {}
    for k_, v in kv.items():
        idx.set_option(opt_name(k_), int(v))
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        offs, sds = idx.search_duplications_raw(sub, st)
        dt = (time.perf_counter() - t0) * 1e3
        s = idx.stats(0)
        if best is None or s.ms_extend < best[1]:
            best = (dt, s.ms_extend, s.segments, s.heavy_segments, s.overflow_segments, len(sds))
    sig = hash((offs.tobytes(), sds.tobytes()))
    flag = "" if ref in (None, sig) else "  RESULT DIFFERS"
    ref = ref or sig
    print(f"[{conf or 'default'}] call {best[0]:.1f} ms extend {best[1]:.2f} ms segs {best[2]} heavy {best[3]} ovf {best[4]} sds {best[5]}{flag}", flush=True)
    for k_ in kv:
        idx.set_option(opt_name(k_), DEFAULTS.get(opt_name(k_), 0))
