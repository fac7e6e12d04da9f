#!/usr/bin/env python3
"""Full-size oracle digests of the benchmarked configurations -> tests/golden/digests.json.

Runs the CPU oracle (oracle/asgart_oracle.c: SA-IS suffix array + literal restatement of
reference src/automaton.rs:57-216, src/searcher.rs:94-180, src/bin/asgart.rs:201-253) on the
synthetic stand-ins of BASELINE.json configs[2] (chr1-sized, `--skip-masked`) and configs[3]
(GRCh38-sized), direct and -RC pass, and records for every pass

    n_families, n_sds, sha256(fam_offsets as <u8), sha256(sds as (n,4) <u8),
    the oracle's probe/hit counters (probes searched, N-skipped, cardinality-skipped,
    probes with hits, raw hits, filtered hits)

plus sha256 of the suffix array (as little-endian u32) and of the prepared text.  The `-m gpu`
tests `test_cfg3_full_skip_masked_digest` / `test_cfg4_full_digest` run the HIP path on the
same seeded inputs and compare against these digests ("for each config:
raw families CPU == HIP").

Usage (CPU only; cfg4 needs ~35 GB of RAM and about an hour on 8 cores):
    python tests/golden/make_digests.py [cfg3s] [cfg4] [cfg5h] [cfg3r] [cfg4rq] [--threads N]
Existing entries of digests.json for configs not named on the command line are kept.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "digests.json")
MODES = (("direct", False, False), ("rc", True, True))
# name -> (synth config id, scale, skip_masked)
CASES = {
    "cfg3s": (3, 1.0, True),
    "cfg4": (4, 1.0, False),
    # BASELINE.json configs[4] (two files: genome + its diverged copy) at the smallest scale whose
    # text needs 64-bit suffix-array entries (n >= 2^32): the natively wide instantiations of every
    # kernel.  ~50 GB of RAM; the -RC pass is digested first (the direct pass of two near-identical
    # genomes is one serial chain per record and takes the oracle hours).
    "cfg5h": (5, 0.7, False),
    # chr1-sized, repeat-rich: young interspersed repeat families (synth.repeat_rich_genome) -- two probes in five
    # pass the presence filter instead of one in five; not a BASELINE.json config, a realism check of the bench
    "cfg3r": (6, 1.0, False),
    # the realism workload of the bench (cfg4r: GRCh38-sized, repeat-rich, higher-order satellite arrays) at a quarter
    # of its size: 770 Mb, the arrays at their full length (0.4-3 Mb each; the full-size input costs the oracle hours)
    "cfg4rq": (7, 0.25, False),
}
WIDE_N = (1 << 32) - 256  # texts from this size on are indexed with 64-bit slots


def sha_array(a: np.ndarray, dtype: str, slab: int = 1 << 26) -> str:
    """sha256 of the array's elements as little-endian `dtype`, hashed slab by slab."""
    h = hashlib.sha256()
    flat = a.reshape(-1)
    for off in range(0, len(flat), slab):
        h.update(np.ascontiguousarray(flat[off:off + slab]).astype(dtype, copy=False).tobytes())
    return h.hexdigest()


def digest_case(name: str, threads: int, save=lambda out: None) -> dict:
    cfg, scale, skip_masked = CASES[name]
    t0 = time.time()
    recs = synth.config_genome(cfg, scale)
    pr = prep.prepare_records(recs, skip_masked=skip_masked)
    del recs
    print(f"[{name}] genome {len(pr.data)} bytes, {len(pr.chunks)} chunks ({time.time() - t0:.0f}s)", flush=True)
    t0 = time.time()
    oidx = oracle.Index.build(pr.data)
    print(f"[{name}] SA-IS + searcher {time.time() - t0:.0f}s", flush=True)
    out = {
        "synth_config": cfg, "scale": scale, "skip_masked": skip_masked,
        "settings": {"k": 20, "gap": 100, "min_length": 1000, "max_cardinality": 500},
        "text_bytes": int(len(pr.data)), "chunks": len(pr.chunks),
        "text_sha256": sha_array(pr.data, "<u1"),
        "chunks_sha256": sha_array(np.array(pr.chunks, dtype=np.uint64), "<u8"),
        "numpy": np.__version__,
        "passes": {},
    }
    if len(pr.data) >= WIDE_N:
        out["sa_sha256_u64"] = sha_array(oidx.sa, "<u8")
    else:
        out["sa_sha256_u32"] = sha_array(oidx.sa, "<u4")
    save(out)
    for label, rev, comp in (MODES[::-1] if len(pr.data) >= WIDE_N else MODES):
        t0 = time.time()
        st = oracle.Stats()
        offs, sds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rev, complement=comp),
                                 threads=threads, stats=st)
        d = st.as_dict()
        out["passes"][label] = {
            "reverse": rev, "complement": comp,
            "n_families": int(len(offs) - 1), "n_sds": int(len(sds)),
            "fam_offsets_sha256": sha_array(offs, "<u8"),
            "sds_sha256": sha_array(sds, "<u8"),
            "counters": {key: d[key] for key in ("probes_total", "probes_n_skipped", "probes_searched",
                                                 "probes_card_skipped", "probes_with_hits", "raw_hits",
                                                 "filtered_hits")},
            "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": threads,
        }
        print(f"[{name}] {label}: {len(offs) - 1} families, {len(sds)} ProtoSDs ({time.time() - t0:.0f}s)", flush=True)
        save(out)
    oidx.close()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    threads = os.cpu_count() or 1
    if "--threads" in sys.argv:
        threads = int(sys.argv[sys.argv.index("--threads") + 1])
        args = [a for a in args if a != str(threads)]
    names = args or list(CASES)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    def save_as(name):
        def save(out):  # after the suffix array and after every pass: a long run leaves what it finished
            res[name] = out
            with open(OUT + ".tmp", "w") as fh:
                json.dump(res, fh, indent=1, sort_keys=True)
                fh.write("\n")
            os.replace(OUT + ".tmp", OUT)
        return save

    for name in names:
        digest_case(name, threads, save_as(name))


if __name__ == "__main__":
    main()
