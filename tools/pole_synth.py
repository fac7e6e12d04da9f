"""Tuning aid: the critical path of the extension phase in isolation.

Builds a small text whose only structure is ONE long tandem satellite array (171-bp monomer x N copies,
mutated), i.e. the kind of segment that sets the critical path of a GRCh38-shaped pass: tens of thousands of
strictly serial probes with hundreds of live arms and ~100 hits each.  Runs the search on it (direct and RC
pass) under each option configuration and prints the extension time; results are checked for equality
across configurations (and against the CPU oracle with --check).

Usage: python tools/pole_synth.py [--copies 3800] [--sub 0.03] [--homolog BP] [--check] ['force_tier=3' 'force_tier=6' ...]
--homolog BP: instead of the array, a random region of BP bases and its 1.2 %-diverged copy further on (a chromosome against
its homologue, in small: ONE segment of BP / 10 sparse probes with one or two long-lived arms; a short repeat family is
sprinkled over both so that some probes have a dozen hits)
Set ASGART_LIB=asgart_amd/libasgart_hip_diag.so for the per-phase cycle breakdown (stderr).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402

DEFAULTS = {"shard_lookback": 4096, "arms_kernel": 1, "kfilter_bits": 30, "long3": 16384, "cap1": 256,
            "test_cap_limit": -1, "test_genbits": 22, "tier_order": 3654217, "cap6_pct": 140, "cap3_pct": 160, "posbits": 1,
            "cap45_pct": 100, "solo": 1, "cap6w_pct": 160, "dense3": 16, "dense6": 32, "fuse_passes": 1, "barren": 2, "split": 1,
            "split_len": 0, "split_warm": 6144, "split_min": 0, "split_runs": 224}


def make_text(copies, sub, seed=5, flank=400_000):
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    mono = rng.integers(0, 4, size=171)
    arr = np.tile(mono, copies)
    mut = rng.random(arr.shape) < sub
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    g = np.concatenate([rng.integers(0, 4, size=flank), arr, rng.integers(0, 4, size=flank)])
    return np.concatenate([bases[g], np.frombuffer(b"$", dtype=np.uint8)])


def make_homolog(bp, seed=5, flank=200_000):
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = rng.integers(0, 4, size=bp)
    rep = rng.integers(0, 4, size=300)
    n_rep = 0 if os.environ.get("HOMOLOG_NO_REPEATS") else max(1, bp // 50_000)
    for at in rng.integers(0, bp - 300, size=n_rep):   # a young repeat family: bursts of hits
        a[at:at + 300] = rep
    b = a.copy()
    mut = rng.random(b.shape) < 0.012
    b[mut] = (b[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    g = np.concatenate([rng.integers(0, 4, size=flank), a, rng.integers(0, 4, size=50_000), b, rng.integers(0, 4, size=flank)])
    return np.concatenate([bases[g], np.frombuffer(b"$", dtype=np.uint8)])


def main():
    args = sys.argv[1:]
    copies, sub, check = 3800, 0.03, False
    homolog = 0
    confs = []
    while args:
        a = args.pop(0)
        if a == "--copies":
            copies = int(args.pop(0))
        elif a == "--sub":
            sub = float(args.pop(0))
        elif a == "--homolog":
            homolog = int(args.pop(0))
        elif a == "--check":
            check = True
        else:
            confs.append(a)
    confs = confs or [""]
    text = make_homolog(homolog) if homolog else make_text(copies, sub)
    chunks = [(0, len(text) - 1)]
    idx = asgart_amd.Index(text, None)
    idx.prepare(20)
    oidx = None
    if check:
        import oracle
        oidx = oracle.Index.build(text, idx.sa_read(0, len(text)))
    for rc in (False, True):
        st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
        ref = None
        if oidx is not None:
            import oracle
            eo, es = oidx.run_raw(chunks, oracle.make_settings(reverse=rc, complement=rc))
            ref = hash((eo.tobytes(), es.tobytes()))
        for conf in confs:
            kv = dict(x.split("=", 1) for x in conf.split()) if conf.strip() else {}
            for k_, v in kv.items():
                idx.set_option(k_, int(v))
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                offs, sds = idx.search_duplications_raw(chunks, st)
                dt = (time.perf_counter() - t0) * 1e3
                s = idx.stats(0)
                if best is None or s.ms_extend < best[1]:
                    best = (dt, s.ms_extend, s.segments, s.heavy_segments, s.overflow_segments, len(sds),
                            s.probes_with_hits, s.filtered_hits)
            sig = hash((offs.tobytes(), sds.tobytes()))
            flag = "" if ref in (None, sig) else "  RESULT DIFFERS"
            ref = ref if ref is not None else sig
            print(f"[{'RC' if rc else 'direct'} {conf or 'default'}] call {best[0]:.1f} ms extend {best[1]:.2f} ms "
                  f"segs {best[2]} heavy {best[3]} ovf {best[4]} sds {best[5]} hit-probes {best[6]} hits {best[7]} "
                  f"({best[7] / max(1, best[6]):.1f}/probe){flag}", flush=True)
            for k_ in kv:
                idx.set_option(k_, DEFAULTS.get(k_, 0))
    idx.close()


if __name__ == "__main__":
    main()
