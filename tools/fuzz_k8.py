"""Randomised parity run of the long-segment kernels against the CPU oracle.

Every case is a small text made of tandem arrays (monomers of 3-400 bp, 20-400 copies, 0-10 % substitutions, one or
two arrays, optionally an interleaved second copy of the first) between random flanks, searched direct and -RC with a
random probe size / gap / minimum length, every segment with a multi-hit probe forced into tier 3 (K8: the kernel with
specialised waves and one barrier per probe), with a generation counter that wraps every few probes in some of the
cases.  Results must equal the oracle's bit for bit.

    python tools/fuzz_k8.py [cases=40] [first seed=0]        (FUZZ_SPLIT=len,warm,min: with the long segments cut into
                                                              ranges of len probes -- option split)
"""
import faulthandler
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd  # noqa: E402
import oracle  # noqa: E402  (the checker)


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = [rng.integers(0, 4, size=int(rng.integers(2_000, 40_000)))]
    n_arrays = int(rng.integers(1, 3))
    first = None
    for _ in range(n_arrays):
        mono = rng.integers(0, 4, size=int(rng.choice([3, 7, 12, 31, 64, 171, 400])))
        # (bounded so that the CPU oracle, whose cost goes with hits x arms per probe, answers in seconds: a 3-bp monomer
        # repeated 3 000 times under max_cardinality 5 000 kept it busy for more than ten minutes)
        copies = int(rng.integers(20, max(21, min(400 if len(mono) >= 31 else 40 * len(mono), 40_000 // len(mono)))))
        arr = np.tile(mono, copies)
        sub = float(rng.choice([0.0, 0.005, 0.02, 0.05, 0.1]))
        mut = rng.random(arr.shape) < sub
        arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
        parts.append(arr)
        first = arr if first is None else first
        parts.append(rng.integers(0, 4, size=int(rng.integers(500, 20_000))))
    if rng.random() < 0.5:  # a diverged second copy of the first array further on
        arr = first.copy()
        mut = rng.random(arr.shape) < 0.03
        arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
        parts.append(arr)
        parts.append(rng.integers(0, 4, size=int(rng.integers(500, 5_000))))
    g = np.concatenate(parts)
    text = np.concatenate([bases[g], np.frombuffer(b"$", dtype=np.uint8)])
    cli = dict(k=int(rng.choice([12, 16, 20, 24, 32])), gap=int(rng.choice([10, 50, 100, 300])),
               min_length=int(rng.choice([20, 100, 1000])), max_cardinality=int(rng.choice([50, 200, 500])))
    genbits = int(rng.choice([3, 4, 22]))
    return text, cli, genbits


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    t_all = time.time()
    for seed in range(seed0, seed0 + cases):
        faulthandler.dump_traceback_later(int(os.environ.get("FUZZ_STALL_S", "120")), exit=True)  # (a stuck case: where)
        text, cli, genbits = make_case(seed)
        chunks = [(0, len(text) - 1)]
        oidx = oracle.Index.build(text)
        with asgart_amd.Index(text, oidx.sa) as idx:
            idx.set_option("force_tier", 3)
            idx.set_option("test_genbits", genbits)
            if os.environ.get("FUZZ_SPLIT"):  # "len,warm,min": long segments as ranges (option split), sized for these cases
                ln, warm, mn = (int(v) for v in os.environ["FUZZ_SPLIT"].split(","))
                idx.set_option("split", 2)   # (ranges with 32- and 64-bit positions: ASGART_FORCE_WIDE=1 runs the latter)
                idx.set_option("split_len", ln)
                idx.set_option("split_warm", warm)
                idx.set_option("split_min", mn)
            else:
                idx.set_option("split", 0)
            for rc in (False, True):
                st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
                trace = os.environ.get("FUZZ_TRACE")
                if trace:
                    print(f"  seed {seed} {'RC' if rc else 'direct'}: device call ...", flush=True)
                offs, sds = idx.search_duplications_raw(chunks, st)
                if trace:
                    print("  ... oracle ...", flush=True)
                eo, es = oidx.run_raw(chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
                ok = np.array_equal(offs, eo) and np.array_equal(sds, es)
                s = idx.stats(0)
                print(f"seed {seed} n={len(text)} {cli} genbits={genbits} {'RC' if rc else 'direct'}: sds {len(sds)} "
                      f"heavy {s.heavy_segments} ovf {s.overflow_segments} cut {s.split_segments} refused {s.split_refused} "
                      f"{'ok' if ok else 'DIFFERS'}", flush=True)
                bad += 0 if ok else 1
    print(f"{cases} cases, {time.time() - t_all:.0f} s, {bad} mismatch(es)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
