#!/usr/bin/env python3
"""Regenerates tests/golden/kat.json.

The reference (Rust, no tests, not buildable here) provides no vectors, so these
known-answer cases are produced by the *brute-force Python restatement*
(tests/bruteforce.py), independently of the C oracle, and cross-checked by
tests/test_oracle_golden.py against the C oracle and (on the GPU) the HIP path.
Inputs follow SURVEY.md section 8c: U = 12 000 bases from the 64-bit LCG with seed 1,
X = U[2003:3503]; every case is a small edit of U with a copy of X.  The expected
tuples are (left, right, left_length, right_length) after the left fix-up.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bruteforce as B  # noqa: E402


def cases():
    U = B.lcg_bases(1, 12000)
    X = U[2003:3503]
    sub = {ord("A"): ord("C"), ord("C"): ord("G"), ord("G"): ord("T"), ord("T"): ord("A")}
    Xc = bytes(sub[c] if j % 97 == 50 else c for j, c in enumerate(X))
    yield "A_direct", U[:8000] + X + U[9500:], dict()
    yield "A_rc", U[:8000] + X + U[9500:], dict(reverse=True, complement=True)
    yield "B_direct", U[:8000] + B.revcomp(X) + U[9500:], dict()
    yield "B_rc", U[:8000] + B.revcomp(X) + U[9500:], dict(reverse=True, complement=True)
    yield "B_r_only", U[:8000] + X[::-1] + U[9500:], dict(reverse=True)
    yield "B_c_only", U[:8000] + B.complemented(X) + U[9500:], dict(complement=True)
    yield "C_substitutions", U[:8000] + Xc + U[9500:], dict()
    yield "D_deletion60", U[:8000] + X[:700] + X[760:] + U[8000 + 1440:], dict()
    yield "E_deletion200", U[:8000] + X[:700] + X[900:] + U[8000 + 1300:], dict(min_len=500)
    yield "F_three_copies", U[:5000] + X + U[6500:8000] + X + U[9500:], dict()
    # live arms at the end of the needle are dropped (automaton.rs:201-203)
    yield "G_copy_at_end", U[:9000] + X, dict()
    # the left copy ends its chunk: the arms are alive at the end of the needle, so nothing is reported
    yield "G_live_at_chunk_end", U[:3503] + b"N" * 6000 + U[3503:8000] + X + U[9500:], dict()
    yield "G_live_at_chunk_end_control", U[:3503] + B.lcg_bases(99, 200) + b"N" * 6000 + U[3503:8000] + X + U[9500:], dict()
    # N-start probes are skipped without ageing (automaton.rs:100-102)
    yield "H_n_inside", U[:8000] + X[:600] + b"N" * 90 + X[690:] + U[9500:], dict()
    # cardinality skip (automaton.rs:115-117)
    yield "I_cardinality", (U[:2000] + X) * 4 + U[3600:5000], dict(max_card=2)
    # --trim START END (src/bin/asgart.rs:142-148): the whole input is searched against the suffixes of
    # data[start..end] only.  F has copies of X at 2003, 5000 and 8000:
    yield "T_trim_right_copy", U[:5000] + X + U[6500:8000] + X + U[9500:], dict(trim=(7500, 10000))
    yield "T_trim_middle_copy", U[:5000] + X + U[6500:8000] + X + U[9500:], dict(trim=(4000, 7000))
    yield "T_trim_cuts_a_copy", U[:5000] + X + U[6500:8000] + X + U[9500:], dict(trim=(8500, 20000))
    yield "T_trim_rc", U[:8000] + B.revcomp(X) + U[9500:], dict(trim=(7000, 11000), reverse=True, complement=True)


def main():
    out = {"generator": "x=(x*6364136223846793005+1442695040888963407) mod 2^64; base='ACGT'[x>>62]",
           "cases": []}
    for name, text, kw in cases():
        strand = text + b"$"
        trim = kw.get("trim")
        if trim is not None:   # the checks of prepare_data, src/bin/asgart.rs:432-463
            trim = (trim[0], min(trim[1], len(strand) - 1))
        sa = B.trim_suffix_array(strand, *trim) if trim else B.suffix_array(strand)
        chunks = B.find_chunks(text)
        fams = B.run(strand, sa, chunks, k=20, gap=100, min_len=kw.get("min_len", 1000),
                     max_card=kw.get("max_card", 500), reverse=kw.get("reverse", False),
                     complement=kw.get("complement", False))
        out["cases"].append({
            "name": name, "text": text.decode(), "chunks": chunks,
            "settings": {"k": 20, "gap": 100, "min_length": kw.get("min_len", 1000),
                         "max_cardinality": kw.get("max_card", 500),
                         "reverse": kw.get("reverse", False), "complement": kw.get("complement", False),
                         "trim": list(trim) if trim else None},
            "families": [[list(sd) for sd in fam] for fam in fams],
        })
        print(name, fams)
    out["d_ss"] = [[[100, 200], [201, 221], 1], [[100, 200], [200, 220], 0], [[100, 200], [50, 70], 30],
                   [[100, 200], [90, 110], 0], [[100, 200], [300, 320], 100], [[100, 200], [0, 20], 80]]
    out["probe_positions"] = {"L": 1000, "k": 20, "first": 10, "last": 970, "count": 97}
    with open(os.path.join(HERE, "kat.json"), "w") as fh:
        json.dump(out, fh, indent=0)


if __name__ == "__main__":
    main()
