"""The C oracle against the committed known-answer vectors (tests/golden/kat.json, produced by
the independent brute-force restatement) -- the pin of the oracle, since the reference has no
vectors of its own (PARITY UNPINNED, oracle/README.md)."""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "kat.json")) as fh:
    KAT = json.load(fh)


@pytest.mark.parametrize("case", KAT["cases"], ids=[c["name"] for c in KAT["cases"]])
def test_kat_families(case):
    text = case["text"].encode()
    s = case["settings"]
    chunks = [tuple(c) for c in case["chunks"]]
    assert oracle.find_chunks(text) == chunks
    if s.get("trim"):
        idx = oracle.Index.build_trim(text + b"$", *s["trim"])   # --trim START END, src/bin/asgart.rs:142-148
    else:
        idx = oracle.Index.build(text + b"$")
        assert oracle.sa_check(idx.text, idx.sa) == 0
    st = oracle.make_settings(k=s["k"], gap=s["gap"], min_length=s["min_length"],
                              max_cardinality=s["max_cardinality"], reverse=s["reverse"],
                              complement=s["complement"])
    got = idx.run(chunks, st)
    assert got == [[tuple(sd) for sd in fam] for fam in case["families"]]
    # thread count must not change the result (ordered collect, asgart.rs:240)
    assert idx.run(chunks, st, threads=4) == got


def test_d_ss_kat():
    for a, m, exp in KAT["d_ss"]:
        assert oracle.d_ss(tuple(a), tuple(m)) == exp


def test_probe_positions_kat():
    pp = KAT["probe_positions"]
    rng = np.random.default_rng(0)
    text = bytes(rng.choice(list(b"ACGT"), size=pp["L"]).astype(np.uint8)) + b"$"
    idx = oracle.Index.build(text)
    st = oracle.make_settings(k=pp["k"])
    status, offs, hits = idx.probe_hits(text[:-1], 0, st)
    assert len(status) == pp["count"]          # probes at i = 10, 20, ..., 970
    assert (len(status)) * (pp["k"] // 2) == pp["last"]


def test_tenth_threshold_is_integer_division():
    """(0.1 * len as f64) as i64 == len / 10 for every length the kernels can see
    (src/automaton.rs:69); the HIP kernel evaluates the same double expression."""
    lens = np.concatenate([np.arange(0, 2_000_000), np.random.default_rng(1).integers(0, 1 << 40, 1_000_000)])
    assert np.array_equal((0.1 * lens.astype(np.float64)).astype(np.int64), lens // 10)


def test_interval_form_of_the_arm_predicate():
    """The HIP kernels test `re - k < x < re + thr` instead of evaluating d_ss
    (pipeline_dev.hpp: arm_accepts).  Check it against the oracle's literal d_ss
    (src/automaton.rs:68-70,207-216) for every small configuration with len(right) >= k."""
    k = 8
    for rs in (0, 5, 40):
        for rlen in range(k, k + 30):
            re = rs + rlen
            for thr in (0, 1, 2, 7, 28, 100):
                for x in range(0, re + thr + 40):
                    literal = oracle.d_ss((rs, re), (x, x + k)) < thr and x + k > re
                    fast = thr > 0 and (re - k < x < re + thr)
                    assert literal == fast, (rs, re, thr, x)
