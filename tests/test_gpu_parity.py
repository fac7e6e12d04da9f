"""GPU parity tests: the HIP path (through the C ABI) must be bit-exact against the
CPU oracle on the same seeded inputs.  Run with `pytest -m gpu` on an MI355X."""
import os
import sys
import time
import numpy as np
import pytest

import oracle
import asgart_amd
from asgart_amd import prep, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODES = [(False, False), (True, False), (False, True), (True, True)]
_ORACLE_CACHE = {}


def _small_genome(seed=7, lens=(180_000, 120_000), **kw):
    args = dict(sd_per_mb=40, sd_len=(1000, 8000), alu_frac=0.05, l1_frac=0.0, sat_per_record=0)
    args.update(kw)
    return synth.make_genome(list(lens), seed=seed, **args)


@pytest.fixture(scope="module")
def case(hiplib):
    pr = prep.prepare_records(_small_genome())
    oidx = oracle.Index.build(pr.data)
    idx = asgart_amd.Index(pr.data, oidx.sa)
    yield pr, oidx, idx
    idx.close()


def test_searcher_cache_matches_oracle(case):
    """Searcher::new entries (reference src/searcher.rs:99-143)."""
    pr, oidx, idx = case
    rng = np.random.default_rng(1)
    pats = [bytes(rng.choice(list(b"ATGCN"), size=8, p=[.24, .24, .24, .24, .04]).astype(np.uint8))
            for _ in range(2000)]
    pats += [bytes(pr.data[p:p + 8]) for p in rng.integers(0, len(pr.data) - 9, size=2000)]
    got = asgart_amd.Searcher(idx).cache(pats)
    for p, (lo, hi) in zip(pats, got):
        elo, ehi = oidx.cache_get(p)
        if ehi > elo:
            assert (lo, hi) == (elo, ehi), p
        else:
            assert hi == lo, p


def test_searcher_search_matches_oracle(case):
    """Searcher::search (reference src/searcher.rs:145-180): same hits, same SA order."""
    pr, oidx, idx = case
    rng = np.random.default_rng(2)
    text = pr.data
    pats = [bytes(text[p:p + 20]) for p in rng.integers(0, len(text) - 21, size=3000)]
    pats += [bytes(rng.choice(list(b"ACGT"), size=20).astype(np.uint8)) for _ in range(500)]
    s = asgart_amd.Searcher(idx)
    ranges = s.search_ranges(pats)
    for p, (lo, hi) in zip(pats, ranges):
        exp, (elo, ehi) = oidx.search(p)
        assert hi - lo == len(exp), p
        if len(exp):
            assert (lo, hi) == (elo, ehi)
            assert np.array_equal(idx.sa_read(lo, hi).astype(np.uint64), exp)


@pytest.mark.parametrize("reverse,complement", MODES)
def test_probe_hits_csr_matches_oracle(case, reverse, complement):
    """Per-probe filtered hit lists (reference src/automaton.rs:96-117)."""
    pr, oidx, idx = case
    st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement)
    ost = oracle.make_settings(reverse=reverse, complement=complement)
    status, offs, hits = idx.probe_hits(pr.chunks, st)
    e_status, e_offs, e_hits = [], [0], []
    for ch in pr.chunks:
        nd = oracle.prepare_needle(pr.data, ch, ost)
        s1, o1, h1 = oidx.probe_hits(nd, ch[0], ost)
        e_status.append(s1)
        e_offs.extend((o1[1:] + e_offs[-1]).tolist())
        e_hits.append(h1)
    assert np.array_equal(status, np.concatenate(e_status))
    assert np.array_equal(offs, np.array(e_offs, dtype=np.uint64))
    assert np.array_equal(hits, np.concatenate(e_hits))


@pytest.mark.parametrize("reverse,complement", MODES)
def test_families_match_oracle(case, reverse, complement):
    """SearchDuplications::run body (reference src/bin/asgart.rs:201-253)."""
    pr, oidx, idx = case
    st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement)
    offs, sds = idx.search_duplications_raw(pr.chunks, st)
    eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=reverse, complement=complement))
    assert np.array_equal(offs, eoffs)
    assert np.array_equal(sds, esds)
    if not reverse and not complement:
        assert len(offs) > 1  # the planted duplications are found


# ---- a battery of harder shapes: every case compares families (and the CSR for one
# orientation) of the HIP path with the oracle ------------------------------------
BATTERY = {
    # dense low-divergence repeat family: hundreds of hits per probe, > 64 live arms
    "dense_repeats": dict(lens=(150_000,), seed=11, gen=dict(alu_frac=0.35, alu_div=(0.005, 0.03),
                                                             sd_per_mb=10, sd_len=(1000, 5000)),
                          cli=dict(min_length=150)),
    # cardinality skips (max_cardinality lowered) + repeats
    "card_skip": dict(lens=(200_000,), seed=12, gen=dict(alu_frac=0.3, alu_div=(0.01, 0.05)),
                      cli=dict(max_cardinality=40)),
    # tandem satellite arrays
    "satellites": dict(lens=(400_000,), seed=13, gen=dict(alu_frac=0.02, sat_per_record=3,
                                                          sat_copies=(50, 400))),
    # long diverged duplications: many retired pieces per family
    "long_sds": dict(lens=(600_000, 300_000), seed=14, gen=dict(alu_frac=0.0, sd_per_mb=12,
                                                                sd_len=(20_000, 80_000))),
    # many tiny families
    "short_min_len": dict(lens=(250_000,), seed=15, gen=dict(alu_frac=0.1), cli=dict(min_length=100)),
    "k12": dict(lens=(120_000,), seed=16, gen=dict(alu_frac=0.05), cli=dict(k=12, gap=50)),
    "k21_odd": dict(lens=(200_000,), seed=17, gen=dict(alu_frac=0.05), cli=dict(k=21, gap=33)),
    "k9_gap0": dict(lens=(60_000,), seed=18, gen=dict(alu_frac=0.02), cli=dict(k=9, gap=0, min_length=200)),
    # probes longer than one 63-bit key word (21 bases): key word + tail compared through the text
    "k22": dict(lens=(200_000,), seed=21, gen=dict(alu_frac=0.1, alu_div=(0.0, 0.02)), cli=dict(k=22, gap=60)),
    "k31_odd": dict(lens=(250_000,), seed=22, gen=dict(alu_frac=0.15, alu_div=(0.0, 0.01), sat_per_record=2,
                                                       sat_copies=(50, 300)), cli=dict(k=31, gap=100)),
    "k42": dict(lens=(300_000, 100_000), seed=23, gen=dict(alu_frac=0.2, alu_div=(0.0, 0.01)),
                cli=dict(k=42, gap=200, min_length=500)),
    "masked": dict(lens=(300_000, 200_000), seed=19, gen=dict(alu_frac=0.2), skip_masked=True),
}


def _battery_case(name):
    spec = BATTERY[name]
    recs = _small_genome(seed=spec["seed"], lens=spec["lens"], **spec.get("gen", {}))
    pr = prep.prepare_records(recs, skip_masked=spec.get("skip_masked", False))
    return pr, spec.get("cli", {})


@pytest.mark.parametrize("name", sorted(BATTERY))
def test_battery_families_and_csr(hiplib, name, monkeypatch):
    pr, cli = _battery_case(name)
    oidx = oracle.Index.build(pr.data)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for reverse, complement in MODES:
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
            ost = oracle.make_settings(reverse=reverse, complement=complement, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, ost, threads=4)
            assert np.array_equal(offs, eoffs), (name, reverse, complement, len(offs), len(eoffs))
            assert np.array_equal(sds, esds), (name, reverse, complement)
        st = asgart_amd.RunSettings.from_cli(reverse=True, complement=False, **cli)
        ost = oracle.make_settings(reverse=True, complement=False, **cli)
        status, offs, hits = idx.probe_hits(pr.chunks, st)
        e_status, e_offs, e_hits = [], [0], []
        for ch in pr.chunks:
            s1, o1, h1 = oidx.probe_hits(oracle.prepare_needle(pr.data, ch, ost), ch[0], ost)
            e_status.append(s1)
            e_offs.extend((o1[1:] + e_offs[-1]).tolist())
            e_hits.append(h1)
        assert np.array_equal(status, np.concatenate(e_status))
        assert np.array_equal(offs, np.array(e_offs, dtype=np.uint64))
        assert np.array_equal(hits, np.concatenate(e_hits))


def test_golden_kat_on_gpu(hiplib):
    """The committed known-answer vectors (tests/golden/kat.json) through the HIP path."""
    import json
    import os

    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat.json")))
    for case in kat["cases"]:
        s = case["settings"]
        text = case["text"].encode() + b"$"
        trim = tuple(s["trim"]) if s.get("trim") else None    # --trim cases: sub-range array built on the GPU
        sa = None if trim else oracle.divsufsort64(text)
        st = asgart_amd.RunSettings.from_cli(k=s["k"], gap=s["gap"], min_length=s["min_length"],
                                             max_cardinality=s["max_cardinality"],
                                             reverse=s["reverse"], complement=s["complement"])
        strand = asgart_amd.Strand("kat", np.frombuffer(text, dtype=np.uint8))
        step = asgart_amd.SearchDuplications([tuple(c) for c in case["chunks"]], trim, st, suffix_array=sa)
        got = [[sd.as_tuple() for sd in fam] for fam in step.run([], strand)]
        assert got == [[tuple(sd) for sd in fam] for fam in case["families"]], case["name"]


def test_tail_corner_on_gpu(hiplib):
    """Probes sharing an 8-mer with the last (< k) bases of the text take the exact-bisection
    path (reference src/searcher.rs:164-170); same answers as the oracle."""
    import random

    rng = random.Random(5)
    core = bytes(rng.choice(b"ACGT") for _ in range(1500))
    motif = b"ACGTTGCAAC"
    for fill in (b"A", b"T", b"G"):
        text = core[:400] + motif + fill * 10 + core[400:900] + motif + b"T" * 10 + core[900:] + motif
        strand = text + b"$"
        oidx = oracle.Index.build(strand)
        with asgart_amd.Index(strand, oidx.sa) as idx:
            s = asgart_amd.Searcher(idx)
            pats = [motif + fill * 10, motif + b"T" * 10, motif + b"A" * 10, motif + b"C" * 10,
                    motif + b"G" * 10]
            for p, (lo, hi) in zip(pats, s.search_ranges(pats)):
                exp, _ = oidx.search(p)
                assert np.array_equal(idx.sa_read(lo, hi).astype(np.uint64), exp), p
            for r, c in MODES:
                st = asgart_amd.RunSettings.from_cli(min_length=100, reverse=r, complement=c)
                offs, sds = idx.search_duplications_raw([(0, len(text))], st)
                eo, es = oidx.run_raw([(0, len(text))], oracle.make_settings(min_length=100, reverse=r, complement=c))
                assert np.array_equal(offs, eo) and np.array_equal(sds, es)


@pytest.mark.parametrize("k", [22, 29, 42, 43, 50, 64, 101, 128])
def test_long_probes_search_and_tail_corner(hiplib, k, monkeypatch):
    """Probe sizes 22..128 (the reference takes any k >= 8, src/bin/asgart.rs:564-631; up to 42 the tail of a probe is a
    second packed word, beyond it is compared base by base through the text): Searcher::search against
    the oracle for patterns from the text, random patterns, patterns that differ from a text k-mer only in their
    last bases (same key word, different tail) and the text-tail corner; then whole runs, also with 64-bit slots."""
    import random

    rng = random.Random(k)
    core = bytes(rng.choice(b"ACGT") for _ in range(3000))
    rep = bytes(rng.choice(b"ACGT") for _ in range(300))
    motif = b"ACGTTGCAAC"
    text = core[:700] + rep + motif + b"A" * 40 + core[700:1500] + rep[:150] + b"G" + rep[151:] + core[1500:] + rep + motif
    strand = text + b"$"
    oidx = oracle.Index.build(strand)
    for wide in (0, 1):
        monkeypatch.setenv("ASGART_FORCE_WIDE", str(wide))
        with asgart_amd.Index(strand, oidx.sa) as idx:
            s = asgart_amd.Searcher(idx)
            pats = [text[p:p + k] for p in range(0, len(text) - k, 7)]
            pats += [bytes(rng.choice(b"ACGT") for _ in range(k)) for _ in range(300)]
            for p0 in range(690, 1050, 5):   # same first 21 bases as a text k-mer, another tail
                t = bytearray(text[p0:p0 + k])
                t[rng.randrange(21, k)] = rng.choice(b"ACGT")
                pats.append(bytes(t))
            if k > 42:
                for p0 in range(690, 1050, 5):   # same first 42 bases (both words), another base further on
                    t = bytearray(text[p0:p0 + k])
                    t[rng.randrange(42, k)] = rng.choice(b"ACGT")
                    pats.append(bytes(t))
            pats += [motif + b"A" * (k - 10), motif + b"T" * (k - 10), (motif + rep)[:k], (motif + b"A" * 40 + b"C" * k)[:k]]
            for p, (lo, hi) in zip(pats, s.search_ranges(pats)):
                exp, (elo, ehi) = oidx.search(p)
                assert hi - lo == len(exp), p
                if len(exp):
                    assert np.array_equal(idx.sa_read(lo, hi).astype(np.uint64), exp), p
            for r, c in MODES:
                st = asgart_amd.RunSettings.from_cli(k=k, gap=50, min_length=60, reverse=r, complement=c)
                offs, sds = idx.search_duplications_raw([(0, len(text))], st)
                eo, es = oidx.run_raw([(0, len(text))], oracle.make_settings(k=k, gap=50, min_length=60, reverse=r, complement=c))
                assert np.array_equal(offs, eo) and np.array_equal(sds, es), (k, r, c)
                assert len(es) > 0 or r or c


def test_errors_on_gpu(hiplib):
    with pytest.raises(asgart_amd.AsgartError) as e:
        asgart_amd.Index(b"ACGTXACGT$")
    assert e.value.code == -1
    text = b"ACGT" * 100 + b"$"
    with asgart_amd.Index(text, oracle.divsufsort64(text)) as idx:
        with pytest.raises(asgart_amd.AsgartError):
            idx.search_duplications_raw([(0, 401)], asgart_amd.RunSettings.from_cli())     # covers '$'
        with pytest.raises(asgart_amd.AsgartError):
            idx.search_duplications_raw([(0, 400)], asgart_amd.RunSettings.from_cli(k=129))  # k > 128
        with pytest.raises(asgart_amd.AsgartError):
            idx.search_duplications_raw([(0, 400)], asgart_amd.RunSettings.from_cli(k=7))   # k < 8 (searcher.rs:95-97)
        offs, sds = idx.search_duplications_raw([], asgart_amd.RunSettings.from_cli())
        assert len(offs) == 1 and len(sds) == 0
        offs, sds = idx.search_duplications_raw([(0, 25)], asgart_amd.RunSettings.from_cli(min_length=10))
        assert len(offs) == 1


def test_gpu_suffix_array_matches_oracle(hiplib):
    """asgart_sa_build64 == divsufsort64 contract (reference src/divsufsort.rs:10): the unique
    suffix array, checked against the oracle's SA-IS and its O(n) verifier."""
    rng = np.random.default_rng(3)
    texts = []
    pr = prep.prepare_records(_small_genome(seed=21, lens=(200_000, 90_000), alu_frac=0.2,
                                            sat_per_record=2, sat_copies=(50, 300)))
    texts.append(pr.data)
    texts.append(np.frombuffer(b"A" * 5000 + b"$", dtype=np.uint8))
    texts.append(np.frombuffer(b"ACG" * 3000 + b"N" * 7000 + b"ACGT" * 100 + b"$", dtype=np.uint8))
    texts.append(np.frombuffer(b"N" * 30000 + b"$", dtype=np.uint8))
    texts.append(rng.integers(0, 256, size=50_000, dtype=np.uint8))       # arbitrary bytes
    texts.append(rng.integers(0, 2, size=70_000, dtype=np.uint8))         # binary, long repeats
    texts.append(np.frombuffer(b"\x00" * 100 + b"\xff" * 100 + b"\x00" * 50, dtype=np.uint8))
    texts.append(np.frombuffer(b"G", dtype=np.uint8))
    for t in texts:
        t = np.ascontiguousarray(t)
        sa = asgart_amd.sa_build64(t)
        assert oracle.sa_check(t, sa) == 0
        assert np.array_equal(sa, oracle.divsufsort64(t))


def test_index_builds_its_own_suffix_array(case):
    pr, oidx, idx = case
    with asgart_amd.Index(pr.data, None) as idx2:
        assert np.array_equal(idx2.sa_read(0, len(pr.data)), oidx.sa)
        st = asgart_amd.RunSettings.from_cli()
        a = idx.search_duplications_raw(pr.chunks, st)
        b = idx2.search_duplications_raw(pr.chunks, st)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("name", ["long_sds", "dense_repeats", "satellites", "masked"])
@pytest.mark.parametrize("halo", ["default", "tiny"])
def test_shards_concatenate_to_unsharded(hiplib, name, halo, monkeypatch):
    """Multi-GPU logic on one GPU: shards 0..R-1, run one after the other, merged by their family keys, must
    give exactly the unsharded result (segments are never split; no exchange): contiguous slices of the probe
    sequence with halos -- `tiny` halos force the look-back / look-ahead retry paths."""
    if halo == "tiny":
        monkeypatch.setenv("ASGART_SHARD_LOOKBACK", "2")
        monkeypatch.setenv("ASGART_SHARD_LOOKAHEAD", "3")
    pr, cli = _battery_case(name)
    sa = oracle.divsufsort64(pr.data)
    with asgart_amd.Index(pr.data, sa) as idx:
        for reverse, complement in ((False, False), (True, True)):
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            for R in (2, 3, 8, 61):
                parts = [idx.search_duplications_raw(pr.chunks, st, shard=r, n_shards=R, with_keys=True) for r in range(R)]
                got_offs, got_sds = asgart_amd.merge_shards(parts)
                # contiguous slices: plain concatenation in shard order is the result as well
                assert np.array_equal(np.concatenate([p[1] for p in parts]), sds)
                assert np.array_equal(got_offs, offs), (name, R, reverse)
                assert np.array_equal(got_sds, sds), (name, R, reverse)


@pytest.mark.parametrize("cap", [24, 100])
def test_overflow_cascade_gives_identical_results(hiplib, cap, monkeypatch):
    """Segments are normally placed in a tier that provably fits them; shrinking the tiers'
    capacity forces the overflow cascade (re-run in the next tier) -- same results."""
    pr, cli = _battery_case("dense_repeats")
    oidx = oracle.Index.build(pr.data)
    monkeypatch.setenv("ASGART_TEST_CAP_LIMIT", str(cap))
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for reverse, complement in ((False, False), (True, True)):
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            assert idx.stats().overflow_segments > 0
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=reverse, complement=complement, **cli), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (cap, reverse)


@pytest.mark.parametrize("tier", [2, 3, 4, 5, 6, 7, "2-lds", "4-lds", "6-lds"])
@pytest.mark.parametrize("name", ["dense_repeats", "satellites", "long_sds"])
def test_escalation_tiers_give_identical_results(hiplib, name, tier, monkeypatch):
    """Segments that do not fit the one-wave kernel go to larger tiers: the arm-resident kernels in five
    shapes (2: one wave, 3: the kernel with specialised waves for long dense segments, 4/5/6: 256/512/1024 threads by
    capacity), the HBM-scratch kernel (7), or -- "N-lds", what max_cardinality > 1024 selects -- the LDS-array workgroup
    kernels in tiers 2, 4 and 6.  Forcing every segment with a multi-hit probe into tier `tier` must not
    change a single ProtoSD."""
    pr, cli = _battery_case(name)
    oidx = oracle.Index.build(pr.data)
    if isinstance(tier, str):
        tier, variant = tier.split("-")
        tier = int(tier)
        if variant == "lds":
            monkeypatch.setenv("ASGART_ARMS_KERNEL", "0")
    monkeypatch.setenv("ASGART_FORCE_TIER", str(tier))
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for reverse, complement in ((False, False), (True, True)):
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            if name == "dense_repeats":
                assert idx.stats().heavy_segments > 0 or tier < 3
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=reverse, complement=complement, **cli), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (name, tier, reverse)


@pytest.mark.parametrize("card", [600, 1024, 3000])
@pytest.mark.parametrize("force", [0, 2, 4, 6])
def test_large_max_cardinality_tier_sets(hiplib, card, force, monkeypatch):
    """max_cardinality decides which kernels exist: up to 512 every arm-resident shape; up to 1024 only the
    shapes that stage 1024 hits per probe (tiers 3, 5, 6 -- tiers 2 and 4 are skipped, a forced tier moves on
    to the next one that exists); above, the LDS-array kernels.  Results match the oracle in every set."""
    pr, cli = _battery_case("dense_repeats")
    cli = dict(cli, max_cardinality=card)
    if force:
        monkeypatch.setenv("ASGART_FORCE_TIER", str(force))
    oidx = oracle.Index.build(pr.data)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), rc


def test_cfg1_ecoli_sized_direct_bit_exact(hiplib):
    """BASELINE.json configs[0]: E. coli MG1655-sized synthetic (4.6 Mb, one record), direct duplications,
    k=20 g=100 -- the reference's own CPU-runnable case.  One chunk, so the reference's outer par_iter has no
    parallelism (src/bin/asgart.rs:201-205); the HIP path (suffix array built on the GPU) must give the
    oracle's families bit for bit, with the oracle's own SA-IS suffix array as well."""
    pr = prep.prepare_records(synth.config_genome(1))
    assert len(pr.data) == 4_641_652 + 1
    osa = oracle.divsufsort64(pr.data)
    with asgart_amd.Index(pr.data, None) as idx:
        assert np.array_equal(idx.sa_read(0, len(pr.data)).astype(np.int64), osa)
        oidx = oracle.Index.build(pr.data, osa)
        st = asgart_amd.RunSettings.from_cli()
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        ost = oracle.Stats()
        eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(), threads=4, stats=ost)
        assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds)
        assert len(sds) > 0
        got, want = idx.stats().as_dict(), ost.as_dict()
        for key in ("probes_total", "probes_n_skipped", "probes_searched", "probes_card_skipped",
                    "probes_with_hits", "raw_hits", "filtered_hits"):
            assert got[key] == want[key], key


def test_cfg2_yeast_sized_direct_and_rc_bit_exact(hiplib):
    """BASELINE.json configs[1]: S. cerevisiae-sized synthetic (12.2 Mb, 17 records), direct + RC
    on one MI355X, suffix array built on the GPU, bit-exact against the CPU oracle; also as
    4 shards (multi-GPU logic)."""
    pr = prep.prepare_records(synth.config_genome(2))
    with asgart_amd.Index(pr.data, None) as idx:
        sa = idx.sa_read(0, len(pr.data))
        oidx = oracle.Index.build(pr.data, sa)
        assert oracle.sa_check(pr.data, sa) == 0
        for reverse, complement in ((False, False), (True, True)):
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=reverse, complement=complement),
                                       threads=16)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds)
            assert len(sds) > 100
            parts = [idx.search_duplications_raw(pr.chunks, st, shard=r, n_shards=4, with_keys=True) for r in range(4)]
            mo, ms = asgart_amd.merge_shards(parts)
            assert np.array_equal(mo, offs) and np.array_equal(ms, sds)


def test_chr1_sized_sample_parity(hiplib):
    """A 40 Mb slice of the chr1-shaped synthetic (satellite arrays, dense repeat family,
    cardinality skips): the HIP path over all chunks vs the oracle."""
    recs = synth.make_genome([40_000_000], seed=synth.SEED_BASE + 3)
    pr = prep.prepare_records(recs)
    with asgart_amd.Index(pr.data, None) as idx:
        sa = idx.sa_read(0, len(pr.data))
        oidx = oracle.Index.build(pr.data, sa)
        st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True)
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=True, complement=True), threads=16)
        assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds)
        assert len(sds) > 0 and idx.stats().heavy_segments >= 0


def _sha_slabs(arrays, dtype):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).astype(dtype, copy=False).tobytes())
    return h.hexdigest()


def _check_against_oracle_digest(name):
    """Full-size parity of a benchmarked configuration: the HIP path on the seeded synthetic input
    against tests/golden/digests.json, which tests/golden/make_digests.py wrote from the CPU oracle
    (its own SA-IS suffix array + the literal restatement of the reference's search path): input and
    suffix array by sha256, then per pass the probe/hit counters, family and ProtoSD counts and the
    sha256 of the complete result arrays."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "digests.json")) as fh:
        d = json.load(fh)[name]
    recs = synth.config_genome(d["synth_config"], d["scale"])
    pr = prep.prepare_records(recs, skip_masked=d["skip_masked"])
    del recs
    n = len(pr.data)
    assert n == d["text_bytes"] and len(pr.chunks) == d["chunks"]
    slab = 1 << 26
    assert _sha_slabs((pr.data[o:o + slab] for o in range(0, n, slab)), "<u1") == d["text_sha256"]
    assert _sha_slabs([np.array(pr.chunks, dtype=np.uint64)], "<u8") == d["chunks_sha256"]
    cli = d["settings"]
    with asgart_amd.Index(pr.data, None) as idx:   # suffix array built on the GPU
        if "sa_sha256_u64" in d:   # n >= 2^32: 64-bit suffix-array entries
            assert _sha_slabs((idx.sa_read(o, min(n, o + slab)) for o in range(0, n, slab)), "<u8") == d["sa_sha256_u64"]
        else:
            assert _sha_slabs((idx.sa_read(o, min(n, o + slab)) for o in range(0, n, slab)), "<u4") == d["sa_sha256_u32"]
        assert d["passes"], name
        for label, want in d["passes"].items():
            st = asgart_amd.RunSettings.from_cli(k=cli["k"], gap=cli["gap"], min_length=cli["min_length"],
                                                 max_cardinality=cli["max_cardinality"],
                                                 reverse=want["reverse"], complement=want["complement"])
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            got = idx.stats().as_dict()
            for key, val in want["counters"].items():
                assert got[key] == val, (name, label, key, got[key], val)
            assert (len(offs) - 1, len(sds)) == (want["n_families"], want["n_sds"]), (name, label)
            assert _sha_slabs([offs], "<u8") == want["fam_offsets_sha256"], (name, label)
            assert _sha_slabs([sds], "<u8") == want["sds_sha256"], (name, label)
        # ... and all passes of the configuration as ONE job (what bench.py times: asgart_search_duplications_passes)
        wants = list(d["passes"].values())
        if len(wants) > 1:
            sts = [asgart_amd.RunSettings.from_cli(k=cli["k"], gap=cli["gap"], min_length=cli["min_length"],
                                                   max_cardinality=cli["max_cardinality"], reverse=w_["reverse"],
                                                   complement=w_["complement"]) for w_ in wants]
            got_all = idx.search_duplications_passes(pr.chunks, sts)
            got = idx.stats().as_dict()
            assert got["passes"] == len(wants), (name, got["passes"])
            for key in wants[0]["counters"]:
                assert got[key] == sum(w_["counters"][key] for w_ in wants), (name, "fused", key)
            for w_, (offs, sds) in zip(wants, got_all):
                assert _sha_slabs([offs], "<u8") == w_["fam_offsets_sha256"], (name, "fused")
                assert _sha_slabs([sds], "<u8") == w_["sds_sha256"], (name, "fused")
            # ... and as R sharded jobs (what N GPUs run: shard r = the r-th slice of every pass), merged by key
            R = 8 if name == "cfg4" else 3
            parts = [idx.search_duplications_passes(pr.chunks, sts, shard=r, n_shards=R, with_keys=True) for r in range(R)]
            for j, w_ in enumerate(wants):
                offs, sds = asgart_amd.merge_shards([p_[j] for p_ in parts])
                assert _sha_slabs([offs], "<u8") == w_["fam_offsets_sha256"], (name, "fused shards", R)
                assert _sha_slabs([sds], "<u8") == w_["sds_sha256"], (name, "fused shards", R)


@pytest.mark.parametrize("kind", ["dna", "bytes", "runs", "dna-batched", "runs-batched"])
def test_wide_suffix_array_builder_matches_oracle(hiplib, kind, monkeypatch):
    """Texts of 2^32 bytes and more take the 64-bit suffix sorter (round 0 class by class, doubling
    rounds as two-word LSD sorts).  ASGART_FORCE_WIDE selects it for small texts: its suffix array must
    equal the oracle's SA-IS, and the GPU verifier must accept it (and reject a damaged one)."""
    rng = np.random.default_rng(11)
    if kind.endswith("-batched"):   # doubling rounds in batches of ~1000 suffixes (whole groups)
        monkeypatch.setenv("ASGART_TEST_WIDE_BATCH", "1000")
        kind = kind.split("-")[0]
    if kind == "dna":
        text = prep.prepare_records(_small_genome(seed=21, lens=(300_000, 200_000), sat_per_record=1,
                                                  sat_copies=(50, 200))).data
    elif kind == "runs":   # long runs and periodic stretches: many doubling rounds
        parts = [np.full(70_000, ord("N"), np.uint8), np.tile(np.frombuffer(b"ACGTTGCA", np.uint8), 9_000),
                 np.full(40_000, ord("A"), np.uint8), rng.choice(np.frombuffer(b"ACGT", np.uint8), size=50_000),
                 np.tile(np.frombuffer(b"AC", np.uint8), 30_000), np.frombuffer(b"$", np.uint8)]
        text = np.concatenate(parts)
    else:                  # arbitrary bytes (asgart_sa_build64 is a general suffix sorter)
        text = rng.integers(0, 256, size=200_000, dtype=np.uint8)
        text[5000:9000] = text[100_000:104_000]
    want = oracle.divsufsort64(text)
    monkeypatch.setenv("ASGART_FORCE_WIDE", "1")
    if kind == "bytes":
        got = asgart_amd.sa_build64(text)           # 32-bit path of the C entry point, for comparison
        assert np.array_equal(got, want)
        return
    with asgart_amd.Index(text, None) as idx:
        assert np.array_equal(idx.sa_read(0, len(text)), want)
        assert idx.check_sa() == 0
    bad = want.copy()
    bad[[1000, 1001]] = bad[[1001, 1000]]
    with asgart_amd.Index(text, bad) as idx:
        assert idx.check_sa() > 0
    monkeypatch.delenv("ASGART_FORCE_WIDE")
    with asgart_amd.Index(text, None) as idx:        # the verifier on a 32-bit index
        assert idx.check_sa() == 0


def test_cfg5_shaped_two_files_wide(hiplib, monkeypatch):
    """BASELINE.json configs[4] scaled down: two "files" (a GRCh38-shaped genome and its 1.2 %-diverged,
    rearranged copy, 50 records) concatenated as the reference concatenates its inputs
    (src/bin/asgart.rs:375-395), through the 64-bit instantiations a 6.1-Gb text selects (suffix sorter,
    slots, positions), direct and RC, against the oracle."""
    pr = prep.prepare_records(synth.config_genome(5, 0.003))
    assert len(pr.map) == 50
    monkeypatch.setenv("ASGART_FORCE_WIDE", "1")
    with asgart_amd.Index(pr.data, None) as idx:
        assert idx.check_sa() == 0
        sa = idx.sa_read(0, len(pr.data))
        oidx = oracle.Index.build(pr.data, sa)
        assert oracle.sa_check(pr.data, sa) == 0
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc), threads=16)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), rc
            assert len(sds) > 100


@pytest.mark.parametrize("build_on_gpu", [False, True])
@pytest.mark.parametrize("window", ["inside", "to_end", "cuts_copy", "tiny_tail"])
def test_trim_matches_oracle(hiplib, window, build_on_gpu):
    """`--trim START END` (reference src/bin/asgart.rs:142-148): suffix array of data[start..end]+'$' shifted by
    +start, whole input searched against it.  Cache entries, searches, per-probe hits and families against the
    oracle, with the sub-range array handed over or built on the GPU, in all four orientations."""
    pr, cli = _battery_case("long_sds")
    n = len(pr.data)
    rng = np.random.default_rng({"inside": 1, "to_end": 2, "cuts_copy": 3, "tiny_tail": 4}[window])
    a = int(rng.integers(0, n // 3))
    b = {"inside": int(rng.integers(n // 2, n - 100)), "to_end": n + 10, "cuts_copy": int(rng.integers(n // 2, n - 100)) | 1,
         "tiny_tail": n - 3}[window]
    trim = prep.validate_trim((a, b), n)
    oidx = oracle.Index.build_trim(pr.data, *trim)
    with asgart_amd.Index(pr.data, None if build_on_gpu else oidx.sa, trim=trim) as idx:
        assert np.array_equal(idx.sa_read(0, len(oidx.sa)), oidx.sa)
        s = asgart_amd.Searcher(idx)
        text = pr.data
        pats8 = [bytes(text[p:p + 8]) for p in rng.integers(0, n - 30, size=1500)]
        pats8 += [bytes(text[p:p + 8]) for p in range(max(0, trim[1] - 40), min(n - 9, trim[1] + 10))]
        pats8 = [p for p in pats8 if b"$" not in p]
        for p, (lo, hi) in zip(pats8, s.cache(pats8)):
            elo, ehi = oidx.cache_get(p)
            assert hi - lo == ehi - elo and (hi == lo or lo == elo), p
        pats = [bytes(text[p:p + 20]) for p in rng.integers(0, n - 30, size=1500)]
        pats += [bytes(text[p:p + 20]) for p in range(max(0, trim[1] - 60), min(n - 21, trim[1] + 10))]
        pats = [p for p in pats if b"$" not in p]
        for p, (lo, hi) in zip(pats, s.search_ranges(pats)):
            exp, (elo, ehi) = oidx.search(p)
            assert hi - lo == len(exp), p
            if len(exp):
                assert np.array_equal(idx.sa_read(lo, hi).astype(np.uint64), exp)
        for reverse, complement in MODES:
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=reverse, complement=complement, **cli), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (window, reverse, complement)
        with pytest.raises(asgart_amd.AsgartError):
            idx.check_sa()
    with pytest.raises(asgart_amd.AsgartError):
        asgart_amd.Index(pr.data, None, trim=(100, n + 5))      # end past the '$': not a validated trim


@pytest.mark.parametrize("k", [25, 40, 47, 90])
def test_trim_with_long_probes(hiplib, k):
    """--trim and a probe longer than one key word together: the literal bisection replay compares key word + tail."""
    pr, cli = _battery_case("long_sds")
    n = len(pr.data)
    rng = np.random.default_rng(k)
    trim = prep.validate_trim((int(rng.integers(0, n // 3)), int(rng.integers(n // 2, n - 100)) | 1), n)
    oidx = oracle.Index.build_trim(pr.data, *trim)
    with asgart_amd.Index(pr.data, None, trim=trim) as idx:
        s = asgart_amd.Searcher(idx)
        text = pr.data
        pats = [bytes(text[p:p + k]) for p in rng.integers(0, n - 50, size=1000)]
        pats += [bytes(text[p:p + k]) for p in range(max(0, trim[1] - 3 * k), min(n - k - 1, trim[1] + 10))]
        pats = [p for p in pats if b"$" not in p]
        for p, (lo, hi) in zip(pats, s.search_ranges(pats)):
            exp, _ = oidx.search(p)
            assert hi - lo == len(exp), p
            if len(exp):
                assert np.array_equal(idx.sa_read(lo, hi).astype(np.uint64), exp)
        for reverse, complement in ((False, False), (True, True)):
            st = asgart_amd.RunSettings.from_cli(k=k, reverse=reverse, complement=complement)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(k=k, reverse=reverse, complement=complement), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (k, reverse)
            assert len(esds) > 0 or reverse or k > 42  # (90 exact bases in a row are rare in this input's diverged copies)


def test_golden_trim_kats_on_gpu(hiplib):
    """The --trim known-answer cases of tests/golden/kat.json through the HIP path."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat.json")))
    seen = 0
    for case in kat["cases"]:
        s = case["settings"]
        if not s.get("trim"):
            continue
        seen += 1
        text = np.frombuffer(case["text"].encode() + b"$", dtype=np.uint8)
        st = asgart_amd.RunSettings.from_cli(k=s["k"], gap=s["gap"], min_length=s["min_length"],
                                             max_cardinality=s["max_cardinality"], reverse=s["reverse"],
                                             complement=s["complement"])
        with asgart_amd.Index(text, None, trim=tuple(s["trim"])) as idx:
            offs, sds = idx.search_duplications_raw([tuple(c) for c in case["chunks"]], st)
        got = [[tuple(int(v) for v in sds[j]) for j in range(int(offs[f]), int(offs[f + 1]))] for f in range(len(offs) - 1)]
        assert got == [[tuple(sd) for sd in fam] for fam in case["families"]], case["name"]
    assert seen >= 4


@pytest.mark.parametrize("name", ["long_sds", "dense_repeats", "satellites"])
def test_multi_device_entry_equals_one_device(hiplib, name):
    """asgart_search_duplications_multi (one host thread per index replica, shard r of n on replica r, results
    concatenated) against the single call and the oracle.  On a one-GPU box the replicas are the same index
    taken several times and clones of it on the same device (asgart_index_clone: device-to-device copy of
    text + suffix array)."""
    pr, cli = _battery_case(name)
    oidx = oracle.Index.build(pr.data)
    with asgart_amd.Index(pr.data, oidx.sa) as idx, idx.clone(0) as rep:
        assert np.array_equal(rep.sa_read(0, 64), oidx.sa[:64])
        for reverse, complement in ((False, False), (True, True)):
            st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
            one = idx.search_duplications_raw(pr.chunks, st)
            exp = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=reverse, complement=complement, **cli), threads=4)
            assert np.array_equal(one[0], exp[0]) and np.array_equal(one[1], exp[1])
            for group in ([idx], [idx, rep], [rep, idx, rep], [idx, idx, rep, rep, idx], [rep] * 8):
                offs, sds = asgart_amd.search_duplications_multi(group, pr.chunks, st)
                assert np.array_equal(offs, one[0]) and np.array_equal(sds, one[1]), (name, len(group), reverse)


def test_progress_array_and_pipelined_calls(hiplib):
    """`progress` (reference src/automaton.rs:98, polled by src/bin/asgart.rs:160-197): every chunk's entry ends
    at its last probe offset; a second call issued from another thread as soon as the first one reports progress
    (what bench.py does with the direct and the -RC pass) returns the same families as calls made one by one."""
    import threading
    import time
    pr, cli = _battery_case("satellites")
    oidx = oracle.Index.build(pr.data)
    k, step, M = 20, 10, cli.get("min_length", 1000)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        sts = [asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli) for rc in (False, True)]
        alone = [idx.search_duplications_raw(pr.chunks, st) for st in sts]
        prog = np.zeros(len(pr.chunks), dtype=np.uint64)
        out = {}
        t = threading.Thread(target=lambda: out.__setitem__(1, idx.search_duplications_raw(pr.chunks, sts[1], 0, 1, prog)))
        t.start()
        while t.is_alive() and not prog.any():
            time.sleep(0.0002)
        out[0] = idx.search_duplications_raw(pr.chunks, sts[0])
        t.join()
        for j in (0, 1):
            assert np.array_equal(out[j][0], alone[j][0]) and np.array_equal(out[j][1], alone[j][1])
        want = [((L - k - step + step - 1) // step) * step if (L >= M and L >= k + step and L - k - step > 0) else 0
                for _, L in pr.chunks]
        assert prog.tolist() == want


def test_passes_entry_point_equals_single_calls(hiplib):
    """asgart_search_duplications_passes: all four orientations in one call (the library issues them itself,
    pipelined, longest extension first) -- every pass equals its own asgart_search_duplications call, whatever
    order the library chose (first call: nothing known; second call: the remembered durations)."""
    pr, cli = _battery_case("satellites")
    oidx = oracle.Index.build(pr.data)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        sts = [asgart_amd.RunSettings.from_cli(reverse=r, complement=c, **cli) for r, c in MODES]
        for _ in range(2):
            got = idx.search_duplications_passes(pr.chunks, sts)
            assert len(got) == 4
            for st, (offs, sds) in zip(sts, got):
                eo, es = idx.search_duplications_raw(pr.chunks, st)
                assert np.array_equal(offs, eo) and np.array_equal(sds, es)
        assert idx.search_duplications_passes(pr.chunks, []) == []
        eo, es = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=True, complement=True, **cli))
        assert np.array_equal(got[3][0], eo) and np.array_equal(got[3][1], es)


def test_cfg3_full_skip_masked_digest(hiplib):
    """BASELINE.json configs[2] as stated: chr1-sized synthetic (249 Mb), direct + RC, --skip-masked."""
    _check_against_oracle_digest("cfg3s")


def test_cfg4_full_digest(hiplib):
    """BASELINE.json configs[3], the benchmarked workload: GRCh38-sized synthetic (3.1 Gb, 25 records),
    direct + RC, on one MI355X, against the oracle's digests of the same seeded genome."""
    _check_against_oracle_digest("cfg4")


def test_cfg3r_repeat_rich_digest(hiplib):
    """A chr1-sized genome whose interspersed repeats are young (synth.repeat_rich_genome: 42 % of it in SINE- /
    LINE-like families at 1-5 % divergence, family sizes below max_cardinality): two probes in five pass the
    presence filter and carry tens of hits, unlike the BASELINE-shaped stand-ins, whose one old high-copy family
    is answered by the filter four times in five.  Direct + RC against the oracle's digests (bench workload cfg3r)."""
    _check_against_oracle_digest("cfg3r")


def test_cfg4rq_realism_workload_digest(hiplib):
    """The realism workload of the bench (cfg4r: GRCh38-sized, repeat-rich, with higher-order satellite arrays -- the input
    whose megabase arrays exposed the cooperative-path cost of round 4 and on which one array IS the extension) at a
    quarter of its size with the arrays at their full length (772 Mb, 44 chunks; the oracle took 74 minutes for the
    direct pass: 1.56 M ProtoSDs): suffix array, both passes -- each as its own call and both as one job -- against
    the oracle's digests."""
    _check_against_oracle_digest("cfg4rq")


def test_cfg5_wide_digest(hiplib):
    """BASELINE.json configs[4] (two files: the GRCh38-shaped genome + its 1.2 %-diverged, rearranged copy,
    reference src/bin/asgart.rs:375-395) at the smallest scale that needs 64-bit suffix-array entries
    (n = 4.32 G >= 2^32): every kernel in its natively wide instantiation -- the 64-bit suffix sorter, 64-bit
    slots in the search, 64-bit positions in every extension tier -- against the oracle's digests of the same
    seeded input (suffix array by sha256, per pass counters, family / ProtoSD counts, sha256 of the results)."""
    _check_against_oracle_digest("cfg5h")


def test_cfg5_full_properties(hiplib):
    """configs[4] at FULL size (6.18 Gb, 64-bit suffix array, one GPU): the oracle cannot hold this input, so the
    size-independent properties are checked -- the suffix array passes the GPU verifier; every ProtoSD lies inside
    the text with arms of at least min_length; families are non-empty and offsets strictly increasing; a second
    call and the concatenation of 3 shards give the identical result; the direct pass pairs the two genomes."""
    pr = prep.prepare_records(synth.config_genome(5, 1.0))
    n = len(pr.data)
    assert n > (1 << 32) and len(pr.chunks) > 50
    with asgart_amd.Index(pr.data, None) as idx:
        assert idx.check_sa() == 0
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            assert len(sds) > 100_000 and int(offs[-1]) == len(sds) and np.all(np.diff(offs.astype(np.int64)) > 0)
            assert np.all(sds[:, 0] + sds[:, 2] <= n) and np.all(sds[:, 1] + sds[:, 3] <= n)
            assert np.all(sds[:, 3] >= 1000)
            offs2, sds2 = idx.search_duplications_raw(pr.chunks, st)
            assert np.array_equal(offs, offs2) and np.array_equal(sds, sds2)
            parts = [idx.search_duplications_raw(pr.chunks, st, shard=r, n_shards=3, with_keys=True) for r in range(3)]
            mo, ms = asgart_amd.merge_shards(parts)
            assert np.array_equal(mo, offs) and np.array_equal(ms, sds), "shards != unsharded"
            cross = int(np.sum((sds[:, 0] < n // 2) != (sds[:, 1] < n // 2)))
            assert cross > 10_000, cross


def test_bench_reads_real_fasta(hiplib, tmp_path):
    """bench.py --fasta: real FASTA input instead of the synthetic workload, one JSON line
    with data = "real"."""
    import json
    import os
    import subprocess
    import sys
    recs = _small_genome(seed=41)
    fa = tmp_path / "tiny.fa"
    with open(fa, "wb") as fh:
        for name, seq in recs:
            fh.write(b">" + name.encode() + b"\n")
            raw = seq.tobytes()
            for o in range(0, len(raw), 80):
                fh.write(raw[o:o + 80] + b"\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--fasta", str(fa), "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["data"] == "real" and line["value"] > 0 and "tiny.fa" in line["config"]["workload"]
    assert line["cpu_baseline"]["kind"] == "port" and line["roofline"]["bound"] == "hbm"
    assert line["work_per_step"]["proto_sds"] > 0


@pytest.mark.parametrize("rc", [False, True])
def test_fasta_to_run_result_end_to_end(hiplib, tmp_path, rc):
    """FASTA files -> prepare -> HIP search step -> FilterNs/ReOrder/ReduceOverlap/Sort -> RunResult,
    against the oracle's search + post-processing chain on the same records."""
    from asgart_amd import postprocess
    recs = synth.make_genome([240_000, 150_000, 90_000], seed=77, sd_per_mb=50, sd_len=(800, 6000),
                             alu_frac=0.05, l1_frac=0.02, sat_per_record=1, sat_copies=(20, 60),
                             short_n_per_mb=15)
    files = []
    for i, part in enumerate((recs[:2], recs[2:])):
        p = tmp_path / f"part{i}.fa"
        with open(p, "wb") as fh:
            for name, seq in part:
                fh.write(b">" + name.encode() + b" some description\n")
                raw = seq.tobytes()
                for o in range(0, len(raw), 70):
                    fh.write(raw[o:o + 70] + b"\n")
        files.append(str(p))
    settings = asgart_amd.RunSettings.from_cli(min_length=400, reverse=rc, complement=rc)
    res = postprocess.search_duplications(files, settings)

    pr = prep.prepare_records(recs)
    oidx = oracle.Index.build(pr.data)
    offs, sds = oidx.run_raw(pr.chunks, oracle.make_settings(min_length=400, reverse=rc, complement=rc))
    eo, es = oracle.postprocess(pr.data, offs, sds)
    want = oracle.families_to_list(eo, es)
    got = [[(sd["global_left_position"], sd["global_right_position"], sd["left_length"], sd["right_length"])
            for sd in fam] for fam in res["families"]]
    assert got == want and len(want) > 0
    assert res["strand"]["name"] == ", ".join(files)
    assert [m["name"] for m in res["strand"]["map"]] == [n for n, _ in recs]
    assert all(sd["reversed"] == rc and sd["complemented"] == rc for fam in res["families"] for sd in fam)
    assert postprocess.to_json(res).startswith('{\n  "strand": {\n    "name": ')


@pytest.mark.parametrize("seed", range(12))
def test_random_sweep_default_and_forced_tiers(hiplib, seed, monkeypatch):
    """Seeded random genomes x random settings (k, gap, min length, cardinality, strand mode), first with
    the default placement and then with every multi-hit segment forced through a randomly chosen tier."""
    rng = np.random.default_rng(5000 + seed)
    lens = [int(x) for x in rng.integers(60_000, 260_000, size=int(rng.integers(1, 4)))]
    gen = dict(sd_per_mb=float(rng.uniform(5, 60)), sd_len=(500, int(rng.integers(2_000, 40_000))),
               alu_frac=float(rng.uniform(0.0, 0.25)), l1_frac=float(rng.uniform(0.0, 0.05)),
               sat_per_record=int(rng.integers(0, 3)), sat_copies=(20, int(rng.integers(60, 500))),
               alu_div=(0.01, float(rng.uniform(0.05, 0.15))))
    recs = synth.make_genome(lens, seed=int(rng.integers(1, 1 << 30)), **gen)
    pr = prep.prepare_records(recs, skip_masked=bool(rng.integers(0, 2)))
    k = int(rng.choice([10, 12, 16, 20, 21]))
    cli = dict(k=k, gap=int(rng.choice([0, 30, 100, 250])), min_length=int(rng.choice([100, 300, 1000])),
               max_cardinality=int(rng.choice([30, 200, 500, 1500])))
    rc = bool(rng.integers(0, 2))
    oidx = oracle.Index.build(pr.data)
    eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
    st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), ("default", cli, rc)
        tier = int(rng.integers(2, 8))
        idx.set_option("force_tier", tier)
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), ("forced", tier, cli, rc)


@pytest.mark.parametrize("tier", [0, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("name", ["dense_repeats", "long_sds", "satellites"])
def test_64bit_slots_and_positions(hiplib, name, tier, monkeypatch):
    """Texts of 2^32 bases and more use 64-bit suffix-array slots and positions (other template
    instantiations of every kernel, smaller arm capacities).  ASGART_FORCE_WIDE selects them for a small
    text: default placement (tier 0) and every forced tier against the oracle."""
    pr, cli = _battery_case(name)
    oidx = oracle.Index.build(pr.data)
    monkeypatch.setenv("ASGART_FORCE_WIDE", "1")
    if tier:
        monkeypatch.setenv("ASGART_FORCE_TIER", str(tier))
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (name, tier, rc)
        # searcher surface on the wide index
        lo, hi = idx.sa_read(0, 16), None
        assert np.array_equal(lo, oidx.sa[:16])


@pytest.mark.parametrize("card", [100, 1000])
@pytest.mark.parametrize("reverse,complement", MODES)
def test_large_intervals_counted_by_bisection(hiplib, reverse, complement, card):
    """k-mer intervals of more than 256 entries are counted from the position-sorted occurrence lists (a bisection
    at the hit filter's threshold, reference src/automaton.rs:105-114) instead of being read.  A tandem array of a
    reverse-palindromic unit makes every probe of every orientation hit hundreds of positions -- in the reversed
    orientations including the position equal to the probe's own offset, which the filter excludes by value.
    Per-probe status / hit lists and the families against the oracle, below and above max_cardinality."""
    rng = np.random.default_rng(5)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    half = rng.integers(0, 4, size=29)
    unit = np.concatenate([half, half[::-1]])                  # its own reverse
    arr = np.tile(unit, 700)
    mut = rng.random(arr.shape) < 0.01
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    text = np.concatenate([bases[arr], np.frombuffer(b"$", dtype=np.uint8)])
    chunks = [(0, len(text) - 1)]
    oidx = oracle.Index.build(text)
    cli = dict(max_cardinality=card, min_length=120)
    with asgart_amd.Index(text, oidx.sa) as idx:
        st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
        ost = oracle.make_settings(reverse=reverse, complement=complement, **cli)
        status, offs, hits = idx.probe_hits(chunks, st)
        nd = oracle.prepare_needle(text, chunks[0], ost)
        e_status, e_offs, e_hits = oidx.probe_hits(nd, 0, ost)
        assert np.array_equal(status, e_status) and np.array_equal(offs, e_offs) and np.array_equal(hits, e_hits)
        if not complement:   # (the unit is not its own complement: those orientations have no hits here)
            assert int(np.diff(e_offs.astype(np.int64)).max(initial=0)) > 256 or int((e_status == 2).sum()) > 1000
        f_offs, sds = idx.search_duplications_raw(chunks, st)
        eoffs, esds = oidx.run_raw(chunks, ost)
        assert np.array_equal(f_offs, eoffs) and np.array_equal(sds, esds), (reverse, complement, card)


@pytest.mark.parametrize("wide", [0, 1])
def test_tier6_with_thousands_of_live_arms(hiplib, wide, monkeypatch):
    """The largest arm-resident shape with several layers in use: a 100-bp tandem array of 650 diverged copies keeps
    many hundreds of arms alive (layers of 512 slots; with 64-bit positions the shape that keeps its left ends in
    HBM).  Forced into tier 6, checked against the oracle, and nothing may fall through to tier 7."""
    rng = np.random.default_rng(77)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    mono = rng.integers(0, 4, size=100)
    arr = np.tile(mono, 650)
    mut = rng.random(arr.shape) < 0.04
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    g = np.concatenate([rng.integers(0, 4, size=3000), arr, rng.integers(0, 4, size=3000)])
    text = np.concatenate([bases[g], np.frombuffer(b"$", dtype=np.uint8)])
    chunks = [(0, len(text) - 1)]
    oidx = oracle.Index.build(text)
    monkeypatch.setenv("ASGART_FORCE_WIDE", str(wide))
    monkeypatch.setenv("ASGART_FORCE_TIER", "6")
    cli = dict(max_cardinality=1000, min_length=150)
    with asgart_amd.Index(text, oidx.sa) as idx:
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
            offs, sds = idx.search_duplications_raw(chunks, st)
            stats = idx.stats()
            key = ("tier6_layers", rc)
            if key not in _ORACLE_CACHE:   # (13 s of oracle time: shared by the two parametrisations)
                _ORACLE_CACHE[key] = oidx.run_raw(chunks, oracle.make_settings(reverse=rc, complement=rc, **cli))
            eoffs, esds = _ORACLE_CACHE[key]
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (wide, rc)
            if not rc:
                assert len(esds) > 1000 and stats.heavy_segments >= 1 and stats.overflow_segments == 0


@pytest.mark.parametrize("wide", [0, 1])
def test_more_live_arms_than_any_fixed_capacity(hiplib, wide, monkeypatch):
    """The reference keeps a chunk's arms in an unbounded Vec (src/automaton.rs:87,173-179).  With
    --max-cardinality 3000 and -g 400 a stretch that strings together short units of nine 2 900-copy repeat
    families holds eight families' worth of live arms at once -- 23 200 (counted by a simulation of the automaton),
    beyond the 16 384 (8 192 with 64-bit positions) slots the last extension tier used to have.  That tier now sizes its HBM slices from the
    bound max_cardinality * (t* + 1); families must equal the oracle's, with 32- and 64-bit positions."""
    rng = np.random.default_rng(31)
    fam, copies, unit, spacer = 9, 2900, 60, 480   # (spacers wider than the gap: a copy's arm never takes the next copy's hit)
    units = [rng.integers(0, 4, size=unit) for _ in range(fam)]
    query = np.concatenate(units)
    parts = [rng.integers(0, 4, size=300), query, rng.integers(0, 4, size=2000)]
    order = rng.permutation(fam * copies) % fam
    for f in order:
        parts.append(units[f])
        parts.append(rng.integers(0, 4, size=spacer))
    text = np.concatenate([np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)], np.frombuffer(b"$", dtype=np.uint8)])
    chunks = [(0, 300 + len(query) + 1500)]   # only the stretch is the needle; its hits lie all over the text
    cli = dict(gap=400, max_cardinality=3000, min_length=50)
    if wide:
        monkeypatch.setenv("ASGART_FORCE_WIDE", "1")
    oidx = oracle.Index.build(text)
    with asgart_amd.Index(text, oidx.sa) as idx:
        st = asgart_amd.RunSettings.from_cli(**cli)
        offs, sds = idx.search_duplications_raw(chunks, st)
        eo, es = oidx.run_raw(chunks, oracle.make_settings(**cli))
        assert np.array_equal(offs, eo) and np.array_equal(sds, es)
        assert len(sds) > 20_000


@pytest.mark.parametrize("bits", [2, 5])
@pytest.mark.parametrize("tier", [2, 3, 5])
def test_arm_kernel_generation_wrap(hiplib, bits, tier, monkeypatch):
    """The per-probe hit tables of the arm-resident kernel are never cleared: heads carry a generation
    number, and the tables are wiped when the counter wraps (every 4M probes of a workgroup).  With a
    2- or 5-bit counter the wrap happens every few probes."""
    pr, cli = _battery_case("satellites")
    oidx = oracle.Index.build(pr.data)
    monkeypatch.setenv("ASGART_FORCE_TIER", str(tier))
    monkeypatch.setenv("ASGART_TEST_GENBITS", str(bits))
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
            offs, sds = idx.search_duplications_raw(pr.chunks, st)
            eoffs, esds = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
            assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (bits, tier, rc)


@pytest.mark.parametrize("rc", [(False, False), (True, True), (True, False), (False, True)])
def test_compute_scores_match_oracle(hiplib, rc):
    """`--compute-score`: Levenshtein identities of the GPU (anti-diagonal DP, LDS and HBM-scratch
    paths) against the oracle's restatement of ProtoSD::levenshtein, bit-exact as f32."""
    rng = np.random.default_rng(900 + 2 * rc[0] + rc[1])
    recs = synth.make_genome([300_000], seed=91, sd_per_mb=30, sd_len=(500, 9000), alu_frac=0.05, short_n_per_mb=30)
    pr = prep.prepare_records(recs)
    n = len(pr.data) - 1
    sds = []   # (incl. arms of one band, several bands, and the multi-wave kernel for >= 8192 rows)
    for ll, rl in [(1, 1), (1, 40), (40, 1), (63, 64), (255, 256), (1000, 900), (4094, 4094), (4095, 100),
                   (4096, 4200), (9000, 8700), (8191, 300), (8192, 8192), (20000, 1000), (17000, 33000), (40000, 150), (9000, 1), (1, 9000),
                   (16384, 64), (16385, 127), (25000, 20000)] + [tuple(int(v) for v in rng.integers(20, 3000, 2)) for _ in range(40)]:
        left = int(rng.integers(0, n - ll - 1)); right = int(rng.integers(0, n - rl - 1))
        sds.append((left, right, ll, rl))
    # planted near-identical pairs: right arm = left arm shifted into a diverged copy region
    for _ in range(10):
        ll = int(rng.integers(200, 2500)); left = int(rng.integers(0, n - 2 * ll - 10))
        sds.append((left, left + 3, ll, ll + int(rng.integers(0, 5))))
    arr = np.array(sds, dtype=np.uint64)
    with asgart_amd.Index(pr.data, None) as idx:
        got = idx.compute_scores(arr, rc[0], rc[1])
        # errors: range past the text, two empty arms
        with pytest.raises(asgart_amd.AsgartError):
            idx.compute_scores(np.array([[n - 5, 0, 10, 3]], dtype=np.uint64))
        with pytest.raises(asgart_amd.AsgartError):
            idx.compute_scores(np.array([[5, 9, 0, 0]], dtype=np.uint64))
    want = np.array([oracle.levenshtein_identity(pr.data, sd, rc[0], rc[1]) for sd in sds], dtype=np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert want.min() < 60.0 and (rc != (False, False) or want.max() > 99.0)


def test_compute_score_step_end_to_end(hiplib, tmp_path):
    """FASTA -> search -> FilterNs/ReOrder/ReduceOverlap -> ComputeScore (GPU) -> Sort -> RunResult JSON:
    the identities in the JSON equal the oracle's for the very same duplications."""
    from asgart_amd import postprocess
    recs = synth.make_genome([260_000, 120_000], seed=92, sd_per_mb=40, sd_len=(800, 5000), alu_frac=0.03)
    p = tmp_path / "g.fa"
    with open(p, "wb") as fh:
        for name, seq in recs:
            fh.write(b">" + name.encode() + b"\n" + seq.tobytes() + b"\n")
    settings = asgart_amd.RunSettings.from_cli(min_length=500, reverse=True, complement=True)
    res = postprocess.search_duplications([str(p)], settings, compute_score=True)
    pr = prep.prepare_records(recs)
    n_checked = 0
    for fam in res["families"]:
        for sd in fam:
            tup = (sd["global_left_position"], sd["global_right_position"], sd["left_length"], sd["right_length"])
            want = oracle.levenshtein_identity(pr.data, tup, True, True)
            assert np.float32(sd["identity"]) == want, tup
            n_checked += 1
    assert n_checked > 5


def test_out_of_memory_paths_release_what_they_hold(hiplib):
    """Every device allocation of an index preparation and of a first search call fails in turn (option
    test_fail_alloc: the (n+1)-th allocation from now on is refused once).  The call then either does without what
    it could not get -- no position-sorted lists, no presence filter: the same duplications -- or fails with
    ASGART_E_OOM; in both cases the device has its memory back once the index is closed (a leak on this path would
    stay for the life of the process: the block kept by build_rank_lists when its second buffer failed)."""
    import torch

    E_OOM = -2  # ASGART_E_OOM (include/asgart_hip.h)
    pr, cli = _battery_case("dense_repeats")
    oidx = oracle.Index.build(pr.data)
    st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True, **cli)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        ref = idx.search_duplications_raw(pr.chunks, st)

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    # Memory the HIP runtime keeps for itself grows while new code paths warm up (kernels that only run without the
    # lists or the filter are loaded, their scratch is reserved), in steps of a few MiB and once: the sweep over all
    # failure points is repeated, and the last repetition must not cost a byte more than the one before.
    done, refused = 0, 0
    after = []
    try:
        for sweep in range(3):
            for n in range(0, 70):
                idx = asgart_amd.Index(pr.data, oidx.sa)
                try:
                    idx.set_option("test_fail_alloc", n)
                    try:
                        idx.prepare(20)
                        got = idx.search_duplications_raw(pr.chunks, st)
                        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), n
                        done += 1
                    except asgart_amd.AsgartError as e:
                        assert e.code == E_OOM, (n, str(e))
                        refused += 1
                    idx.set_option("test_fail_alloc", -1)
                finally:
                    idx.close()
            after.append(free_bytes())
    finally:
        with asgart_amd.Index(pr.data, oidx.sa) as idx:
            idx.set_option("test_fail_alloc", -1)
    assert after[1] - after[2] <= (1 << 20), [a - after[0] for a in after]
    assert refused > 0 and done > 0, (done, refused)


def test_watchdog_gives_up_on_a_stalled_device_and_names_the_phase(hiplib):
    """Option test_stall_s parks a kernel that does nothing on the call's stream; with watchdog_s below it the call
    must come back with ASGART_E_HIP, say what it waited for, and the index must refuse further calls (its stream still
    holds the stuck work) -- instead of sitting in the call for as long as the device takes."""
    pr, cli = _battery_case("dense_repeats")
    st = asgart_amd.RunSettings.from_cli(**cli)
    with asgart_amd.Index(pr.data, None) as idx:
        ref = idx.search_duplications_raw(pr.chunks, st)
        idx.set_option("watchdog_s", 1)
        idx.set_option("test_stall_s", 4)
        t0 = time.time()
        with pytest.raises(asgart_amd.AsgartError) as e:
            idx.search_duplications_raw(pr.chunks, st)
        assert e.value.code == -3 and "watchdog" in str(e.value) and "probe search" in str(e.value)
        assert time.time() - t0 < 3.5
        idx.set_option("test_stall_s", 0)
        with pytest.raises(asgart_amd.AsgartError) as e2:
            idx.search_duplications_raw(pr.chunks, st)
        assert "fresh process" in str(e2.value)
        time.sleep(3.5)   # (the parked kernel ends by itself; the index is closed behind it: its teardown drains the streams
                          # with a polled wait and would leak the index rather than hang if the kernel were still there)
    # a long wait with a patient watchdog is not an error: the same stall, the default limit
    with asgart_amd.Index(pr.data, None) as idx:
        idx.set_option("test_stall_s", 2)
        got = idx.search_duplications_raw(pr.chunks, st)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_block_cache_is_trimmed_and_filter_can_be_switched_off(hiplib):
    """(advisor, round 3) The blocks the library keeps for reuse go back to the device when the last index closes and on
    asgart_trim_cache; option kfilter_bits = 0 really searches without the presence filter (its position bitmaps used
    to survive and go on filtering) and switching it back on rebuilds both without leaking the old bitmap."""
    import torch

    pr, cli = _battery_case("dense_repeats")
    oidx = oracle.Index.build(pr.data)
    st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True, **cli)
    exp = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=True, complement=True, **cli))

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        a = idx.search_duplications_raw(pr.chunks, st)
        rej_on = idx.stats(1).probes_filter_rejected
        idx.set_option("kfilter_bits", 0)
        b = idx.search_duplications_raw(pr.chunks, st)
        rej_off = idx.stats(1).probes_filter_rejected
        idx.set_option("kfilter_bits", 30)
        f0 = None
        for _ in range(4):   # (rebuilding the filter again and again must not grow the footprint)
            idx.set_option("kfilter_bits", 0)
            idx.set_option("kfilter_bits", 30)
            c = idx.search_duplications_raw(pr.chunks, st)
            f1 = free_bytes()
            assert f0 is None or f1 >= f0 - (1 << 20), (f0, f1)
            f0 = f1
        for got in (a, b, c):
            assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        assert rej_on > 0 and rej_off == 0
    assert asgart_amd.trim_cache(0) >= 0
    assert asgart_amd.trim_cache(0) == 0   # nothing left to give back


def test_replica_from_device_buffers_matches(hiplib):
    """asgart_index_export + asgart_index_create_device: a second index built from the first one's device buffers
    (what a one-process-per-GPU host does with the broadcast text and suffix array) gives the same families."""
    pr, cli = _battery_case("dense_repeats")
    st = asgart_amd.RunSettings.from_cli(**cli)
    with asgart_amd.Index(pr.data, None) as idx:
        ref = idx.search_duplications_raw(pr.chunks, st)
        t_ptr, sa_ptr, width = idx.export()
        assert width == 4 and t_ptr and sa_ptr
        with asgart_amd.Index.from_device(t_ptr, len(pr.data), sa_ptr, width) as rep:
            got = rep.search_duplications_raw(pr.chunks, st)
            assert rep.check_sa() == 0
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    with pytest.raises(asgart_amd.AsgartError):
        asgart_amd.Index.from_device(t_ptr, len(pr.data), sa_ptr, 8)   # wrong entry width for this text


@pytest.mark.parametrize("name", ["dense_repeats", "satellites", "masked", "k12"])
def test_refined_position_bits_reject_more_and_change_nothing(hiplib, name, monkeypatch):
    """Option posbits = 2 (default): a text position keeps its filter bit only if a hit of its probe can be KEPT (an
    occurrence of the probe's k-mer behind the probe, src/automaton.rs:105-114 in text coordinates) -- decided once, when
    the bits are built.  Against posbits = 1 (the k-mer filter's answer): more probes answered without a lookup, and
    NOTHING else changes -- families, per-probe hit rows, and every counter incl. raw_hits (the intervals of the probes
    the bits answered are looked up when the statistics are asked for), in every orientation, 32- and 64-bit slots."""
    pr, cli = _battery_case(name)
    seen = {}
    for wide in ("0", "1"):
        monkeypatch.setenv("ASGART_FORCE_WIDE", wide)
        for pb in ("1", "2"):
            monkeypatch.setenv("ASGART_POSBITS", pb)
            with asgart_amd.Index(pr.data, None) as idx:
                for reverse, complement in MODES:
                    st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
                    fam = idx.search_duplications_raw(pr.chunks, st)
                    stt = idx.stats(1).as_dict()
                    rows = idx.probe_hits(pr.chunks, st)
                    key = (wide, reverse, complement)
                    if pb == "1":
                        seen[key] = (fam, stt, rows)
                        continue
                    fam1, stt1, rows1 = seen[key]
                    assert all(np.array_equal(a, b) for a, b in zip(fam, fam1)), (name, key)
                    assert all(np.array_equal(a, b) for a, b in zip(rows, rows1)), (name, key)
                    for c in ("probes_total", "probes_n_skipped", "probes_searched", "probes_card_skipped", "probes_with_hits",
                              "raw_hits", "filtered_hits", "segments", "families", "proto_sds"):
                        assert stt[c] == stt1[c], (name, key, c, stt[c], stt1[c])
                    assert stt["probes_filter_rejected"] >= stt1["probes_filter_rejected"], (name, key)
                    if (reverse, complement) == (False, False) and name != "k12":
                        assert stt["probes_filter_rejected"] > stt1["probes_filter_rejected"], (name, key)


@pytest.mark.parametrize("name", ["dense_repeats", "satellites", "masked"])
def test_learned_position_bits_hold_for_any_chunk_list(hiplib, name, monkeypatch):
    """lazy_aux (the default): an orientation's position bits are LEARNED by its searches -- a probe found without an
    occurrence that could be kept clears the bit of the text position it covers.  The fact is one about the text and
    the position (src/automaton.rs:105-114 in text coordinates), so the bits must serve any later chunk list over the
    same index: whole records first, then the same records cut differently (shifted starts: other probe phases; halves:
    other needle offsets, other `L - i` of the reversed needle), each compared with the oracle run over the same list,
    in every orientation, three rounds over the lists (bits learned under one list are used under the others)."""
    monkeypatch.setenv("ASGART_LAZY_AUX", "1")
    pr, cli = _battery_case(name)
    oidx = oracle.Index.build(pr.data)
    lists = [list(pr.chunks),
             [(s0 + 3, l0 - 7) for s0, l0 in pr.chunks if l0 > 2000],
             [c for s0, l0 in pr.chunks if l0 > 4000 for c in ((s0, l0 // 2 + 11), (s0 + l0 // 2 - 5, l0 - l0 // 2 + 5))]]
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for rnd in range(3):
            for li, chunks in enumerate(lists):
                for reverse, complement in MODES:
                    st = asgart_amd.RunSettings.from_cli(reverse=reverse, complement=complement, **cli)
                    got = idx.search_duplications_raw(chunks, st)
                    key = ("learned", name, li, reverse, complement)
                    if key not in _ORACLE_CACHE:
                        _ORACLE_CACHE[key] = oidx.run_raw(chunks, oracle.make_settings(reverse=reverse, complement=complement, **cli), threads=4)
                    exp = _ORACLE_CACHE[key]
                    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), (name, rnd, li, reverse, complement)
        assert idx.stats(1).probes_filter_rejected > 0


def test_lazy_filter_and_lists_same_results(hiplib, monkeypatch):
    """Default behaviour of an index (option lazy_aux = 1; the suite otherwise runs with 0 so that every first call
    takes the filtered paths): the first search of an orientation runs without the presence filter and without the
    position-sorted lists, the second one with both -- identical families, and identical to the oracle."""
    monkeypatch.setenv("ASGART_LAZY_AUX", "1")
    pr, cli = _battery_case("dense_repeats")
    oidx = oracle.Index.build(pr.data)
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
            exp = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli))
            first = idx.search_duplications_raw(pr.chunks, st)
            rej_first = idx.stats(1).probes_filter_rejected
            second = idx.search_duplications_raw(pr.chunks, st)
            rej_second = idx.stats(1).probes_filter_rejected
            assert rej_first == 0 and rej_second > 0
            for got in (first, second):
                assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        # the passes call: both orientations in one call, every structure in place by now
        sts = [asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli) for rc in (False, True)]
        both = idx.search_duplications_passes(pr.chunks, sts)
        for rc, got in zip((False, True), both):
            exp = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli))
            assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])


@pytest.mark.parametrize("wide", [0, 1])
def test_specialised_wave_kernel_with_generation_wraps(hiplib, wide, monkeypatch):
    """The kernel with specialised waves (K8, extend_k8_dev.hpp: a planning wave writes the steps' commands, a ranking
    wave makes the new arms' first offers, arm waves do nothing but their arms) runs tier 3 in two shapes: 5 x 896
    slots and, with 64-bit positions, 4 x 896.  Every segment with a multi-hit probe forced through it, with a
    generation counter that wraps every few probes in one of the passes: identical to the oracle."""
    tier = 3
    monkeypatch.setenv("ASGART_FORCE_TIER", str(tier))
    monkeypatch.setenv("ASGART_FORCE_WIDE", str(wide))
    for name in ("dense_repeats", "satellites"):
        pr, cli = _battery_case(name)
        oidx = oracle.Index.build(pr.data)
        with asgart_amd.Index(pr.data, oidx.sa) as idx:
            for rc in (False, True):
                idx.set_option("test_genbits", 3 if rc else 22)
                st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
                offs, sds = idx.search_duplications_raw(pr.chunks, st)
                key = ("k8_battery", name, rc)
                if key not in _ORACLE_CACHE:
                    _ORACLE_CACHE[key] = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
                eoffs, esds = _ORACLE_CACHE[key]
                assert np.array_equal(offs, eoffs) and np.array_equal(sds, esds), (name, tier, wide, rc)


@pytest.mark.parametrize("delay", [0, 2_000, 20_000])
def test_k8_free_counts_do_not_depend_on_timing(hiplib, delay, monkeypatch):
    """K8 (tier 3's kernel) has ONE barrier per hit-probe: the ranking wave reads the free-slot counts the arm waves
    published in front of that barrier, and the arm waves publish the next counts in front of the next one.  With one
    block of counts nothing but time separated the two (round 4: documented, shipped); they are double-buffered by
    step parity now.  Option test_k8_delay makes the ranking wave wait that many cycles right before the read -- far
    longer than an arm wave with a few arms needs to reach its next publication -- and the families must not change:
    tandem-array cases of tools/fuzz_k8.py, every multi-hit segment forced through tier 3, generation wraps included.
    (tools/k8_race.sh builds the kernel with the single block again, -DK8_SINGLE_FREE, and shows the same cases fail.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_k8

    monkeypatch.setenv("ASGART_TEST_K8_DELAY", str(delay))
    for seed in (0, 3, 5, 8, 13, 21):
        text, cli, genbits = fuzz_k8.make_case(seed)
        chunks = [(0, len(text) - 1)]
        oidx = oracle.Index.build(text)
        with asgart_amd.Index(text, oidx.sa) as idx:
            idx.set_option("force_tier", 3)
            idx.set_option("test_genbits", genbits)
            for rc in (False, True):
                st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli)
                offs, sds = idx.search_duplications_raw(chunks, st)
                key = ("k8_delay", seed, rc)
                if key not in _ORACLE_CACHE:
                    _ORACLE_CACHE[key] = oidx.run_raw(chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4)
                eo, es = _ORACLE_CACHE[key]
                assert np.array_equal(offs, eo) and np.array_equal(sds, es), (seed, rc, delay, cli)


@pytest.mark.parametrize("shape", [(256, 128, 512, 0), (256, 1024, 512, 0), (128, 512, 256, 0), (256, 128, 512, 1024), (128, 64, 256, 24576)])
@pytest.mark.parametrize("wide", [0, 1])
def test_long_segments_as_ranges_equal_whole_segments(hiplib, shape, wide, monkeypatch):
    """Option split: a long segment runs as RANGES side by side -- every range starts from an empty arm list split_warm
    probes in front of its cut, and what it holds at the cut (every arm, every field; family open or not; a held flush) is
    compared on the device with what the range in front of the cut holds there.  Where the cuts hold, the ranges' records
    (family ordinals counted on from the ranges before, creation order by (probe, hit)) must be the whole segment's; where
    one does not, the ranges in front of it stand, the rest of the segment runs as ONE more run from a checked state (the whole
    segment again when its first cut fails), and the index gives that segment's ranges a longer warm-up in the next call -- as
    far back as the oldest arm at the failed cut was born, or the longest the limit allows -- while that stays within two ranges and
    option split_warm_max (the shape's 4th number; 0: never); beyond, it plans only the cuts that held: after
    log2(limit / split_warm) + 2 calls nothing is refused any more.  Tandem-array cases of
    tools/fuzz_k8.py with ranges of 128-256 probes (the shipped 8192 never cut a test-sized segment), every multi-hit
    segment forced through the long shape, generation wraps every few probes in some cases: families, ProtoSDs AND keys
    equal to the uncut run and to the oracle, for single calls and for both orientations as one job; both outcomes
    (joined up / refused) must occur over the cases.  wide = 1: the same with 64-bit positions (ASGART_FORCE_WIDE: the
    instantiations a text of 2^32 bases and more runs -- four layers of arms, dumps of twelve words per arm)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_k8

    monkeypatch.setenv("ASGART_FORCE_WIDE", str(wide))
    ln, warm, mn, warm_max = shape
    n_grow = 0
    while (warm << n_grow) < min(warm_max, 2 * ln):
        n_grow += 1
    joined = refused = sharded_cut = 0
    for seed in (1, 4, 5, 6, 8, 11, 101, 107, 117, 123):
        text, cli, genbits = fuzz_k8.make_case(seed)
        chunks = [(0, len(text) - 1)]
        oidx = oracle.Index.build(text)
        with asgart_amd.Index(text, oidx.sa) as idx:
            idx.set_option("force_tier", 3)
            idx.set_option("test_genbits", genbits)
            sts = [asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli) for rc in (False, True)]
            idx.set_option("split", 0)
            whole = [idx.search_duplications_raw(chunks, st, with_keys=True) for st in sts]
            assert idx.stats().split_segments == 0
            key = ("ranges", seed)
            if key not in _ORACLE_CACHE:
                _ORACLE_CACHE[key] = oidx.run_raw(chunks, oracle.make_settings(reverse=False, complement=False, **cli), threads=4)
            eo, es = _ORACLE_CACHE[key]
            assert np.array_equal(whole[0][0], eo) and np.array_equal(whole[0][1], es), (seed, cli)
            idx.set_option("split", 1 + wide)   # (64-bit positions: option split = 2)
            idx.set_option("split_len", ln)
            idx.set_option("split_warm", warm)
            idx.set_option("split_min", mn)
            idx.set_option("split_warm_max", warm_max)
            # (a segment with a cut that did not hold keeps the ranges in front of it and runs the rest as one more run; with
            # the warm-up at its limit the next call plans only the cuts that held: it refuses nothing)
            for rep in range(n_grow + 3):
                for j, st in enumerate(sts):
                    got = idx.search_duplications_raw(chunks, st, with_keys=True)
                    stt = idx.stats()
                    assert all(np.array_equal(a, b) for a, b in zip(got, whole[j])), (seed, shape, rep, j, cli)
                    if rep == 0:
                        joined += stt.split_segments - stt.split_refused
                        refused += stt.split_refused
                    elif rep > n_grow:
                        assert stt.split_refused == 0, (seed, shape, rep, j)
            both = idx.search_duplications_passes(chunks, sts, with_keys=True)
            for j in range(2):
                assert all(np.array_equal(a, b) for a, b in zip(both[j], whole[j])), (seed, shape, "one job", j)
            # sharded calls cut the segments their window holds whole (a fresh verdict per segment: new index)
        with asgart_amd.Index(text, oidx.sa) as idx:
            idx.set_option("force_tier", 3)
            idx.set_option("test_genbits", genbits)
            idx.set_option("split", 1 + wide)
            idx.set_option("split_len", ln)
            idx.set_option("split_warm", warm)
            idx.set_option("split_min", mn)
            idx.set_option("split_warm_max", warm_max)
            n_cut = 0
            parts = []
            for r in range(3):
                parts.append(idx.search_duplications_raw(chunks, sts[0], shard=r, n_shards=3, with_keys=True))
                n_cut += idx.stats().split_segments
            mo, ms = asgart_amd.merge_shards(parts)
            assert np.array_equal(mo, whole[0][0]) and np.array_equal(ms, whole[0][1]), (seed, shape, "3 shards")
            sharded_cut += n_cut
            # ... and so does a sharded call over BOTH orientations as one job: shard r takes slice r of each pass
            for R in (2, 3, 8):
                parts2 = []
                for r in range(R):
                    parts2.append(idx.search_duplications_passes(chunks, sts, shard=r, n_shards=R, with_keys=True))
                    assert idx.stats().passes == 2, (seed, shape, R, r)
                for j in range(2):
                    mo, ms = asgart_amd.merge_shards([q[j] for q in parts2])
                    assert np.array_equal(mo, whole[j][0]) and np.array_equal(ms, whole[j][1]), (seed, shape, R, "fused shards", j)
                    assert np.array_equal(np.sort(np.concatenate([q[j][2] for q in parts2])), whole[j][2]), (seed, shape, R, "keys", j)
    assert sharded_cut > 0, shape
    assert joined > 0 and refused > 0, (shape, joined, refused)


@pytest.mark.parametrize("block", range(6))
def test_randomised_cut_cases_against_the_oracle(hiplib, block, monkeypatch):
    """The randomised evidence of the range scheme, on the driver's box: 60 cases of tools/fuzz_k8.py (tandem arrays of
    3-400 bp monomers, random probe size / gap / minimum length, 3- and 4-bit table generations among them) with the long
    segments cut into ranges of 128-256 probes, every multi-hit segment forced through the long shape.  Per case: both
    orientations as single calls, both as ONE job, and both as one job over 2 shards merged by key -- families and
    ProtoSDs equal to the ORACLE's, twice (the second round gives the segments with a failed cut a longer warm-up).  Odd
    blocks run with 64-bit positions (ASGART_FORCE_WIDE)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_k8

    monkeypatch.setenv("ASGART_FORCE_WIDE", str(block % 2))

    shapes = ((256, 128, 512), (128, 512, 256), (256, 1024, 512))
    cut = refused = 0
    for seed in range(200 + 10 * block, 210 + 10 * block):
        text, cli, genbits = fuzz_k8.make_case(seed)
        chunks = [(0, len(text) - 1)]
        oidx = oracle.Index.build(text)
        exp = [oidx.run_raw(chunks, oracle.make_settings(reverse=rc, complement=rc, **cli), threads=4) for rc in (False, True)]
        ln, warm, mn = shapes[seed % 3]
        with asgart_amd.Index(text, oidx.sa) as idx:
            idx.set_option("force_tier", 3)
            idx.set_option("test_genbits", genbits)
            idx.set_option("split", 1 + block % 2)   # (64-bit positions: option split = 2)
            idx.set_option("split_len", ln)
            idx.set_option("split_warm", warm)
            idx.set_option("split_min", mn)
            sts = [asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc, **cli) for rc in (False, True)]
            for rep in range(2):
                for j, st in enumerate(sts):
                    got = idx.search_duplications_raw(chunks, st)
                    assert np.array_equal(got[0], exp[j][0]) and np.array_equal(got[1], exp[j][1]), (seed, cli, genbits, rep, j)
                    stt = idx.stats()
                    cut += stt.split_segments
                    refused += stt.split_refused
                both = idx.search_duplications_passes(chunks, sts)
                for j in range(2):
                    assert np.array_equal(both[j][0], exp[j][0]) and np.array_equal(both[j][1], exp[j][1]), (seed, cli, "one job", rep, j)
                parts = [idx.search_duplications_passes(chunks, sts, shard=r, n_shards=2, with_keys=True) for r in range(2)]
                for j in range(2):
                    mo, ms = asgart_amd.merge_shards([q[j] for q in parts])
                    assert np.array_equal(mo, exp[j][0]) and np.array_equal(ms, exp[j][1]), (seed, cli, "2 shards", rep, j)
    assert cut > 0, (block, cut, refused)


@pytest.mark.parametrize("name", ["satellites", "dense_repeats", "masked", "k31_odd", "long_sds"])
def test_fused_passes_equal_single_calls_keys_included(hiplib, name):
    """asgart_search_duplications_passes runs passes that differ in orientation only as ONE job (the probe sequence is
    pass 0's chunks, then pass 1's ...; one front, one launch per extension tier over the merged segment list): every
    pass must come back exactly as its own single call returns it -- family offsets, ProtoSDs AND keys (segment start
    probe counted from the start of the pass) -- for any selection and order of orientations; passes with different
    settings and option fuse_passes = 0 take the pipelined path and give the same."""
    pr, cli = _battery_case(name)
    with asgart_amd.Index(pr.data, None) as idx:
        sts = [asgart_amd.RunSettings.from_cli(reverse=r, complement=c, **cli) for r, c in MODES]
        single = [idx.search_duplications_raw(pr.chunks, st, with_keys=True) for st in sts]
        tot = [idx.search_duplications_raw(pr.chunks, st) and idx.stats().as_dict() for st in sts]
        for sel in ([0, 3], [3, 0], [0, 1, 2, 3], [2, 1, 3], [3, 3]):
            got = idx.search_duplications_passes(pr.chunks, [sts[j] for j in sel], with_keys=True)
            stt = idx.stats().as_dict()
            assert stt["passes"] == len(sel)
            for key in ("probes_total", "probes_searched", "raw_hits", "filtered_hits", "families", "proto_sds"):
                assert stt[key] == sum(tot[j][key] for j in sel), (name, sel, key)
            for j, g in zip(sel, got):
                for a, b in zip(g, single[j]):
                    assert np.array_equal(a, b), (name, sel, j)
        # SHARDED: shard r of R runs the r-th slice of EVERY pass as one job (one window per pass); the shards' families merged
        # by key -- or concatenated in shard order -- are each pass's own result, keys included
        for R, sel in ((2, [0, 3]), (3, [3, 0]), (8, [0, 3]), (3, [0, 1, 2, 3]), (61, [2, 3])):
            parts = []
            for r in range(R):
                parts.append(idx.search_duplications_passes(pr.chunks, [sts[j] for j in sel], shard=r, n_shards=R, with_keys=True))
                assert idx.stats().as_dict()["passes"] == len(sel), (name, R, r)
            for q, j in enumerate(sel):
                mo, ms = asgart_amd.merge_shards([p_[q] for p_ in parts])
                assert np.array_equal(mo, single[j][0]) and np.array_equal(ms, single[j][1]), (name, R, sel, j)
                assert np.array_equal(np.concatenate([p_[q][1] for p_ in parts]), single[j][1]), (name, R, sel, j, "rank order")
                assert np.array_equal(np.concatenate([p_[q][2] for p_ in parts]), single[j][2]), (name, R, sel, j, "keys")
        other = asgart_amd.RunSettings.from_cli(reverse=True, complement=True, **dict(cli, min_length=cli.get("min_length", 1000) + 7))
        got = idx.search_duplications_passes(pr.chunks, [sts[0], other], with_keys=True)
        assert idx.stats().as_dict()["passes"] == 1      # different settings: two pipelined calls
        assert all(np.array_equal(a, b) for a, b in zip(got[0], single[0]))
        assert all(np.array_equal(a, b) for a, b in zip(got[1], idx.search_duplications_raw(pr.chunks, other, with_keys=True)))
        idx.set_option("fuse_passes", 0)
        got = idx.search_duplications_passes(pr.chunks, [sts[0], sts[3]], with_keys=True)
        assert idx.stats().as_dict()["passes"] == 1
        for j, g in zip((0, 3), got):
            assert all(np.array_equal(a, b) for a, b in zip(g, single[j]))
        # the default (fuse_passes = 1) MEASURES: once a call that ran as one job has seen one segment be its extension (with the
        # threshold at 1 % every extension counts as that), the calls are timed both ways in turn -- one job, pipelined, one
        # job, pipelined -- and the faster way is kept; 1000 %: always one job
        idx.set_option("fuse_passes", 1)
        for pct in (1, 1000):
            idx.set_option("fuse_pole_pct", pct)
            sel = (0, 0) if pct == 1 else (0, 0, 3)   # (selections the index has no verdict for yet, with hits in them)
            pair = [sts[j] for j in sel]
            passes, poles = [], []
            for call in range(7):
                got = idx.search_duplications_passes(pr.chunks, pair, with_keys=True)
                st1 = idx.stats().as_dict()
                passes.append(st1["passes"])
                # (the longest segment is measured in the workgroup tiers: an input whose segments all run on the one-wave tier
                # has none, and stays one job)
                if st1["passes"] == len(sel):
                    poles.append(st1["ms_longest_segment"] * 100.0 > st1["ms_extend"] * pct)
                for j, g in zip(sel, got):
                    assert all(np.array_equal(a, b) for a, b in zip(g, single[j])), (name, pct, call)
            n = len(sel)
            if pct == 1 and all(poles):
                assert passes[:5] == [n, n, 1, n, 1] and passes[5] == passes[6], (name, passes)
            elif not any(poles):
                assert passes == [n] * 7, (name, pct, passes)


@pytest.mark.parametrize("skip_masked", [False, True])
def test_prepare_data_on_the_gpu_matches_host_and_oracle(hiplib, skip_masked):
    """asgart_prepare_data (normalisation, chunking at N-runs > 5000 per record, '$', index from the same device buffer)
    against the numpy statement of prepare_data AND the oracle's literal one (oracle_normalise / oracle_find_chunks):
    records with lower-case stretches, foreign letters, N-runs of 4999 / 5000 / 5001 / 30 000 bases at the start, in the
    middle, at the end and across a record boundary, an all-N record, an empty record; then the index it returns gives
    the same families as one built from the host text."""
    rng = np.random.default_rng(31)

    def dna(n):
        return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()

    def with_runs(n, runs):
        s_ = dna(n)
        low = rng.random(n) < 0.2
        s_[low] |= 0x20                                   # soft-masked
        odd = rng.random(n) < 0.001
        s_[odd] = rng.choice(np.frombuffer(b"RYKMxn-*", dtype=np.uint8), size=int(odd.sum()))
        for a, ln in runs:
            s_[a:a + ln] = ord("N") if (a // 7) % 2 else ord("n")
        return s_

    recs = [("r0", with_runs(120_000, [(0, 4999), (20_000, 5000), (40_000, 5001), (70_000, 30_000), (115_000, 5000)])),
            ("r1", with_runs(90_000, [(0, 6000), (50_000, 5001), (84_000, 6000)])),      # long runs at both ends
            ("r2", np.full(7000, ord("N"), dtype=np.uint8)),                              # nothing but a long run
            ("r3", np.zeros(0, dtype=np.uint8)),
            ("r4", with_runs(60_000, [(59_000, 1000)])), ("r5", with_runs(60_000, [(0, 4500)])),  # 5500 N across a boundary
            ("r6", _small_genome(seed=3, lens=(150_000,))[0][1])]
    want = prep.prepare_records(recs, skip_masked=skip_masked)
    got, idx = prep.prepare_records_gpu(recs, skip_masked=skip_masked)
    try:
        assert np.array_equal(got.data, want.data)
        assert got.chunks == want.chunks
        assert [(m.name, m.position, m.length) for m in got.map] == [(m.name, m.position, m.length) for m in want.map]
        # the oracle's literal statement, record by record
        off = 0
        lit = []
        for _, seq in recs:
            t_ = oracle.normalise(seq, skip_masked)
            assert np.array_equal(t_, want.data[off:off + len(seq)])
            lit.extend((off + a, ln) for a, ln in oracle.find_chunks(t_))
            off += len(seq)
        assert lit == got.chunks
        st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True, skip_masked=skip_masked)
        with asgart_amd.Index(want.data, None) as ref:
            exp = ref.search_duplications_raw(want.chunks, st)
            assert np.array_equal(ref.sa_read(0, len(want.data)), idx.sa_read(0, len(want.data)))
        res = idx.search_duplications_raw(got.chunks, st)
        assert np.array_equal(res[0], exp[0]) and np.array_equal(res[1], exp[1])
    finally:
        idx.close()
    # the strand may stay on the device; too small a chunk array is reported with the room needed
    got2, idx2 = prep.prepare_records_gpu(recs, skip_masked=skip_masked, want_text=False, want_index=False)
    assert got2.data is None and idx2 is None and got2.chunks == want.chunks
