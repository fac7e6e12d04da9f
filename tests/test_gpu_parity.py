"""GPU parity tests: the HIP path (through the C ABI) must be bit-exact against the
CPU oracle on the same seeded inputs.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest

import oracle
import asgart_amd
from asgart_amd import prep, synth

pytestmark = pytest.mark.gpu

MODES = [(False, False), (True, False), (False, True), (True, True)]


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
