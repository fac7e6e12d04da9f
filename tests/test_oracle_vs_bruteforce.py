"""C oracle == independent pure-Python restatement on seeded random inputs."""
import random

import numpy as np
import pytest

import bruteforce as B
import oracle


def _rand_text(rng, n, planted=True):
    t = bytearray(rng.choice(b"ACGT") for _ in range(n))
    if planted and n > 600:
        for _ in range(rng.randint(1, 4)):
            ln = rng.randint(150, min(600, n // 3))
            a = rng.randrange(0, n - ln)
            b = rng.randrange(0, n - ln)
            seg = bytes(t[a:a + ln])
            if rng.random() < 0.5:
                seg = B.revcomp(seg)
            seg = bytearray(seg)
            for j in range(len(seg)):
                if rng.random() < 0.02:
                    seg[j] = rng.choice(b"ACGT")
            t[b:b + ln] = seg
        for _ in range(rng.randint(0, 3)):
            a = rng.randrange(0, n - 40)
            t[a:a + rng.randint(1, 40)] = b"N" * 40
    return bytes(t[:n])


@pytest.mark.parametrize("seed", range(12))
def test_sa_cache_search(seed):
    rng = random.Random(seed)
    text = _rand_text(rng, rng.randint(50, 3000)) + b"$"
    sa = B.suffix_array(text)
    idx = oracle.Index.build(text)
    assert list(idx.sa) == sa
    for _ in range(40):
        p = rng.randrange(0, len(text) - 9)
        p8 = text[p:p + 8]
        if b"$" in p8:
            continue
        lo, hi = idx.cache_get(p8)
        elo, ehi = B.cache_entry(text, sa, p8)
        assert hi - lo == ehi - elo and (hi == lo or (lo, hi) == (elo, ehi))
    for _ in range(60):
        k = rng.choice([8, 12, 20, 21, 22, 30, 42])
        p = rng.randrange(0, max(1, len(text) - k - 1))
        pat = text[p:p + k]
        if len(pat) < k or b"$" in pat[:8]:
            continue
        got, _ = idx.search(pat)
        assert list(got) == B.search(text, sa, pat)


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("mode", [(False, False), (True, False), (False, True), (True, True)])
def test_run_matches_bruteforce(seed, mode):
    rng = random.Random(1000 + seed)
    n = rng.randint(700, 2500)
    text = _rand_text(rng, n)
    strand = text + b"$"
    chunks = B.find_chunks(text)
    k = rng.choice([10, 12, 16, 20])
    gap = rng.choice([0, 20, 100])
    min_len = rng.choice([50, 100, 200])
    card = rng.choice([1, 3, 500])
    sa = B.suffix_array(strand)
    exp = B.run(strand, sa, chunks, k=k, gap=gap, min_len=min_len, max_card=card,
                reverse=mode[0], complement=mode[1])
    idx = oracle.Index.build(strand)
    st = oracle.make_settings(k=k, gap=gap, min_length=min_len, max_cardinality=card,
                              reverse=mode[0], complement=mode[1])
    assert idx.run(chunks, st) == exp


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("mode", [(False, False), (True, True)])
def test_run_with_long_probes_matches_bruteforce(seed, mode):
    """probe sizes above 21 (one 63-bit key word of the HIP path): the oracle itself is pinned for them"""
    rng = random.Random(3000 + seed)
    n = rng.randint(1200, 3000)
    text = _rand_text(rng, n)
    strand = text + b"$"
    chunks = B.find_chunks(text)
    k = rng.choice([22, 25, 31, 42])
    gap = rng.choice([0, 40, 100])
    sa = B.suffix_array(strand)
    exp = B.run(strand, sa, chunks, k=k, gap=gap, min_len=60, max_card=500, reverse=mode[0], complement=mode[1])
    idx = oracle.Index.build(strand)
    st = oracle.make_settings(k=k, gap=gap, min_length=60, max_cardinality=500, reverse=mode[0], complement=mode[1])
    assert idx.run(chunks, st) == exp


@pytest.mark.parametrize("seed", range(16))
def test_trim_matches_bruteforce(seed):
    """`--trim START END` (src/bin/asgart.rs:142-148): the suffix array covers data[start..end]+'$' only and
    is compared through the whole text, so the suffixes that end within 8 (cache) / k (search) bases of `end`
    are out of place for the bisections.  Both restatements follow the same published routines
    (libdivsufsort `sa_search`, superslice `equal_range_by`) step by step; random windows, including windows
    that end inside a planted copy or at the end of the text."""
    rng = random.Random(4000 + seed)
    n = rng.randint(900, 2600)
    text = _rand_text(rng, n)
    strand = text + b"$"
    chunks = B.find_chunks(text)
    a = rng.randrange(0, n - 50)
    b = rng.choice([n, n + 5, rng.randrange(a + 30, n + 1)])
    from asgart_amd.prep import validate_trim
    trim = validate_trim((a, b), len(strand))
    assert trim is not None and trim[1] <= len(strand) - 1
    k = rng.choice([10, 12, 20])
    mode = rng.choice([(False, False), (True, True), (True, False), (False, True)])
    sa = B.trim_suffix_array(strand, *trim)
    idx = oracle.Index.build_trim(strand, *trim)
    assert list(idx.sa) == sa and len(sa) == trim[1] - trim[0] + 1
    for _ in range(60):   # cache entries and searches, the two bisections that see the misplaced suffixes
        p = rng.randrange(0, len(text) - 21)
        pat = text[p:p + k]
        if b"N" in pat:
            continue
        left, count = B.sa_search_published(strand, pat[:8], sa, 0, len(sa))
        lo, hi = idx.cache_get(pat[:8])
        assert (hi - lo == count) and (count == 0 or lo == left)
        got, _ = idx.search(pat)
        assert list(got) == B.search(strand, sa, pat)
    exp = B.run(strand, sa, chunks, k=k, gap=50, min_len=100, max_card=500, reverse=mode[0], complement=mode[1])
    st = oracle.make_settings(k=k, gap=50, min_length=100, max_cardinality=500, reverse=mode[0], complement=mode[1])
    assert idx.run(chunks, st) == exp


def test_trim_validation():
    from asgart_amd.prep import validate_trim
    assert validate_trim(None, 100) is None
    assert validate_trim((10, 50), 100) == (10, 50)
    assert validate_trim((10, 100), 100) == (10, 99)     # stop >= len: clamped to the '$' (asgart.rs:437-446)
    assert validate_trim((10, 500), 100) == (10, 99)
    assert validate_trim((50, 50), 100) is None          # stop <= shift (:448-453)
    assert validate_trim((60, 50), 100) is None
    assert validate_trim((100, 200), 100) is None        # clamped stop 99 <= shift (:448)


def test_tail_corner_bisection_is_emulated():
    """Text whose last bases share an 8-mer with a probe: the comparator of
    src/searcher.rs:164-170 calls the short tail suffixes Less.  Both restatements follow the same
    (recalled) superslice bisection; this pins them to each other on that corner."""
    rng = random.Random(5)
    core = bytes(rng.choice(b"ACGT") for _ in range(1500))
    motif = b"ACGTTGCAAC"                     # 10 bases: the text ends with it
    for fill in (b"A", b"T", b"G"):
        text = core[:400] + motif + fill * 10 + core[400:900] + motif + b"T" * 10 + core[900:] + motif
        strand = text + b"$"
        sa = B.suffix_array(strand)
        idx = oracle.Index.build(strand)
        for pat in (motif + fill * 10, motif + b"T" * 10, motif + b"A" * 10, motif + b"C" * 10):
            got, _ = idx.search(pat)
            assert list(got) == B.search(strand, sa, pat, exact_bisection=True)
        chunks = [(0, len(text))]
        exp = B.run(strand, sa, chunks, min_len=100)
        assert idx.run(chunks, oracle.make_settings(min_length=100)) == exp


def test_short_and_empty_needles():
    text = b"ACGT" * 10 + b"$"
    idx = oracle.Index.build(text)
    assert idx.run([(0, 40)], oracle.make_settings(min_length=10)) == []      # L < k + step... few probes
    assert idx.run([(0, 20)], oracle.make_settings(min_length=1000)) == []    # L < M
    assert idx.run([], oracle.make_settings()) == []
