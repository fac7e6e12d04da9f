"""Host-side input preparation (asgart_amd.prep) against the oracle's restatement of
prepare_data (reference src/bin/asgart.rs:273-471)."""
import numpy as np
import pytest

import bruteforce as B
import oracle
from asgart_amd import prep, synth


@pytest.mark.parametrize("skip_masked", [False, True])
def test_normalise_matches_oracle(skip_masked):
    seq = np.arange(256, dtype=np.uint8).repeat(3)
    assert np.array_equal(prep.normalise(seq, skip_masked), oracle.normalise(seq.copy(), skip_masked))


@pytest.mark.parametrize("seed", range(6))
def test_find_chunks_matches_oracle_and_bruteforce(seed):
    rng = np.random.default_rng(seed)
    n = 60_000
    s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    for _ in range(6):
        ln = int(rng.choice([1, 100, 4999, 5000, 5001, 9000]))
        a = int(rng.integers(0, n - ln))
        s[a:a + ln] = ord("N")
    if seed % 2:
        s[:5500] = ord("N")
        s[-7000:] = ord("n")
    got = prep.find_chunks_to_process(s)
    assert got == oracle.find_chunks(s) == B.find_chunks(bytes(s))


def test_find_chunks_degenerate():
    for s in (b"", b"N" * 6000, b"N" * 10, b"ACGT", b"N" * 6000 + b"A"):
        a = np.frombuffer(s, dtype=np.uint8)
        assert prep.find_chunks_to_process(a) == oracle.find_chunks(a) == B.find_chunks(s)


def test_prepare_records_layout():
    recs = synth.make_genome([40_000, 30_000], seed=3, sd_per_mb=30, sd_len=(1000, 3000))
    pr = prep.prepare_records(recs, skip_masked=True)
    assert pr.data[-1] == ord("$") and len(pr.data) == 70_001
    assert set(np.unique(pr.data[:-1])) <= set(b"ACGTN")
    assert [s.position for s in pr.map] == [0, 40_000]
    for start, ln in pr.chunks:                      # chunks never straddle records
        assert (start < 40_000) == (start + ln <= 40_000)


def test_fasta_reader(tmp_path):
    p = tmp_path / "x.fa"
    p.write_bytes(b">chrA desc\nACGT\nacgtnn\n>chrB\nGG\n")
    recs = list(prep.read_records(str(p)))
    assert [r[0] for r in recs] == ["chrA", "chrB"]
    assert bytes(recs[0][1]) == b"ACGTacgtnn" and bytes(recs[1][1]) == b"GG"
