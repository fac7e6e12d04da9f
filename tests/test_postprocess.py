"""N1/N2 of SURVEY.md section 8f: the host post-processing chain and the RunResult surface against
the oracle's C restatement (oracle/postprocess.c) on the families of seeded synthetic genomes."""
import json

import numpy as np
import pytest

import oracle
import asgart_amd
from asgart_amd import postprocess, prep, synth


def _to_families(offs, sds, reverse=False, complement=False):
    return [[asgart_amd.ProtoSD(int(r[0]), int(r[1]), int(r[2]), int(r[3]), 0.0, reverse, complement)
             for r in sds[int(offs[f]):int(offs[f + 1])]] for f in range(len(offs) - 1)]


def _case(seed, **kw):
    args = dict(sd_per_mb=60, sd_len=(1000, 6000), alu_frac=0.05, l1_frac=0.02, sat_per_record=1,
                sat_copies=(20, 80), short_n_per_mb=20)
    args.update(kw)
    recs = synth.make_genome([220_000, 130_000], seed=seed, **args)
    pr = prep.prepare_records(recs)
    return pr, oracle.Index.build(pr.data)


@pytest.mark.parametrize("seed,mode,min_len", [(31, (False, False), 300), (32, (True, True), 300),
                                               (33, (False, False), 1000), (34, (False, False), 200)])
def test_chain_matches_oracle(seed, mode, min_len):
    pr, oidx = _case(seed)
    offs, sds = oidx.run_raw(pr.chunks, oracle.make_settings(min_length=min_len, reverse=mode[0], complement=mode[1]))
    assert len(sds) > 3
    eoffs, esds = oracle.postprocess(pr.data, offs, sds)
    strand = asgart_amd.Strand("x.fa", pr.data, pr.map)
    fams = postprocess.post_process(_to_families(offs, sds, *mode), strand)
    got = [[sd.as_tuple() for sd in fam] for fam in fams]
    assert got == oracle.families_to_list(eoffs, esds)
    assert all(sd.reversed == mode[0] and sd.complemented == mode[1] for fam in fams for sd in fam)


def test_filter_ns_inclusive_range_and_threshold():
    text = np.frombuffer(b"A" * 100 + b"N" * 21 + b"A" * 200 + b"$", dtype=np.uint8)
    strand = asgart_amd.Strand("t", text, [prep.Start("c", 0, 320)])
    # left arm [90 ..= 190] holds 21 N over length 100 -> 0.21 > 0.2 : dropped
    # left arm [91 ..= 196] holds 21 N over length 105 -> exactly 0.2 in f32 : kept (<=)
    fams = [[asgart_amd.ProtoSD(90, 200, 100, 100)], [asgart_amd.ProtoSD(91, 200, 105, 100)]]
    out = postprocess.FilterNs().run(fams, strand)
    assert [[sd.as_tuple() for sd in f] for f in out] == [[(91, 200, 105, 100)]]
    offs = np.array([0, 1, 2], dtype=np.uint64)
    sds = np.array([[90, 200, 100, 100], [91, 200, 105, 100]], dtype=np.uint64)
    eo, es = oracle.postprocess(text, offs, sds)
    assert oracle.families_to_list(eo, es) == [[(91, 200, 105, 100)]]


def test_reorder_keeps_lengths_and_merge_quirk():
    sd = asgart_amd.ProtoSD(500, 100, 30, 40)
    postprocess.ReOrder().run([[sd]], None)
    assert sd.as_tuple() == (100, 500, 30, 40)
    x = asgart_amd.ProtoSD(10, 1000, 50, 70)
    y = asgart_amd.ProtoSD(40, 1030, 60, 80)
    z = postprocess._merge(x, y)        # x uses left_length (50), y right_length (80) on both arms
    assert z.as_tuple() == (10, 1000, max(10 + 50, 40 + 80) - 10, max(1000 + 50, 1030 + 80) - 1000)


def test_run_result_json_surface(tmp_path):
    pr, oidx = _case(35)
    settings = asgart_amd.RunSettings.from_cli(min_length=300, reverse=True, complement=True)
    offs, sds = oidx.run_raw(pr.chunks, oracle.make_settings(min_length=300, reverse=True, complement=True))
    strand = asgart_amd.Strand("a.fa, b.fa", pr.data, pr.map)
    fams = postprocess.post_process(_to_families(offs, sds, True, True), strand)
    res = postprocess.run_result(fams, strand, settings)
    txt = postprocess.to_json(res)
    back = json.loads(txt)
    assert list(back.keys()) == ["strand", "settings", "families"]
    assert list(back["settings"].keys()) == ["probe_size", "max_gap_size", "min_duplication_length",
                                             "max_cardinality", "trim", "skip_masked"]
    sd0 = back["families"][0][0]
    assert list(sd0.keys()) == ["chr_left", "chr_right", "global_left_position", "global_right_position",
                                "chr_left_position", "chr_right_position", "left_length", "right_length",
                                "left_seq", "right_seq", "identity", "reversed", "complemented"]
    assert sd0["identity"] == 0.0 and '"identity": 0.0' in txt and '"trim": null' in txt
    assert sd0["reversed"] is True and sd0["left_seq"] is None
    for fam in back["families"]:
        for sd in fam:
            c = next(m for m in back["strand"]["map"] if m["name"] == sd["chr_left"])
            assert sd["chr_left_position"] == sd["global_left_position"] - c["position"]
    assert back["strand"]["length"] == len(pr.data) - 1
    assert postprocess.out_filename(["/x/a.fa", "b.fasta"], settings) == "a-b_RC.json"


def test_json_text_is_serde_exact_for_f32_identity():
    """`identity` is an f32 (src/structs.rs:485); serde_json prints the shortest decimal that reads
    back as the same f32 (ryu), always with a fraction: 98.7, not 98.69999694824219.  Full text of a
    small RunResult, laid out as serde_json::to_string_pretty lays it out (src/exporters.rs:12-25)."""
    strand = asgart_amd.Strand("t.fa", np.frombuffer(b"ACGT" * 10 + b"$", dtype=np.uint8),
                               [postprocess.Start("chr\u00e9 1", 0, 30), postprocess.Start("c2", 30, 10)])
    settings = asgart_amd.RunSettings.from_cli(reverse=True)
    fams = [[asgart_amd.ProtoSD(2, 31, 5, 6, float(np.float32(98.7)), True, False)],
            [asgart_amd.ProtoSD(1, 100, 3, 3, float(np.float32(100.0)), True, False),
             asgart_amd.ProtoSD(4, 20, 3, 3, float(np.float32(100.0 * (1.0 - 1.0 / 3.0))), True, False)]]
    txt = postprocess.to_json(postprocess.run_result(fams, strand, settings))
    want = """{
  "strand": {
    "name": "t.fa",
    "length": 40,
    "map": [
      {
        "name": "chr\u00e9 1",
        "position": 0,
        "length": 30
      },
      {
        "name": "c2",
        "position": 30,
        "length": 10
      }
    ]
  },
  "settings": {
    "probe_size": 20,
    "max_gap_size": 120,
    "min_duplication_length": 1000,
    "max_cardinality": 500,
    "trim": null,
    "skip_masked": false
  },
  "families": [
    [
      {
        "chr_left": "chr\u00e9 1",
        "chr_right": "c2",
        "global_left_position": 2,
        "global_right_position": 31,
        "chr_left_position": 2,
        "chr_right_position": 1,
        "left_length": 5,
        "right_length": 6,
        "left_seq": null,
        "right_seq": null,
        "identity": 98.7,
        "reversed": true,
        "complemented": false
      }
    ],
    [
      {
        "chr_left": "chr\u00e9 1",
        "chr_right": "unknown",
        "global_left_position": 1,
        "global_right_position": 100,
        "chr_left_position": 1,
        "chr_right_position": 100,
        "left_length": 3,
        "right_length": 3,
        "left_seq": null,
        "right_seq": null,
        "identity": 100.0,
        "reversed": true,
        "complemented": false
      },
      {
        "chr_left": "chr\u00e9 1",
        "chr_right": "chr\u00e9 1",
        "global_left_position": 4,
        "global_right_position": 20,
        "chr_left_position": 4,
        "chr_right_position": 20,
        "left_length": 3,
        "right_length": 3,
        "left_seq": null,
        "right_seq": null,
        "identity": 66.666664,
        "reversed": true,
        "complemented": false
      }
    ]
  ]
}"""
    assert txt == want
    assert json.loads(txt)["families"][0][0]["identity"] == 98.7
    # the printer itself: ryu's f32 rules (plain decimals for 1e-6 <= |x| < 1e13, else d.ddde[-]N)
    for v, text in ((0.0, "0.0"), (97.3, "97.3"), (98.69999694824219, "98.7"), (1e-7, "1e-7"),
                    (1.5e-7, "1.5e-7"), (1e-6, "0.000001"), (1 / 3, "0.33333334"), (1e13, "1e13"),
                    (123456.0, "123456.0"), (99.99999, "99.99999"), (float("nan"), "null")):
        assert postprocess.f32_repr(v) == text, (v, postprocess.f32_repr(v))
        if text != "null":
            assert np.float32(float(text)) == np.float32(v)


@pytest.mark.parametrize("seed", range(6))
def test_chain_random_families(seed):
    """Arbitrary (not search-produced) families: every branch of _reduce, swapped arms, N-rich arms."""
    rng = np.random.default_rng(900 + seed)
    n = 20_000
    text = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n + 1)
    for _ in range(12):
        s = int(rng.integers(0, n - 600)); text[s:s + int(rng.integers(20, 500))] = ord("N")
    text[n] = ord("$")
    offs, rows = [0], []
    for _ in range(40):
        base_l, base_r = int(rng.integers(0, n - 3000)), int(rng.integers(0, n - 3000))
        for _ in range(int(rng.integers(1, 9))):
            l = base_l + int(rng.integers(0, 1200)); r = base_r + int(rng.integers(0, 1200))
            rows.append((l, r, int(rng.integers(50, 900)), int(rng.integers(50, 900))))
        offs.append(len(rows))
    offs = np.array(offs, dtype=np.uint64); sds = np.array(rows, dtype=np.uint64)
    eo, es = oracle.postprocess(text, offs, sds)
    strand = asgart_amd.Strand("r", text, [prep.Start("c", 0, n)])
    fams = postprocess.post_process(_to_families(offs, sds), strand)
    assert [[sd.as_tuple() for sd in f] for f in fams] == oracle.families_to_list(eo, es)
    assert len(es) < len(sds)


def _py_levenshtein(a: bytes, b: bytes) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j - 1] + (ca != cb), prev[j] + 1, cur[j - 1] + 1))
        prev = cur
    return prev[-1]


def test_oracle_levenshtein_identity_against_plain_python():
    """The oracle's ComputeScore restatement vs a textbook DP, incl. inclusive ranges, reverse and
    complement of the right arm (src/structs.rs:439-452)."""
    rng = np.random.default_rng(77)
    text = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=4000, p=[.24, .24, .24, .24, .04])
    text = np.concatenate([text, np.frombuffer(b"$", dtype=np.uint8)])
    tr = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    for _ in range(40):
        ll, rl = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        left, right = int(rng.integers(0, 3800)), int(rng.integers(0, 3800))
        for rev, comp in ((False, False), (True, False), (False, True), (True, True)):
            a = text[left:left + ll + 1].tobytes()
            b = text[right:right + rl + 1].tobytes()
            if rev:
                b = b[::-1]
            if comp:
                b = b.translate(tr)
            want = np.float32(100.0 * (1.0 - _py_levenshtein(a, b) / max(ll, rl)))
            got = oracle.levenshtein_identity(text, (left, right, ll, rl), rev, comp)
            assert got == want, (left, right, ll, rl, rev, comp)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,mode,min_len", [(31, (False, False), 300), (32, (True, True), 300), (34, (False, False), 200),
                                               (35, (True, True), 1000)])
def test_native_chain_matches_oracle_and_python(seed, mode, min_len, hiplib):
    """asgart_post_process (N counts on the GPU, reduction on host threads) against the oracle's C chain and the Python
    statement of the same steps, on the families the HIP search itself returns; the array JSON writer against the
    per-object one."""
    pr, oidx = _case(seed, short_n_per_mb=60)
    st = asgart_amd.RunSettings.from_cli(min_length=min_len, reverse=mode[0], complement=mode[1])
    with asgart_amd.Index(pr.data, oidx.sa) as idx:
        offs, sds = idx.search_duplications_raw(pr.chunks, st)
        assert len(sds) > 3
        for threads in (1, 0):
            go, gs = postprocess.post_process_arrays(idx, offs, sds, threads)
            eoffs, esds = oracle.postprocess(pr.data, offs, sds)
            assert np.array_equal(go, eoffs) and np.array_equal(gs, esds)
        strand = asgart_amd.Strand("x.fa", pr.data, pr.map)
        fams = postprocess.post_process(_to_families(offs, sds, *mode), strand)
        assert [[sd.as_tuple() for sd in fam] for fam in fams] == oracle.families_to_list(go, gs)
        assert postprocess.to_json_arrays(go, gs, strand, st) == postprocess.to_json(postprocess.run_result(fams, strand, st))
        # an arm whose inclusive range leaves the text is an error (the reference panics), not a silent clamp
        bad = np.array([[0, len(pr.data) - 5, 10, 5]], dtype=np.uint64)
        with pytest.raises(asgart_amd.AsgartError):
            idx.post_process(np.array([0, 1], dtype=np.uint64), bad)
        # nothing in, nothing out
        eo, es = idx.post_process(np.zeros(1, np.uint64), np.zeros((0, 4), np.uint64))
        assert len(eo) == 1 and len(es) == 0


@pytest.mark.gpu
def test_native_chain_large_family(hiplib):
    """A tandem array gives one family of thousands of duplications: the quadratic reduction on it, natively, equals
    the oracle's."""
    rng = np.random.default_rng(3)
    mono = rng.integers(0, 4, size=171)
    arr = np.tile(mono, 700)
    mut = rng.random(arr.shape) < 0.03
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    g = np.concatenate([rng.integers(0, 4, size=20_000), arr, rng.integers(0, 4, size=20_000)])
    text = np.concatenate([np.frombuffer(b"ACGT", dtype=np.uint8)[g], np.frombuffer(b"$", dtype=np.uint8)])
    chunks = [(0, len(text) - 1)]
    st = asgart_amd.RunSettings.from_cli()
    with asgart_amd.Index(text, None) as idx:
        offs, sds = idx.search_duplications_raw(chunks, st)
        assert np.diff(offs.astype(np.int64)).max() > 500
        go, gs = idx.post_process(offs, sds)
    eoffs, esds = oracle.postprocess(text, offs, sds)
    assert np.array_equal(go, eoffs) and np.array_equal(gs, esds)
