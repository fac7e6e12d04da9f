"""The RANGES scheme of DESIGN.md 4.8, restated on the CPU (no GPU, no HIP code): a long segment of the automaton is cut
at hit-probes; every range runs from an EMPTY arm list `warm` probes in front of its cut, reports from the cut on, and is
accepted only if what it holds when it reaches its cut equals what the range in front of it holds when it stops there
(every arm, every field, by creation key; family open or not).  Where the first cut fails, the ranges in front of it stand
and ONE more run covers the rest from a checked state.  The joined output -- family ordinals counted on from the flushes
of the ranges before, creation order by (probe, hit) -- must be the whole run's, which is itself pinned against the
oracle here.

What the test restates (plain Python over the oracle's per-probe hit rows, live arms only):
    src/automaton.rs:119-171   one processed probe: the first accepting arm in list order takes a hit, the last hit of
                               an arm wins, unmatched hits start arms in hit order, unextended arms age by `step`
    src/automaton.rs:173-200   dead arms retire (reported when len(right) >= M), the family is flushed when no arm is left
The kernels do the same per range (extend_k8_kernel<RANGE>), with lazily applied ages and waves a step apart -- which is
why their cuts sit at hit-probes; this model ages eagerly, so any probe would do, but the cuts are placed the same way.
"""
import numpy as np
import pytest

import oracle

K, GAP, STEP = 20, 100, 10
G = GAP + K
M = 200
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _text(seed, period, copies, sub):
    rng = np.random.default_rng(seed)
    arr = np.tile(rng.integers(0, 4, size=period), copies)
    mut = rng.random(arr.shape) < sub
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    g = np.concatenate([rng.integers(0, 4, size=3000), arr, rng.integers(0, 4, size=3000)])
    return np.concatenate([BASES[g], np.frombuffer(b"$", dtype=np.uint8)])


class Run:
    """the automaton over probes [t0, t1) from an empty arm list; records from probe `emit_from` on"""

    def __init__(self, status, offs, hits, t0, t1, emit_from, snap_at=None):
        self.arms = []          # [key, ls, le, rs, re, gap] in creation order; key = (probe, index of the creating hit)
        self.open = False
        self.flushes = 0        # flushes at probes >= emit_from
        self.recs = []          # (family ordinal counted from emit_from, key, ls, le, rs, re)
        self.at_cut = None
        for t in range(t0, t1):
            if t == snap_at:
                self.at_cut = self.state()
            if status[t]:
                continue        # skipped probes neither age nor reset (:100-102, :115-117)
            self.probe(t, (t + 1) * STEP, hits[offs[t]:offs[t + 1]], t >= emit_from)
        self.end = self.state()

    def state(self):
        return (sorted(tuple(a) for a in self.arms), self.open)

    def probe(self, t, i, xs, emit):
        won = {}
        new = []
        for h, x in enumerate(xs):
            for n, a in enumerate(self.arms):           # first accepting arm in list order (:67-78)
                thr = max(G, (a[2] - a[1]) // 10)
                if a[4] - K < x < a[4] + thr:           # the whole predicate
                    won[n] = x                          # the LAST hit of an arm wins (:136-143)
                    break
            else:
                new.append((h, int(x)))
        for n, a in enumerate(self.arms):
            if n in won:
                a[2], a[4], a[5] = i + K, int(won[n]) + K, 0
            else:
                a[5] += STEP                            # (:166-171)
        for h, x in new:                                # NewArm in hit order (:145-163); they age at once
            self.arms.append([(t, h), i, i + K, x, x + K, STEP])
            self.open = True
        alive = []
        for a in self.arms:
            if a[5] < G:
                alive.append(a)
            elif a[4] - a[3] >= M and emit:             # retired: reported when long enough (:186-196)
                self.recs.append((self.flushes, a[0], a[1], a[2], a[3], a[4]))
        self.arms = alive
        if self.open and not self.arms:                 # the family is flushed when no arm is left (:182-200)
            self.open = False
            if emit:
                self.flushes += 1


def _families(recs):
    fams = {}
    for f, key, ls, le, rs, re in sorted(recs):
        fams.setdefault(f, []).append((ls, le - ls, rs, re - rs))
    return [fams[f] for f in sorted(fams)]


def _ranges(status, offs, hits, s0, s1, last_hit_probe, n_ranges, warm):
    """-> (joined records, cuts that held, number of runs)"""
    cnt = np.diff(offs)
    cuts = []
    for j in range(1, n_ranges):
        c = s0 + (last_hit_probe - s0) * j // n_ranges
        while status[c] or cnt[c] == 0:                 # moved forward to a hit-probe
            c += 1
        cuts.append(c)
    starts = [s0] + cuts
    stops = cuts + [s1]
    begin = [s0] + [max(s0, c - warm) for c in cuts]
    runs = [Run(status, offs, hits, begin[j], stops[j], starts[j], snap_at=starts[j] if j else None) for j in range(n_ranges)]
    held = 0
    while held < n_ranges - 1 and runs[held].end == runs[held + 1].at_cut:
        held += 1
    n_runs = n_ranges
    kept = runs[:held + 1]
    if held < n_ranges - 1:                             # the rest as ONE more run, from where the last good range started
        kept = runs[:held + 1] + [Run(status, offs, hits, begin[held], s1, stops[held])]
        n_runs += 1
    out, base = [], 0
    for r in kept:
        out += [(base + f,) + tuple(rest) for (f, *rest) in r.recs]
        base += r.flushes
    return out, held, n_runs


@pytest.mark.parametrize("period,copies,sub,n_ranges,warm,expect", [
    (64, 150, 0.06, 5, 400, "all"),     # arms die of the substitutions within the warm-up: every cut holds
    (64, 150, 0.06, 5, 8, "some"),      # a warm-up of 8 probes: cuts fail, the rest runs from the last good one
    (171, 60, 0.0, 4, 200, "some"),     # exact copies: one arm per offset lives through the whole array
])
def test_ranges_joined_equal_the_whole_run_and_the_oracle(period, copies, sub, n_ranges, warm, expect):
    text = _text(11, period, copies, sub)
    oidx = oracle.Index.build(text)
    st = oracle.make_settings(k=K, gap=GAP, min_length=M)
    chunk = (0, len(text) - 1)
    status, offs, hits = oidx.probe_hits(oracle.prepare_needle(text, chunk, st), 0, st)
    offs, hits = offs.astype(np.int64), hits.astype(np.int64)
    hp = np.nonzero((status == 0) & (np.diff(offs) > 0))[0]
    # the segment: from its first hit-probe to the quiet probes behind its last one that let every arm die
    s0, s1 = int(hp[0]), min(len(status), int(hp[-1]) + 1 + 2 * ((G + STEP - 1) // STEP))
    whole = Run(status, offs, hits, s0, s1, s0)
    assert not whole.arms                               # (the flank is unique: nothing alive at the end of the chunk)
    want = _families(whole.recs)
    # the restatement itself against the oracle's run over the same chunk
    eo, es = oidx.run_raw([chunk], st)
    got = [[(int(a), int(b), int(c), int(d)) for a, c, b, d in es[eo[f]:eo[f + 1]]] for f in range(len(eo) - 1)]
    assert got == want and len(want) > 0
    joined, held, n_runs = _ranges(status, offs, hits, s0, s1, int(hp[-1]), n_ranges, warm)
    assert _families(joined) == want
    if expect == "all":
        assert held == n_ranges - 1 and n_runs == n_ranges
    else:
        assert held < n_ranges - 1 and n_runs == n_ranges + 1


@pytest.mark.parametrize("period,copies,sub,n_ranges,warm", [
    (64, 150, 0.06, 5, 8),
    (64, 150, 0.06, 8, 16),
    (171, 60, 0.0, 4, 200),
])
def test_a_failed_cut_tells_the_warm_up_its_segment_needs(period, copies, sub, n_ranges, warm):
    """What the library does with a cut that did not hold (pipeline.hip: remember(); validate_cuts_kernel reports where the
    oldest arm in front of the cut was born).  A run that is to hold that arm at the cut must START in front of the probe
    that created it -- creation keys are (probe, hit) in every run -- so a warm-up shorter than that distance cannot hold:
    the next call gives the segment's ranges that much.  That is necessary, not sufficient (what an arm looks like also
    depends on arms that died before the cut): a cut that fails although every arm was born inside the warm-up gets the
    longest warm-up a range is worth -- two ranges -- and beyond that only the cuts that held are planned again.  Here: the
    failures of a noisy array are of the first kind at short warm-ups and go away within two ranges of warm-up; in an array
    of exact copies the cuts do not hold even then, although every arm at a cut was born well inside the warm-up (an arm's
    left end and threshold carry the history of the arms before it): the library ends at "only the cuts that held"."""
    text = _text(11, period, copies, sub)
    oidx = oracle.Index.build(text)
    st = oracle.make_settings(k=K, gap=GAP, min_length=M)
    chunk = (0, len(text) - 1)
    status, offs, hits = oidx.probe_hits(oracle.prepare_needle(text, chunk, st), 0, st)
    offs, hits = offs.astype(np.int64), hits.astype(np.int64)
    hp = np.nonzero((status == 0) & (np.diff(offs) > 0))[0]
    s0, s1 = int(hp[0]), min(len(status), int(hp[-1]) + 1 + 2 * ((G + STEP - 1) // STEP))
    whole = Run(status, offs, hits, s0, s1, s0)
    want = _families(whole.recs)
    cnt = np.diff(offs)
    born_outside = born_inside = 0
    needs = []
    for j in range(1, n_ranges):
        c = s0 + (int(hp[-1]) - s0) * j // n_ranges
        while status[c] or cnt[c] == 0:
            c += 1
        true = Run(status, offs, hits, s0, c, s0).end          # what the segment really holds in front of cut c
        cold = Run(status, offs, hits, max(s0, c - warm), c, c).end
        if cold == true:
            continue
        assert true[0], (j, c)                                   # (a cut in front of which nothing lives holds from anywhere)
        born = min(a[0][0] for a in true[0])
        needs.append(c - born)
        if c - born > warm:
            born_outside += 1                                    # the warm-up started behind that birth: it could not hold
            assert not any(a[0][0] == born for a in cold[0])
        else:
            born_inside += 1
    assert needs                                                # (the cases are chosen so that cuts fail)
    span = int(hp[-1]) - s0
    limit = 2 * (span // n_ranges)                              # the longest warm-up a range is worth
    joined, held, n_runs = _ranges(status, offs, hits, s0, s1, int(hp[-1]), n_ranges, limit)
    assert _families(joined) == want
    if sub > 0:
        assert born_outside > 0 and max(needs) < limit and held == n_ranges - 1, (needs, held)
    else:
        # every arm in front of a cut was born well inside two ranges of warm-up -- and still the cuts do not hold
        assert max(needs) < limit and held < n_ranges - 1, (needs, held)


def _plan(cuts, skipped):
    return [c for j, c in enumerate(cuts) if j not in skipped]


def _run_plan(status, offs, hits, s0, s1, kept, warm):
    """the ranges of one call over the kept cuts -> (runs, index of the first kept cut that does not hold or None)"""
    starts = [s0] + kept
    stops = kept + [s1]
    begin = [s0] + [max(s0, c - warm) for c in kept]
    runs = [Run(status, offs, hits, begin[j], stops[j], starts[j], snap_at=starts[j] if j else None) for j in range(len(starts))]
    for j in range(len(kept)):
        if runs[j].end != runs[j + 1].at_cut:
            return runs, j
    return runs, None


def _text_two_arrays(seed):
    """an array of exact copies (no cut inside it holds) and, right behind it, a noisy one (its cuts hold): ONE segment"""
    rng = np.random.default_rng(seed)
    a = np.tile(rng.integers(0, 4, size=171), 25)
    b = np.tile(rng.integers(0, 4, size=64), 120)
    mut = rng.random(b.shape) < 0.06
    b[mut] = (b[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    g = np.concatenate([rng.integers(0, 4, size=3000), a, b, rng.integers(0, 4, size=3000)])
    return np.concatenate([BASES[g], np.frombuffer(b"$", dtype=np.uint8)])


@pytest.mark.parametrize("period,copies,sub,n_ranges,warm", [
    (64, 150, 0.06, 12, 8),       # a noisy array, a warm-up far too short: most cuts fail
    (0, 0, 0.0, 10, 300),         # exact copies, then a noisy array: the cuts inside the first fail at any warm-up
])
def test_a_set_of_failed_cuts_instead_of_a_prefix(period, copies, sub, n_ranges, warm):
    """The next step of the scheme, restated before it is built (DESIGN_HISTORY.md, round 6: a two-genome call's longest
    segment keeps three of its 28 cuts because the library remembers a PREFIX of the cuts that held).  Remember WHICH cuts
    failed instead: the range in front of a failed cut runs on to the next cut that is kept.  Call by call -- each call drops
    the first kept cut that does not hold, as the library's validation finds it -- the plan settles on a set of cuts that all
    hold, and the joined output is the whole run's at every stage (what is behind the failed cut is covered by ONE more run
    from the last checked state, as today).  The same set comes out of ONE call when the run over the rest also writes
    down what it holds at every later cut: it is exact from the last checked state on, so comparing it there with what the
    later ranges hold at their cuts tells every cut that fails."""
    text = _text(11, period, copies, sub) if period else _text_two_arrays(11)
    oidx = oracle.Index.build(text)
    st = oracle.make_settings(k=K, gap=GAP, min_length=M)
    chunk = (0, len(text) - 1)
    status, offs, hits = oidx.probe_hits(oracle.prepare_needle(text, chunk, st), 0, st)
    offs, hits = offs.astype(np.int64), hits.astype(np.int64)
    hp = np.nonzero((status == 0) & (np.diff(offs) > 0))[0]
    s0, s1 = int(hp[0]), min(len(status), int(hp[-1]) + 1 + 2 * ((G + STEP - 1) // STEP))
    want = _families(Run(status, offs, hits, s0, s1, s0).recs)
    cnt = np.diff(offs)
    cuts = []
    for j in range(1, n_ranges):
        c = s0 + (int(hp[-1]) - s0) * j // n_ranges
        while status[c] or cnt[c] == 0:
            c += 1
        cuts.append(c)
    # call by call: the first kept cut that fails is dropped
    skipped, calls = set(), 0
    while True:
        calls += 1
        kept = _plan(cuts, skipped)
        runs, bad = _run_plan(status, offs, hits, s0, s1, kept, warm)
        if bad is None:
            break
        # what this call returns: the ranges in front of the failed cut, and one more run over the rest
        begin = s0 if bad == 0 else max(s0, kept[bad - 1] - warm)
        tail = Run(status, offs, hits, begin, s1, kept[bad - 1] if bad else s0)
        out, base = [], 0
        for r in runs[:bad] + [tail]:
            out += [(base + f,) + tuple(rest) for (f, *rest) in r.recs]
            base += r.flushes
        assert _families(out) == want, (calls, bad)
        skipped.add(cuts.index(kept[bad]))
        assert calls <= len(cuts) + 1
    out, base = [], 0
    for r in runs:
        out += [(base + f,) + tuple(rest) for (f, *rest) in r.recs]
        base += r.flushes
    assert _families(out) == want
    assert 0 < len(skipped) < len(cuts), (skipped, len(cuts))     # some cuts fail, some hold: a prefix would lose the latter
    first_bad = min(skipped)
    assert any(j > first_bad for j in range(len(cuts)) if j not in skipped), skipped   # cuts BEHIND the first failure hold
    # in one call: truth at every cut from the exact run, compared with what a range started `warm` probes in front of it holds
    at_once = set()
    last_good = s0
    for j, c in enumerate(cuts):
        true = Run(status, offs, hits, s0, c, s0).end
        cold = Run(status, offs, hits, max(s0, c - warm), c, c).end if c - warm > s0 else true
        if cold != true:
            at_once.add(j)
    assert at_once == skipped, (at_once, skipped)
