"""Feasibility of INTRA-SEGMENT parallelism (DESIGN.md section 8 "next (4)") -- on the CPU.

The floor of a pass is its longest automaton segment: tens of thousands of strictly serial probes (a tandem array, a
higher-order satellite array, a chromosome against its homologue).  Could such a segment be cut into P probe ranges that
run side by side?  The only exact scheme anybody has proposed: range j starts H probes BEFORE its cut from an EMPTY arm
list (speculation), and is accepted if, at the cut, its live arms equal the live arms its predecessor ends with.
This script measures how often that holds.

For each segment kind, the reference automaton (src/automaton.rs:96-201, restated here on numpy arrays, live arms only:
inactive arms never match again, src/automaton.rs:67) is run once over the whole segment ("true" run) with its live-arm
list recorded at P cuts, and again from an empty list started H probes before every cut, for H = 1, 4, 16, 64, 256 x t*
(t* = ceil(G / step) = 12 probes: the horizon after which an UNEXTENDED arm is dead).  Compared at the cut, in list order:

    L1  same number of live arms and the same (right.end, gap) sequence      -- what the next probe's matching reads,
                                                                               thresholds aside
    L2  L1 and the same threshold max(G, len(left) / 10) for every arm       -- src/automaton.rs:69: the next decisions agree
    L3  L2 and the same left.start / right.start                             -- everything; L3 <=> no live arm at the cut is
                                                                               older than the speculation AND the young ones
                                                                               were not influenced by older ones

Only L3 is safe without further checks: thresholds GROW with len(left) = i + k - left.start, so an arm whose left.start
differs takes different decisions later even when it agrees now (L2), unless no hit ever falls between the two windows.
Also reported: how old the live arms at a cut are (the necessary condition for L3: nobody older than H).

Round 6: what THRESHOLD-INTERVAL CARRY could join at best.  The carry accepts a cut at level L1
when, through the range behind it, no decision of the speculative run would have come out differently under the true (larger)
thresholds.  Its ceiling is therefore the share of cuts that are L1 at the cut AND STAY L1 -- same (right.end, gap) sequence
after every probe -- until the next cut ("L1 stable"): where the true run's old arm, with its window of len / 10, takes a hit
the young arm of the speculative run cannot reach, the two runs part for good.  `homolog5` is the segment cfg5 is made of: a
region of the config-4 genome (old high-copy interspersed repeats) against its 1.2 %-diverged copy.

    python tools/split_feasibility.py [pole|pole_rc|hor|homolog|homolog5 ...]   (CPU only: the oracle's suffix array and hit rows)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (suffix array + per-probe hit rows in SA order; this tool is test infrastructure)

K, GAP, STEP = 20, 100, 10
G = GAP + K
TSTAR = (G + STEP - 1) // STEP
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(rng, arr, rate):
    arr = arr.copy()
    mut = rng.random(arr.shape) < rate
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()))) & 3
    return arr


def make_segment(kind, rng):
    """-> (text incl. '$', reverse/complement flag): one long segment between random flanks"""
    flank = 200_000
    if kind == "pole":                     # tools/pole_synth.py: flat tandem array, 171 bp x 3800, 3 % (cfg4's pole)
        arr = mutate(rng, np.tile(rng.integers(0, 4, size=171), 3800), 0.03)
    elif kind == "pole_rc":                # the same array and an inverted copy of it further on, searched -RC: every probe
        a = np.tile(rng.integers(0, 4, size=171), 1900)   # of one meets the whole of the other (cfg4's longest -RC segment)
        arr = np.concatenate([mutate(rng, a, 0.03), rng.integers(0, 4, size=50_000), (3 - mutate(rng, a, 0.03))[::-1]])
    elif kind == "hor":                    # synth.plant_hor_arrays: 8 monomers at 25 % from one another, unit x 730 at 1.5 %
        base = rng.integers(0, 4, size=171)
        unit = np.concatenate([mutate(rng, base, 0.25) for _ in range(8)])
        arr = mutate(rng, np.tile(unit, 730), 0.015)
    elif kind == "homolog":                # a 1-Mb region and its 1.2 %-diverged copy (cfg5's chromosome pairs, in small)
        a = rng.integers(0, 4, size=1_000_000)
        arr = np.concatenate([a, rng.integers(0, 4, size=50_000), mutate(rng, a, 0.012)])
    elif kind == "homolog5":               # the same with the repeats of the config-4 genome in it (what cfg5 is made of)
        from asgart_amd import synth
        rec = synth.make_genome([1_500_000], 1234)[0][1]
        code = np.zeros(256, dtype=np.int64)
        for i_, ch in enumerate(b"ACGT"):
            code[ch] = i_
            code[ch | 0x20] = i_
        a = code[rec]
        a[(rec == ord("N")) | (rec == ord("n"))] = rng.integers(0, 4, size=int(((rec == ord("N")) | (rec == ord("n"))).sum()))
        arr = np.concatenate([a, rng.integers(0, 4, size=50_000), mutate(rng, a, 0.012)])
    else:
        raise SystemExit(f"unknown segment kind {kind}")
    g = np.concatenate([rng.integers(0, 4, size=flank), arr, rng.integers(0, 4, size=flank)])
    return np.concatenate([BASES[g], np.frombuffer(b"$", dtype=np.uint8)]), kind == "pole_rc"


class Arms:
    """live arms in list (= creation) order"""
    __slots__ = ("ls", "le", "rs", "re", "gap", "born")

    def __init__(self):
        z = np.zeros(0, dtype=np.int64)
        self.ls, self.le, self.rs, self.re, self.gap, self.born = z, z, z, z, z, z

    def snapshot(self):
        return tuple(a.copy() for a in (self.ls, self.le, self.rs, self.re, self.gap, self.born))


def step_probe(A, t, i, x):
    """one PROCESSED probe (src/automaton.rs:119-171) at needle offset i with hits x (SA order), live arms A"""
    n_a = len(A.re)
    new_mask = np.ones(len(x), dtype=bool)
    dirty = np.zeros(n_a, dtype=bool)
    if n_a and len(x):
        thr = np.maximum(G, (A.le - A.ls) // 10)
        d = x[:, None] - A.re[None, :]
        acc = (d > -K) & (d < thr[None, :])            # re - k < x < re + thr  (the whole predicate)
        has = acc.any(axis=1)
        first = acc.argmax(axis=1)                      # first accepting arm in list order (:67-78)
        new_mask = ~has
        hs = np.nonzero(has)[0]
        if len(hs):
            arms = first[hs]
            # ExtendArm applied in hit order: the LAST hit of an arm wins (:136-143)
            last = np.full(n_a, -1, dtype=np.int64)
            np.maximum.at(last, arms, hs)
            w = last >= 0
            A.le = np.where(w, i + K, A.le)
            A.re = np.where(w, x[np.maximum(last, 0)] + K, A.re)
            A.gap = np.where(w, 0, A.gap)
            dirty = w
    # non-dirty arms age (:166-171)
    A.gap = np.where(dirty, A.gap, A.gap + STEP)
    keep = A.gap < G
    nx = x[new_mask]
    if len(nx):                                         # NewArm in hit order (:145-163); they age at once: gap = step
        n_new = len(nx)
        new_alive = STEP < G
        if new_alive:
            A.ls = np.concatenate([A.ls[keep], np.full(n_new, i, dtype=np.int64)])
            A.le = np.concatenate([A.le[keep], np.full(n_new, i + K, dtype=np.int64)])
            A.rs = np.concatenate([A.rs[keep], nx])
            A.re = np.concatenate([A.re[keep], nx + K])
            A.gap = np.concatenate([A.gap[keep], np.full(n_new, STEP, dtype=np.int64)])
            A.born = np.concatenate([A.born[keep], np.full(n_new, t, dtype=np.int64)])
            return
    if not keep.all():
        A.ls, A.le, A.rs, A.re, A.gap, A.born = (a[keep] for a in (A.ls, A.le, A.rs, A.re, A.gap, A.born))


def run(status, offs, hits, t0, t1, A, cuts=None, sig=None, sig_from=0):
    """probes t0 .. t1-1 of the chunk; -> {cut: snapshot} for the cuts passed on the way.  sig: a dict that receives, per
    probe t >= sig_from, a signature of the (right.end, gap) sequence the arms have AFTER the probe (level L1)"""
    snaps = {}
    for t in range(t0, t1):
        if cuts is not None and t in cuts:
            snaps[t] = A.snapshot()
        if not status[t]:                               # skipped probes neither age nor reset (:100-102, :115-117)
            step_probe(A, t, (t + 1) * STEP, hits[offs[t]:offs[t + 1]])
        if sig is not None and t >= sig_from:
            sig[t] = hash((A.re.tobytes(), A.gap.tobytes()))
    return snaps


def main():
    kinds = sys.argv[1:] or ["pole", "pole_rc", "hor", "homolog"]
    P = int(os.environ.get("SPLIT_CUTS", "48"))
    print(f"k={K} g={GAP} (G={G}, t*={TSTAR}); {P} cuts per segment; H in units of t*\n")
    for kind in kinds:
        rng = np.random.default_rng(5)
        text, rc = make_segment(kind, rng)
        t_a = time.time()
        oidx = oracle.Index.build(text)
        st = oracle.make_settings(k=K, gap=GAP, reverse=rc, complement=rc)
        chunk = (0, len(text) - 1)
        status, offs, hits = oidx.probe_hits(oracle.prepare_needle(text, chunk, st), 0, st)
        offs = offs.astype(np.int64)
        hits = hits.astype(np.int64)
        n_p = len(status)
        cnt = np.diff(offs)
        hp = np.nonzero((status == 0) & (cnt > 0))[0]
        # the longest segment: from the first to the last hit-probe of the array region (quiet runs >= t* split segments;
        # the flanks are unique, so the array is ONE segment unless it has a quiet run inside)
        quiet_break = np.nonzero(np.diff(hp) > TSTAR + 64)[0]
        bounds = np.concatenate([[0], quiet_break + 1, [len(hp)]])
        j = int(np.argmax(np.diff(bounds)))
        s0, s1 = int(hp[bounds[j]]), int(hp[bounds[j + 1] - 1]) + 1
        cuts = sorted({int(s0 + (s1 - s0) * (q + 1) // (P + 1)) for q in range(P)})
        A = Arms()
        true_sig = {}
        snaps = run(status, offs, hits, s0, s1, A, set(cuts), true_sig)
        n_live = np.array([len(snaps[c][3]) for c in cuts])
        ages = [c - snaps[c][5] for c in cuts]
        oldest = np.array([a.max() if len(a) else 0 for a in ages])
        print(f"== {kind}: {len(text) - 1} bp, {'-RC' if rc else 'direct'}; longest segment: probes {s0}..{s1} ({s1 - s0} probes, "
              f"{int(((status[s0:s1] == 0) & (cnt[s0:s1] > 0)).sum())} hit-probes, {cnt[s0:s1].mean():.1f} hits per probe); "
              f"live arms at the cuts: mean {n_live.mean():.0f}, max {n_live.max()}; oldest live arm at a cut: median "
              f"{int(np.median(oldest))} probes, max {oldest.max()} (true run {time.time() - t_a:.0f} s)")
        print(f"   {'H':>8} {'probes':>7} | {'L1 (re,gap)':>12} {'L1 stable':>10} {'L2 (+thr)':>10} {'L3 (all)':>9} | cuts with no live arm older than H | "
              f"live arms older than H")
        for mult in (1, 4, 16, 64, 256, 1024):
            H = mult * TSTAR
            ok = [0, 0, 0]
            stable = 0
            young_cuts = 0
            older = 0
            total = 0
            for c in cuts:
                start = max(s0, c - H)
                B = Arms()
                run(status, offs, hits, start, c, B)
                ls, le, rs, re, gap, born = snaps[c]
                total += len(re)
                older += int((c - born > H).sum()) if start > s0 else 0
                young_cuts += int(start == s0 or not (c - born > H).any())
                l1 = len(B.re) == len(re) and np.array_equal(B.re, re) and np.array_equal(B.gap, gap)
                l2 = l1 and np.array_equal(np.maximum(G, (B.le - B.ls) // 10), np.maximum(G, (le - ls) // 10))
                l3 = l2 and np.array_equal(B.ls, ls) and np.array_equal(B.rs, rs) and np.array_equal(B.le, le)
                ok[0] += l1
                ok[1] += l2
                ok[2] += l3
                if l1:   # ... and does it STAY L1 until the next cut (the ceiling of threshold-interval carry)?
                    nxt = min([c2 for c2 in cuts if c2 > c] or [s1])
                    cold_sig = {}
                    run(status, offs, hits, c, nxt, B, None, cold_sig)
                    stable += all(cold_sig[t] == true_sig[t] for t in range(c, nxt))
            n_c = len(cuts)
            print(f"   {mult:>5} t* {H:>7} | {ok[0] / n_c:>11.0%} {stable / n_c:>10.0%} {ok[1] / n_c:>10.0%} {ok[2] / n_c:>9.0%} | "
                  f"{young_cuts / n_c:>33.0%} | {older / max(1, total):>21.1%}", flush=True)
        print()


if __name__ == "__main__":
    main()
