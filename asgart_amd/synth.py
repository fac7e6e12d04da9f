"""Seeded synthetic genomes shaped like the assemblies BASELINE.json names.

No real FASTA exists offline (no network), so every benchmark/test input is
synthetic and reproducible: i.i.d. background at ~41 % GC
with planted segmental duplications (direct and reverse-complemented, with
substitutions and indels), high-copy interspersed repeat families (exercise the
`max_cardinality` skip), tandem satellite arrays, N-runs (record ends, one large
gap per big record, scattered short runs) and soft-masked (lower-case) repeats
for `--skip-masked`.  Records are returned as raw FASTA-like byte arrays; feed
them to asgart_amd.prep.prepare_records.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

SEED_BASE = 0xA5617A27

# record length tables (bp) of the named assemblies
ECOLI_MG1655 = [4641652]
SCEREVISIAE_S288C = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888,
                     745751, 666816, 1078177, 924431, 784333, 1091291, 948066, 85779]
GRCH38_PRIMARY = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
                  145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
                  101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
                  156040895, 57227415, 16569]

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
# byte -> base index with P(A)=P(T)=76/256, P(C)=P(G)=52/256  (GC = 40.6 %)
_BG_LUT = np.concatenate([np.full(76, 0), np.full(52, 1), np.full(52, 2), np.full(76, 3)]).astype(np.uint8)
_COMP_IDX = np.array([3, 2, 1, 0], dtype=np.uint8)


def _background(rng: np.random.Generator, n: int) -> np.ndarray:
    """base indices 0..3"""
    out = np.empty(n, dtype=np.uint8)
    blk = 1 << 26
    for off in range(0, n, blk):
        m = min(blk, n - off)
        out[off:off + m] = _BG_LUT[rng.integers(0, 256, size=m, dtype=np.uint8)]
    return out


def _mutate(rng: np.random.Generator, seq: np.ndarray, sub_rate: float) -> np.ndarray:
    seq = seq.copy()
    if sub_rate > 0:
        mask = rng.random(seq.shape) < sub_rate
        cnt = int(mask.sum())
        if cnt:
            seq[mask] = (seq[mask] + rng.integers(1, 4, size=cnt, dtype=np.uint8)) & 3
    return seq


def _indels(rng: np.random.Generator, seq: np.ndarray, rate: float, max_len: int = 50) -> np.ndarray:
    n_ev = int(rng.poisson(len(seq) * rate))
    if n_ev == 0 or len(seq) < 4 * max_len:
        return seq
    pos = np.sort(rng.integers(1, len(seq) - 1, size=n_ev))
    parts, prev = [], 0
    for p in pos:
        if p <= prev:
            continue
        parts.append(seq[prev:p])
        ln = int(rng.integers(1, max_len + 1))
        if rng.random() < 0.5:
            parts.append(_background(rng, ln))  # insertion
            prev = p
        else:
            prev = min(len(seq), p + ln)        # deletion
    parts.append(seq[prev:])
    return np.concatenate(parts)


def make_genome(record_lengths: Sequence[int], seed: int, *, sd_per_mb: float = 2.0,
                sd_len: Tuple[int, int] = (1000, 200_000), alu_frac: float = 0.08,
                l1_frac: float = 0.02, sat_per_record: int = 2, gaps: bool = True,
                short_n_per_mb: float = 0.5, soft_mask: bool = True,
                alu_div: Tuple[float, float] = (0.10, 0.15), sat_copies: Tuple[int, int] = (200, 10_000)
                ) -> List[Tuple[str, np.ndarray]]:
    """-> [(name, uint8 sequence with upper/lower-case ACGT and N)]"""
    rng = np.random.default_rng(seed)
    lens = [int(x) for x in record_lengths]
    offs = np.concatenate(([0], np.cumsum(lens)))
    total = int(offs[-1])
    g = _background(rng, total)             # base indices
    lower = np.zeros(total, dtype=bool) if soft_mask else None
    is_n = np.zeros(total, dtype=bool)

    def rec_of(pos):
        return int(np.searchsorted(offs, pos, side="right") - 1)

    def place(seq: np.ndarray, mask_lower: bool = False):
        """overwrite a random location that keeps `seq` inside one record"""
        for _ in range(8):
            r = int(rng.integers(0, len(lens)))
            if lens[r] > len(seq) + 2:
                p = int(offs[r] + rng.integers(0, lens[r] - len(seq)))
                g[p:p + len(seq)] = seq
                if mask_lower and lower is not None:
                    lower[p:p + len(seq)] = True
                return p
        return -1

    # high-copy ~300 bp family at 10-15 % divergence (cardinality skip, repeat tails)
    n_alu = int(total * alu_frac / 300)
    if n_alu:
        cons = _background(rng, 300)
        blk = 1 << 16
        for off in range(0, n_alu, blk):
            m = min(blk, n_alu - off)
            copies = np.broadcast_to(cons, (m, 300)).copy()
            rate = rng.uniform(alu_div[0], alu_div[1], size=(m, 1))
            mask = rng.random((m, 300)) < rate
            copies[mask] = (copies[mask] + rng.integers(1, 4, size=int(mask.sum()), dtype=np.uint8)) & 3
            rev = rng.random(m) < 0.5
            copies[rev] = _COMP_IDX[copies[rev][:, ::-1]]
            for row in copies:
                place(row, True)
    # ~6 kb family, fewer copies, 5-15 % divergence
    n_l1 = int(total * l1_frac / 6000)
    if n_l1:
        cons = _background(rng, 6000)
        for _ in range(n_l1):
            cp = _mutate(rng, cons, float(rng.uniform(0.05, 0.15)))
            ln = int(rng.integers(500, 6001))      # 5'-truncated copies
            cp = cp[6000 - ln:]
            if rng.random() < 0.5:
                cp = _COMP_IDX[cp[::-1]]
            place(cp, True)
    # tandem satellite arrays (171-bp monomer)
    for r in range(len(lens)):
        if lens[r] < 2_000_000 and sat_copies[1] > 1000:
            continue
        for _ in range(sat_per_record):
            mono = _background(rng, 171)
            copies = int(rng.integers(sat_copies[0], max(sat_copies[0] + 1, min(sat_copies[1], lens[r] // 171 // 50))))
            arr = np.tile(mono, copies)
            arr = _mutate(rng, arr, float(rng.uniform(0.01, 0.05)))
            p = int(offs[r] + rng.integers(0, lens[r] - len(arr)))
            g[p:p + len(arr)] = arr
            if lower is not None:
                lower[p:p + len(arr)] = True
    # segmental duplications: copy number 2-6, intra- and inter-record
    n_sd = max(1, int(total / 1e6 * sd_per_mb))
    hi = max(sd_len[0] + 1, min(sd_len[1], max(lens) // 8))
    for _ in range(n_sd):
        ln = int(np.exp(rng.uniform(np.log(sd_len[0]), np.log(hi))))
        r = int(rng.integers(0, len(lens)))
        if lens[r] <= ln + 2:
            continue
        src = int(offs[r] + rng.integers(0, lens[r] - ln))
        base = g[src:src + ln].copy()
        for _c in range(int(rng.integers(1, 6))):
            cp = _mutate(rng, base, float(rng.uniform(0.01, 0.10)))
            cp = _indels(rng, cp, float(rng.uniform(0.001, 0.01)) / 25.0)
            if rng.random() < 0.5:
                cp = _COMP_IDX[cp[::-1]]
            place(cp)
    # N-runs
    if gaps:
        for r in range(len(lens)):
            if lens[r] >= 100_000:
                e = min(10_000, lens[r] // 50)
                is_n[offs[r]:offs[r] + e] = True
                is_n[offs[r + 1] - e:offs[r + 1]] = True
            if lens[r] >= 20_000_000:
                gl = lens[r] // 100
                p = int(offs[r] + lens[r] * 0.4)
                is_n[p:p + gl] = True
        n_short = int(total / 1e6 * short_n_per_mb)
        for _ in range(n_short):
            ln = int(rng.integers(1, 5000))
            p = int(rng.integers(0, max(1, total - ln)))
            if rec_of(p) == rec_of(p + ln - 1):
                is_n[p:p + ln] = True
    seq = _BASES[g]
    if lower is not None:
        seq[lower] |= 0x20
    seq[is_n] = ord("N")
    return [(f"chr{r + 1}", seq[offs[r]:offs[r + 1]]) for r in range(len(lens))]


def scaled(lengths: Sequence[int], total: int) -> List[int]:
    f = total / float(sum(lengths))
    return [max(1000, int(x * f)) for x in lengths]


_COMP_BYTE = np.arange(256, dtype=np.uint8)
for _a, _b in ((b"A", b"T"), (b"C", b"G"), (b"a", b"t"), (b"c", b"g")):
    _COMP_BYTE[_a[0]], _COMP_BYTE[_b[0]] = _b[0], _a[0]
_SUB_LUT = np.zeros((256, 3), dtype=np.uint8)          # base -> its three substitutes, case kept
for _set in (b"ACGT", b"acgt"):
    for _j, _c in enumerate(_set):
        _SUB_LUT[_c] = [_set[(_j + d) % 4] for d in (1, 2, 3)]


def diverged_genome(records: List[Tuple[str, np.ndarray]], seed: int, sub_rate: float = 0.012,
                    prefix: str = "b_") -> List[Tuple[str, np.ndarray]]:
    """A second genome derived from `records` the way a sister species' assembly relates to the
    first: every base substituted with probability `sub_rate`
    (case and N kept), then per record a few large inversions (reverse-complemented in place) and
    block swaps."""
    rng = np.random.default_rng(seed)
    out = []
    blk = 1 << 26
    for name, seq in records:
        s = seq.copy()
        n = len(s)
        for off in range(0, n, blk):
            part = s[off:off + blk]
            hit = np.flatnonzero(rng.random(len(part), dtype=np.float32) < sub_rate)
            alt = _SUB_LUT[part[hit], rng.integers(0, 3, size=len(hit))]
            keep = alt != 0                     # N and anything else that is not a base stays
            part[hit[keep]] = alt[keep]
        if n >= 200_000:
            for _ in range(max(1, n // 50_000_000)):   # inversions, 0.2-2 % of the record each
                ln = int(rng.integers(n // 500, n // 50))
                p = int(rng.integers(0, n - ln))
                s[p:p + ln] = _COMP_BYTE[s[p:p + ln][::-1]]
            for _ in range(max(1, n // 100_000_000)):  # swaps of two equally long blocks
                ln = int(rng.integers(n // 1000, n // 100))
                p, q = sorted(int(x) for x in rng.integers(0, n - ln, size=2))
                if q - p >= ln:
                    tmp = s[p:p + ln].copy()
                    s[p:p + ln] = s[q:q + ln]
                    s[q:q + ln] = tmp
        out.append((prefix + name, s))
    return out


def repeat_rich_genome(record_lengths: Sequence[int], seed: int, *, sine_frac: float = 0.27, line_frac: float = 0.15,
                       div: Tuple[float, float] = (0.01, 0.05), sine_family: Tuple[int, int] = (60, 220),
                       line_family: Tuple[int, int] = (15, 60)) -> List[Tuple[str, np.ndarray]]:
    """A genome whose interspersed repeats are YOUNG: `sine_frac` of it 300-bp elements and `line_frac` 5'-truncated
    6-kb elements, in many families of moderate copy number at 1-5 % divergence from their consensus, half of the
    copies reverse-complemented.  Most 20-mers of a copy then recur exactly in its relatives (0.97^20 = 54 %), so
    well over a third of the probes pass the presence filter and carry tens of hits -- the regime of the young
    Alu / L1 subfamilies of a real assembly, which config_genome's single old high-copy family (10-15 % diverged:
    four probes in five have no second occurrence) does not cover.  Family sizes stay below max_cardinality so the
    probes are extended, not skipped.  Plus what make_genome plants at its defaults' scale: a few segmental
    duplications, satellite arrays and N-runs."""
    recs = make_genome(record_lengths, seed, alu_frac=0.0, l1_frac=0.0)
    rng = np.random.default_rng(seed + 1000)
    lens = [len(s) for _, s in recs]
    offs = np.concatenate(([0], np.cumsum(lens)))
    total = int(offs[-1])
    seq = np.concatenate([s for _, s in recs])
    code = np.zeros(256, dtype=np.uint8)
    for i_, ch in enumerate(b"ACGT"):
        code[ch] = i_
        code[ch | 0x20] = i_

    def plant(cons: np.ndarray, n_copies: int, truncate: bool):
        ln0 = len(cons)
        for _ in range(n_copies):
            cp = _mutate(rng, cons, float(rng.uniform(div[0], div[1])))
            if truncate:
                cp = cp[ln0 - int(rng.integers(500, ln0 + 1)):]
            if rng.random() < 0.5:
                cp = _COMP_IDX[cp[::-1]]
            r = int(rng.integers(0, len(lens)))
            if lens[r] <= len(cp) + 2:
                continue
            p = int(offs[r] + rng.integers(0, lens[r] - len(cp)))
            keep_n = seq[p:p + len(cp)] == ord("N")
            new = _BASES[cp] | 0x20           # soft-masked, like a RepeatMasker'd assembly
            new[keep_n] = ord("N")
            seq[p:p + len(cp)] = new

    planted = 0
    while planted < total * sine_frac:
        n = int(rng.integers(sine_family[0], sine_family[1] + 1))
        plant(_background(rng, 300), n, False)
        planted += 300 * n
    planted = 0
    while planted < total * line_frac:
        n = int(rng.integers(line_family[0], line_family[1] + 1))
        plant(_background(rng, 6000), n, True)
        planted += 3250 * n
    return [(name, seq[offs[r]:offs[r + 1]]) for r, (name, _) in enumerate(recs)]


def plant_hor_arrays(records: List[Tuple[str, np.ndarray]], seed: int, *, per_record: int = 1,
                     array_bp: Tuple[int, int] = (400_000, 3_000_000), min_record: int = 40_000_000,
                     monomer_div: Tuple[float, float] = (0.18, 0.32), copy_div: Tuple[float, float] = (0.01, 0.02)
                     ) -> List[Tuple[str, np.ndarray]]:
    """Centromere-like satellite arrays with HIGHER-ORDER structure, planted over `records` (in place of what was
    there): a 171-bp monomer, m = 6..12 variants of it that differ from one another by `monomer_div` (as the monomers
    of an alpha-satellite higher-order repeat do), and that m-monomer unit repeated over `array_bp` bases with only
    `copy_div` substitutions between copies of the unit.  A probe inside such an array matches the SAME monomer of every
    other unit copy (period 171 m, near-identical) and hardly its neighbours -- unlike make_genome's flat arrays, whose
    copies all derive from one monomer."""
    rng = np.random.default_rng(seed)
    out = []
    for name, seq in records:
        seq = seq.copy()
        if len(seq) >= min_record:
            for _ in range(per_record):
                base = _background(rng, 171)
                m = int(rng.integers(6, 13))
                unit = np.concatenate([_mutate(rng, base, float(rng.uniform(*monomer_div))) for _ in range(m)])
                total = int(min(rng.integers(array_bp[0], array_bp[1] + 1), len(seq) // 20))
                copies = max(2, total // len(unit))
                arr = np.tile(unit, copies)
                rate = float(rng.uniform(*copy_div))
                blk = 1 << 22
                for off in range(0, len(arr), blk):
                    arr[off:off + blk] = _mutate(rng, arr[off:off + blk], rate)
                p = int(rng.integers(len(seq) // 3, len(seq) // 3 + len(seq) // 4))
                keep_n = seq[p:p + len(arr)] == ord("N")
                new = _BASES[arr] | 0x20
                new[keep_n[:len(new)]] = ord("N")
                seq[p:p + len(arr)] = new[:len(seq) - p]
        out.append((name, seq))
    return out


def config_genome(cfg: int, scale: float = 1.0) -> List[Tuple[str, np.ndarray]]:
    """Synthetic stand-ins for BASELINE.json configs 1-5 (scale<1 shrinks them).  Config 5 is two
    "files": the config-4 genome plus a 1.2 %-diverged, rearranged copy of it (the cross-genome
    run concatenates the records of all input files, reference src/bin/asgart.rs:375-395)."""
    if cfg == 5:
        first = config_genome(4, scale)
        return first + diverged_genome(first, SEED_BASE + 5)
    if cfg == 6:   # chr1-sized, repeat-rich (young interspersed repeats): not a BASELINE.json config, a realism check
        lens6 = GRCH38_PRIMARY[:1] if scale == 1.0 else scaled(GRCH38_PRIMARY[:1], int(GRCH38_PRIMARY[0] * scale))
        return repeat_rich_genome(lens6, SEED_BASE + 6)
    if cfg == 7:   # GRCh38-sized, repeat-rich, with higher-order satellite arrays: the realism check at full size
        lens7 = GRCH38_PRIMARY if scale == 1.0 else scaled(GRCH38_PRIMARY, int(sum(GRCH38_PRIMARY) * scale))
        return plant_hor_arrays(repeat_rich_genome(lens7, SEED_BASE + 7), SEED_BASE + 77,
                                min_record=int(40_000_000 * min(1.0, scale)) if scale < 1.0 else 40_000_000)
    table = {1: ECOLI_MG1655, 2: SCEREVISIAE_S288C, 3: GRCH38_PRIMARY[:1], 4: GRCH38_PRIMARY}[cfg]
    lens = table if scale == 1.0 else scaled(table, int(sum(table) * scale))
    return make_genome(lens, SEED_BASE + cfg)
