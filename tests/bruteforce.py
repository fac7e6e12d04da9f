"""Second, independent restatement of the reference hot path in pure Python.

Used ONLY to cross-check the C oracle on small inputs (<= ~100 kb): it shares
no code with oracle/asgart_oracle.c, builds its suffix array by sorting suffix
slices, and writes the automaton with Python lists/dicts straight from the
Rust text (src/automaton.rs:57-216, src/searcher.rs:145-180,
src/bin/asgart.rs:201-253).  Test infrastructure, never shipped.
"""
from __future__ import annotations

from typing import List, Tuple

ALPHABET = b"ATGCN"  # src/structs.rs:10


def suffix_array(text: bytes) -> List[int]:
    """Bytewise-lexicographic suffix order (what divsufsort64 returns)."""
    return sorted(range(len(text)), key=lambda i: text[i:])


def cache_entry(text: bytes, sa: List[int], p8: bytes) -> Tuple[int, int]:
    """Searcher::new entry: SA slots whose suffix starts with the 8-mer."""
    slots = [r for r, x in enumerate(sa) if text[x:x + 8] == p8]
    if not slots:
        return (0, 0)
    assert slots == list(range(slots[0], slots[-1] + 1))
    return (slots[0], slots[-1] + 1)


def _published_compare(T: bytes, P: bytes, suf: int, match: int):
    """libdivsufsort lib/utils.c `_compare` (published upstream source, restated)."""
    i, j, r = suf + match, match, 0
    while i < len(T) and j < len(P):
        r = T[i] - P[j]
        if r != 0:
            break
        i += 1
        j += 1
    return ((-1 if j != len(P) else 0) if r == 0 else r), j


def sa_search_published(T: bytes, P: bytes, SA: List[int], init_left: int, init_right: int) -> Tuple[int, int]:
    """libdivsufsort lib/utils.c `sa_search` started on [init_left, init_right): what the fork's
    `sa_searchb64` (src/searcher.rs:118-128) is taken to be.  -> (left, count).  Matters only when SA
    is not sorted under the comparator (--trim: suffixes of the sub-strand compared through the whole
    text)."""
    if len(T) == 0 or init_right <= init_left:
        return (-1, 0)
    i = j = k = init_left
    lmatch = rmatch = 0
    size = init_right - init_left
    half = size >> 1
    while 0 < size:
        match = min(lmatch, rmatch)
        r, match = _published_compare(T, P, SA[i + half], match)
        if r < 0:
            i += half + 1
            half -= (size & 1) ^ 1
            lmatch = match
        elif r > 0:
            rmatch = match
        else:
            lsize, j, rsize, k = half, i, size - half - 1, i + half + 1
            llmatch, lrmatch, half = lmatch, match, lsize >> 1
            while 0 < lsize:
                lmatch = min(llmatch, lrmatch)
                r, lmatch = _published_compare(T, P, SA[j + half], lmatch)
                if r < 0:
                    j += half + 1
                    half -= (lsize & 1) ^ 1
                    llmatch = lmatch
                else:
                    lrmatch = lmatch
                lsize, half = half, half >> 1
            rlmatch, rrmatch, half = match, rmatch, rsize >> 1
            while 0 < rsize:
                rmatch = min(rlmatch, rrmatch)
                r, rmatch = _published_compare(T, P, SA[k + half], rmatch)
                if r <= 0:
                    k += half + 1
                    half -= (rsize & 1) ^ 1
                    rlmatch = rmatch
                else:
                    rrmatch = rmatch
                rsize, half = half, half >> 1
            break
        size, half = half, half >> 1
    return ((j if 0 < k - j else i), k - j)


def trim_suffix_array(text: bytes, start: int, end: int) -> List[int]:
    """src/bin/asgart.rs:142-148: SA of data[start..end] + '$', shifted by +start."""
    sub = text[start:end] + b"$"
    return [x + start for x in suffix_array(sub)]


def _cmp(text: bytes, x: int, pattern: bytes) -> int:
    """comparator of src/searcher.rs:164-170: -1 Less, 0 Equal, 1 Greater"""
    if x + len(pattern) > len(text):
        return -1
    s = text[x:x + len(pattern)]
    return -1 if s < pattern else (1 if s > pattern else 0)


def equal_range_superslice(text: bytes, sl: List[int], pattern: bytes) -> Tuple[int, int]:
    """superslice::Ext::equal_range_by as recalled (two-base halving bisection)."""
    size = len(sl)
    if size == 0:
        return (0, 0)
    b0 = b1 = 0
    while size > 1:
        half = size // 2
        m0, m1 = b0 + half, b1 + half
        c0, c1 = _cmp(text, sl[m0], pattern), _cmp(text, sl[m1], pattern)
        if c0 == -1:
            b0 = m0
        if c1 != 1:
            b1 = m1
        size -= half
    c0, c1 = _cmp(text, sl[b0], pattern), _cmp(text, sl[b1], pattern)
    return (b0 + (1 if c0 == -1 else 0), b1 + (1 if c1 != 1 else 0))


def search(text: bytes, sa: List[int], pattern: bytes, exact_bisection: bool = True) -> List[int]:
    """Searcher::search -> hit starts in SA order."""
    assert all(c in ALPHABET for c in pattern[:8]), "reference panics here"
    if len(sa) != len(text):   # --trim: the cache entries are what the bisection of sa_searchb64 finds
        left, count = sa_search_published(text, pattern[:8], sa, 0, len(sa))
        lo, hi = left, left + count
    else:
        lo, hi = cache_entry(text, sa, pattern[:8])
    if exact_bisection:
        s, e = equal_range_superslice(text, sa[lo:hi], pattern)
        return sa[lo + s:lo + max(s, e)]
    return [x for x in sa[lo:hi] if text[x:x + len(pattern)] == pattern and x + len(pattern) <= len(text)]


def d_ss(a: Tuple[int, int], m: Tuple[int, int]) -> int:
    """src/automaton.rs:207-216"""
    if (a[0] <= m[0] <= a[1]) or (a[0] <= m[1] <= a[1]):
        return 0
    return min(abs(a[0] - m[1]), abs(a[1] - m[0]))


def search_duplications(needle: bytes, needle_offset: int, strand: bytes, sa: List[int],
                        k: int, max_gap_size: int, min_len: int, max_card: int, reverse: bool,
                        exact_bisection: bool = True):
    """automaton::search_duplications; returns (families, per-probe log).

    families: list of lists of (left_local, right, left_length, right_length)
    log: list of (i, status, [filtered hit starts]) with status 0/1/2.
    """
    fams, log = [], []
    arms: List[dict] = []
    step = k // 2
    L = len(needle)
    if L < min_len or L < k + step:
        return fams, log
    i = 0
    while i < L - k - step:
        i += step
        if needle[i] == ord("N"):
            log.append((i, 1, []))
            continue
        ms = [x for x in search(strand, sa, needle[i:i + k], exact_bisection) if x != i]
        if not reverse:
            ms = [x for x in ms if x > i + needle_offset]
        else:
            ms = [x for x in ms if x >= needle_offset + L - i]
        if len(ms) > max_card:
            log.append((i, 2, []))
            continue
        log.append((i, 0, list(ms)))
        for a in arms:
            a["dirty"] = False
        todo = []
        for x in ms:
            m = (x, x + k)
            op = ("new", i, m[0], m[1])
            for j, a in enumerate(arms):
                llen = a["l"][1] - a["l"][0]
                if (a["active"] and d_ss(tuple(a["r"]), m) < max(max_gap_size, int(0.1 * float(llen)))
                        and m[1] > a["r"][1]):
                    op = ("ext", j, i + k, m[1])
                    break
            todo.append(op)
        for op in todo:
            if op[0] == "ext":
                a = arms[op[1]]
                a["l"][1] = op[2]
                a["r"][1] = op[3]
                a["dirty"] = True
                a["gap"] = 0
        for op in todo:
            if op[0] == "new":
                arms.append({"l": [op[1], op[1] + k], "r": [op[2], op[3]], "active": True,
                             "dirty": False, "gap": 0})
        for a in arms:
            if not a["dirty"]:
                a["gap"] += step
                if a["gap"] >= max_gap_size:
                    a["active"] = False
        if len(arms) > 200:
            arms = [a for a in arms if a["active"] or a["l"][1] - a["l"][0] >= min_len
                    or a["r"][1] - a["r"][0] >= min_len]
        if arms and not any(a["active"] for a in arms):
            fam = [(a["l"][0], a["r"][0], a["l"][1] - a["l"][0], a["r"][1] - a["r"][0])
                   for a in arms if a["r"][1] - a["r"][0] >= min_len]
            if fam:
                fams.append(fam)
            arms = []
    return fams, log


_COMP = {ord("A"): ord("T"), ord("T"): ord("A"), ord("G"): ord("C"), ord("C"): ord("G"),
         ord("N"): ord("N"), ord("a"): ord("t"), ord("t"): ord("a"), ord("g"): ord("c"),
         ord("c"): ord("g"), ord("n"): ord("n")}


def complemented(seq: bytes) -> bytes:
    """src/utils.rs:1-23"""
    return bytes(_COMP.get(c, ord("N")) for c in seq)


def run(strand: bytes, sa: List[int], chunks, k=20, gap=100, min_len=1000, max_card=500,
        reverse=False, complement=False, exact_bisection: bool = True):
    """SearchDuplications::run body, src/bin/asgart.rs:201-253."""
    out = []
    for (start, length) in chunks:
        needle = strand[start:start + length]
        if complement:
            needle = complemented(needle)
        if reverse:
            needle = needle[::-1]
        fams, _ = search_duplications(needle, start, strand, sa, k, gap + k, min_len, max_card,
                                      reverse, exact_bisection)
        for fam in fams:
            fixed = []
            for (l, r, ll, rl) in fam:
                gl = l + start if not reverse else start + length - l - ll
                fixed.append((gl, r, ll, rl))
            out.append(fixed)
    return out


def find_chunks(strand: bytes):
    """src/bin/asgart.rs:317-366"""
    threshold = 5000
    start = count = i = 0
    chunks = []
    n = len(strand)
    while i < n:
        if strand[i] in b"nN":
            j = i
            while j < n and strand[j] in b"nN":
                j += 1
            nc = j - i
            if nc > threshold:
                if count > 0:
                    chunks.append((start, count))
                    count = 0
                start = i + nc
            else:
                count += nc
            i += nc
        else:
            if count == 0:
                count, start = 1, i
            else:
                count += 1
            i += 1
    if count:
        chunks.append((start, count))
    if not chunks:
        chunks.append((0, n))
    return chunks


# ---- KAT text generator -----------------------------
def lcg_bases(seed: int, n: int) -> bytes:
    x = seed
    out = bytearray()
    for _ in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out.append(b"ACGT"[x >> 62])
    return bytes(out)


def revcomp(seq: bytes) -> bytes:
    return complemented(seq)[::-1]
