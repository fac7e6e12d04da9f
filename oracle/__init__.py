"""ctypes binding of the CPU oracle (oracle/asgart_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (asgart_amd) never imports
this module.  PARITY UNPINNED -- see oracle/README.md.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libasgart_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make).  Idempotent."""
    srcs = [os.path.join(_HERE, f) for f in ("asgart_oracle.c", "sais.c", "postprocess.c", "asgart_oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "libasgart_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Settings(C.Structure):
    """oracle_settings == RunSettings fields on the path (src/structs.rs:36-58)."""

    _fields_ = [
        ("probe_size", C.c_uint64),
        ("max_gap_size", C.c_uint32),
        ("min_duplication_length", C.c_uint64),
        ("max_cardinality", C.c_uint64),
        ("reverse", C.c_uint8),
        ("complement", C.c_uint8),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "probes_total", "probes_n_skipped", "probes_searched", "probes_card_skipped",
        "probes_with_hits", "bisect_steps", "raw_hits", "filtered_hits", "arm_tests",
        "proto_sds", "families")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def make_settings(k=20, gap=100, min_length=1000, max_cardinality=500, reverse=False,
                  complement=False) -> Settings:
    """CLI -> RunSettings as src/bin/asgart.rs:677-693 (max_gap_size = g + k)."""
    return Settings(k, gap + k, min_length, max_cardinality, int(reverse), int(complement))


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u8p, i64p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    vp = C.c_void_p
    L.oracle_divsufsort64.argtypes = [vp, vp, C.c_int64]
    L.oracle_divsufsort64.restype = C.c_int32
    L.oracle_sa_check.argtypes = [vp, vp, C.c_int64]
    L.oracle_sa_check.restype = C.c_int64
    L.oracle_searcher_new.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_uint64]
    L.oracle_searcher_new.restype = vp
    L.oracle_searcher_free.argtypes = [vp]
    L.oracle_searcher_free.restype = None
    L.oracle_searcher_cache_get.argtypes = [vp, vp, u64p, u64p]
    L.oracle_searcher_cache_get.restype = C.c_int32
    L.oracle_searcher_search.argtypes = [vp, vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64,
                                         u64p, u64p, u64p]
    L.oracle_searcher_search.restype = C.c_int64
    L.oracle_families_counts.argtypes = [vp, u64p, u64p]
    L.oracle_families_counts.restype = None
    L.oracle_families_copy.argtypes = [vp, vp, vp]
    L.oracle_families_copy.restype = None
    L.oracle_families_free.argtypes = [vp]
    L.oracle_families_free.restype = None
    L.oracle_search_duplications.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_int64, vp, vp,
                                             vp, C.POINTER(Settings), C.POINTER(Stats),
                                             C.POINTER(vp)]
    L.oracle_search_duplications.restype = C.c_int32
    L.oracle_run.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int64, C.POINTER(Settings),
                             C.c_int32, vp, C.POINTER(Stats), C.POINTER(vp)]
    L.oracle_run.restype = C.c_int32
    L.oracle_probe_hits.argtypes = [vp, C.c_uint64, C.c_uint64, vp, C.c_int64, vp, vp,
                                    C.POINTER(Settings), vp, vp, vp, u64p]
    L.oracle_probe_hits.restype = C.c_int64
    L.oracle_normalise.argtypes = [vp, C.c_uint64, C.c_int32]
    L.oracle_normalise.restype = None
    L.oracle_find_chunks.argtypes = [vp, C.c_uint64, vp, C.c_int64]
    L.oracle_find_chunks.restype = C.c_int64
    L.oracle_complemented.argtypes = [vp, vp, C.c_uint64]
    L.oracle_complemented.restype = None
    L.oracle_postprocess.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(vp)]
    L.oracle_postprocess.restype = C.c_int32
    L.oracle_d_ss.argtypes = [C.c_uint64] * 4
    L.oracle_d_ss.restype = C.c_int64
    _lib = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def as_text(seq) -> np.ndarray:
    if isinstance(seq, np.ndarray):
        assert seq.dtype == np.uint8
        return np.ascontiguousarray(seq)
    if isinstance(seq, str):
        seq = seq.encode()
    return np.frombuffer(bytes(seq), dtype=np.uint8).copy()


def divsufsort64(text: np.ndarray) -> np.ndarray:
    """r_divsufsort (src/bin/asgart.rs:473-479)."""
    text = as_text(text)
    sa = np.empty(len(text), dtype=np.int64)
    rc = lib().oracle_divsufsort64(_ptr(text), _ptr(sa), len(text))
    if rc != 0:
        raise RuntimeError(f"oracle_divsufsort64 -> {rc}")
    return sa


def sa_check(text: np.ndarray, sa: np.ndarray) -> int:
    return int(lib().oracle_sa_check(_ptr(text), _ptr(sa), len(text)))


def d_ss(a: Tuple[int, int], m: Tuple[int, int]) -> int:
    return int(lib().oracle_d_ss(a[0], a[1], m[0], m[1]))


def normalise(seq, skip_masked: bool) -> np.ndarray:
    a = as_text(seq)
    lib().oracle_normalise(_ptr(a), len(a), int(skip_masked))
    return a


def find_chunks(strand) -> List[Tuple[int, int]]:
    a = as_text(strand)
    n = lib().oracle_find_chunks(_ptr(a), len(a), None, 0)
    out = np.zeros(2 * n, dtype=np.uint64)
    lib().oracle_find_chunks(_ptr(a), len(a), _ptr(out), n)
    return [(int(out[2 * j]), int(out[2 * j + 1])) for j in range(n)]


def complemented(seq) -> np.ndarray:
    a = as_text(seq)
    out = np.empty_like(a)
    lib().oracle_complemented(_ptr(a), _ptr(out), len(a))
    return out


Families = List[List[Tuple[int, int, int, int]]]


def _take_families(handle) -> Tuple[np.ndarray, np.ndarray]:
    L = lib()
    nf, ns = C.c_uint64(), C.c_uint64()
    L.oracle_families_counts(handle, C.byref(nf), C.byref(ns))
    offs = np.zeros(nf.value + 1, dtype=np.uint64)
    sds = np.zeros((ns.value, 4), dtype=np.uint64)
    L.oracle_families_copy(handle, _ptr(offs), _ptr(sds))
    L.oracle_families_free(handle)
    return offs, sds


def families_to_list(offs: np.ndarray, sds: np.ndarray) -> Families:
    out = []
    for f in range(len(offs) - 1):
        out.append([tuple(int(v) for v in sds[j]) for j in range(int(offs[f]), int(offs[f + 1]))])
    return out


@dataclass
class Index:
    """Strand data + SA + Searcher, i.e. what SearchDuplications::run builds at
    src/bin/asgart.rs:141-155 before the timed part starts."""

    text: np.ndarray  # includes the trailing '$' (asgart.rs:430)
    sa: np.ndarray
    searcher: int

    @classmethod
    def build(cls, text, sa: Optional[np.ndarray] = None) -> "Index":
        text = as_text(text)
        if sa is None:
            sa = divsufsort64(text)
        sa = np.ascontiguousarray(sa, dtype=np.int64)
        h = lib().oracle_searcher_new(_ptr(text), len(text), _ptr(sa), len(sa), 0)
        if not h:
            raise MemoryError("oracle_searcher_new")
        return cls(text, sa, h)

    @classmethod
    def build_trim(cls, text, start: int, end: int) -> "Index":
        """`--trim START END` (src/bin/asgart.rs:142-148): the suffix array of data[start..end] + '$',
        every entry shifted by +start; the Searcher is built over it against the FULL text."""
        text = as_text(text)
        sub = np.concatenate([text[start:end], np.frombuffer(b"$", dtype=np.uint8)])
        sa = divsufsort64(sub) + np.int64(start)
        return cls.build(text, sa)

    def close(self):
        if self.searcher:
            lib().oracle_searcher_free(self.searcher)
            self.searcher = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cache_get(self, p8: bytes) -> Tuple[int, int]:
        lo, hi = C.c_uint64(), C.c_uint64()
        buf = as_text(p8)
        rc = lib().oracle_searcher_cache_get(self.searcher, _ptr(buf), C.byref(lo), C.byref(hi))
        if rc != 0:
            raise KeyError(p8)
        return lo.value, hi.value

    def search(self, pattern) -> Tuple[np.ndarray, Tuple[int, int]]:
        """Searcher::search -> (hit starts in SA order, SA slot range)."""
        p = as_text(pattern)
        lo, hi, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L = lib()
        cnt = L.oracle_searcher_search(self.searcher, _ptr(self.text), len(self.text),
                                       _ptr(self.sa), _ptr(p), len(p), None, 0, C.byref(lo),
                                       C.byref(hi), C.byref(b))
        if cnt < 0:
            raise KeyError(bytes(p[:8]))
        out = np.zeros(cnt, dtype=np.uint64)
        L.oracle_searcher_search(self.searcher, _ptr(self.text), len(self.text), _ptr(self.sa),
                                 _ptr(p), len(p), _ptr(out), cnt, None, None, None)
        return out, (lo.value, hi.value)

    def search_duplications(self, needle, needle_offset: int, settings: Settings,
                            stats: Optional[Stats] = None) -> Families:
        """automaton::search_duplications on one prepared needle."""
        nd = as_text(needle)
        h = C.c_void_p()
        rc = lib().oracle_search_duplications(_ptr(nd), len(nd), needle_offset, _ptr(self.text),
                                              len(self.text), _ptr(self.sa), self.searcher, None,
                                              C.byref(settings),
                                              C.byref(stats) if stats is not None else None,
                                              C.byref(h))
        if rc != 0:
            if h:
                lib().oracle_families_free(h)
            raise RuntimeError(f"oracle_search_duplications -> {rc}")
        return families_to_list(*_take_families(h))

    def run_raw(self, chunks: Sequence[Tuple[int, int]], settings: Settings, threads: int = 1,
                stats: Optional[Stats] = None) -> Tuple[np.ndarray, np.ndarray]:
        ch = np.array(chunks, dtype=np.uint64).reshape(-1)
        h = C.c_void_p()
        rc = lib().oracle_run(_ptr(self.text), len(self.text), _ptr(self.sa), self.searcher,
                              _ptr(ch), len(chunks), C.byref(settings), threads, None,
                              C.byref(stats) if stats is not None else None, C.byref(h))
        if rc != 0:
            if h:
                lib().oracle_families_free(h)
            raise RuntimeError(f"oracle_run -> {rc}")
        return _take_families(h)

    def run(self, chunks, settings, threads: int = 1, stats: Optional[Stats] = None) -> Families:
        """SearchDuplications::run body (src/bin/asgart.rs:201-253)."""
        return families_to_list(*self.run_raw(chunks, settings, threads, stats))

    def probe_hits(self, needle, needle_offset: int, settings: Settings):
        """(status[j], row_offsets, hits) for probe j at i=(j+1)*step."""
        nd = as_text(needle)
        nh = C.c_uint64()
        L = lib()
        args = (_ptr(nd), len(nd), needle_offset, _ptr(self.text), len(self.text), _ptr(self.sa),
                self.searcher, C.byref(settings))
        n_probes = L.oracle_probe_hits(*args, None, None, None, C.byref(nh))
        if n_probes < 0:
            raise RuntimeError(f"oracle_probe_hits -> {n_probes}")
        status = np.zeros(n_probes, dtype=np.uint8)
        offs = np.zeros(n_probes + 1, dtype=np.uint64)
        hits = np.zeros(nh.value, dtype=np.uint64)
        L.oracle_probe_hits(*args, _ptr(status), _ptr(offs), _ptr(hits), C.byref(nh))
        return status, offs, hits


def postprocess(strand: np.ndarray, offs: np.ndarray, sds: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """FilterNs -> ReOrder -> ReduceOverlap -> Sort on raw family arrays (src/bin/asgart.rs:738-747)."""
    strand = as_text(strand)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    sds = np.ascontiguousarray(sds, dtype=np.uint64)
    h = C.c_void_p()
    rc = lib().oracle_postprocess(_ptr(strand), _ptr(offs), len(offs) - 1, _ptr(sds), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"oracle_postprocess -> {rc}")
    return _take_families(h)


def levenshtein_identity(strand: np.ndarray, sd, reversed_: bool = False, complemented_: bool = False) -> np.float32:
    """ProtoSD::levenshtein as f32 (src/structs.rs:439-452, src/bin/asgart.rs:108)."""
    strand = as_text(strand)
    f = lib().oracle_levenshtein_identity
    f.restype = C.c_double
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]
    left, right, ll, rl = (int(v) for v in sd)
    return np.float32(f(_ptr(strand), left, right, ll, rl, int(reversed_), int(complemented_)))


def prepare_needle(text: np.ndarray, chunk: Tuple[int, int], settings: Settings) -> np.ndarray:
    """Needle preparation of src/bin/asgart.rs:206-218."""
    nd = np.ascontiguousarray(text[chunk[0]:chunk[0] + chunk[1]])
    if settings.complement:
        nd = complemented(nd)
    if settings.reverse:
        nd = np.ascontiguousarray(nd[::-1])
    return nd
