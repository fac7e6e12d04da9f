"""asgart_amd -- host-side mirror of ASGART's search-core interface over libasgart_hip.so.

The product is the HIP library (asgart_amd/csrc -> asgart_amd/libasgart_hip.so, C ABI in
include/asgart_hip.h).  This module only binds it with ctypes and mirrors the names of the
reference's operator interface for the path (reference src/bin/asgart.rs:28-31,114-259,
src/searcher.rs:94-180, src/structs.rs:36-58,418-429):

    RunSettings, ProtoSD, Strand, Searcher (new / search), SearchDuplications (Step.run)

There is no CPU fallback: importing works without a GPU (so the symbol table can be
checked), but every compute entry point raises AsgartError when the library or a gfx950
device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "AsgartError", "RunSettings", "ProtoSD", "Strand", "Index", "Searcher", "SearchDuplications",
    "load_library", "library_path", "ABI_SYMBOLS", "sa_build64", "search_duplications_multi", "merge_shards",
]

_HERE = os.path.dirname(os.path.abspath(__file__))

ABI_SYMBOLS = (
    "asgart_sa_build64", "asgart_index_create", "asgart_index_destroy", "asgart_index_prepare",
    "asgart_search_duplications", "asgart_search_duplications_shard", "asgart_families_counts",
    "asgart_families_copy", "asgart_families_free", "asgart_searcher_cache_get",
    "asgart_searcher_search", "asgart_sa_read", "asgart_probe_hits", "asgart_get_stats",
    "asgart_last_error", "asgart_version", "asgart_compute_scores", "asgart_index_set_option",
    "asgart_index_check_sa", "asgart_index_create_trim", "asgart_index_clone",
    "asgart_search_duplications_multi", "asgart_search_duplications_ex", "asgart_search_duplications_passes",
    "asgart_search_duplications_passes_shard", "asgart_families_keys",
    "asgart_index_export", "asgart_index_create_device", "asgart_trim_cache", "asgart_post_process",
    "asgart_debug_dump_stacks", "asgart_prepare_data",
)


def merge_shards(parts) -> Tuple[np.ndarray, np.ndarray]:
    """(fam_offsets, sds, keys) of every shard of one call -> (fam_offsets, sds) of the whole call: families ordered
    by key (segment start probe, family ordinal) == reference order.  What rank 0 does after the gather."""
    keys = np.concatenate([p[2] for p in parts]) if parts else np.zeros(0, np.uint64)
    if len(keys) == 0:
        return np.zeros(1, np.uint64), np.zeros((0, 4), np.uint64)
    starts = np.concatenate([p[0][:-1].astype(np.int64) + b for p, b in
                             zip(parts, np.cumsum([0] + [len(p[1]) for p in parts[:-1]]))])
    lens = np.concatenate([np.diff(p[0].astype(np.int64)) for p in parts])
    sds = np.concatenate([p[1] for p in parts])
    order = np.argsort(keys, kind="stable")
    lens_o, starts_o = lens[order], starts[order]
    offs = np.concatenate([[0], np.cumsum(lens_o)]).astype(np.uint64)
    idx = np.repeat(starts_o - offs[:-1].astype(np.int64), lens_o) + np.arange(int(offs[-1]), dtype=np.int64)
    return offs, sds[idx]


class AsgartError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libasgart_hip error {code}: {msg}")
        self.code = code


def library_path() -> str:
    # ASGART_LIB selects a diagnostic build of the same library (profiling counters)
    return os.environ.get("ASGART_LIB") or os.path.join(_HERE, "libasgart_hip.so")


class _Settings(C.Structure):
    _fields_ = [
        ("probe_size", C.c_uint64),
        ("max_gap_size", C.c_uint32),
        ("min_duplication_length", C.c_uint64),
        ("max_cardinality", C.c_uint64),
        ("reverse", C.c_uint8),
        ("complement", C.c_uint8),
    ]


class Stats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_search", C.c_double), ("ms_scan", C.c_double),
                ("ms_fill", C.c_double), ("ms_extend", C.c_double)] + [
        (n, C.c_uint64) for n in (
            "probes_total", "probes_n_skipped", "probes_searched", "probes_card_skipped",
            "probes_with_hits", "raw_hits", "filtered_hits", "segments", "families", "proto_sds",
            "bisect_steps", "search_launches", "overflow_segments")] + [("ms_extend_tier2", C.c_double), ("heavy_segments", C.c_uint64), ("ms_probe_count", C.c_double),
        ("search_bytes", C.c_uint64), ("probes_filter_rejected", C.c_uint64), ("search_bytes_wide_loads", C.c_uint64),
        ("ms_longest_tier", C.c_double), ("passes", C.c_uint64), ("ms_longest_segment", C.c_double),
        ("split_segments", C.c_uint64), ("split_refused", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib: Optional[C.CDLL] = None

# see the load-time note in csrc/index.hip: one hardware queue per tier stream (effective before the first HIP call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def load_library() -> C.CDLL:
    """dlopen libasgart_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise AsgartError(-3, f"{path} not built: run `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.asgart_sa_build64.argtypes = [vp, vp, C.c_int64]
    L.asgart_sa_build64.restype = C.c_int32
    L.asgart_index_create.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_int32, C.POINTER(vp)]
    L.asgart_index_create.restype = C.c_int32
    L.asgart_index_create_trim.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                           C.POINTER(vp)]
    L.asgart_index_create_trim.restype = C.c_int32
    L.asgart_index_clone.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.asgart_index_clone.restype = C.c_int32
    L.asgart_index_export.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32)]
    L.asgart_index_export.restype = C.c_int32
    L.asgart_index_create_device.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.asgart_index_create_device.restype = C.c_int32
    L.asgart_trim_cache.argtypes = [C.c_int32]
    L.asgart_trim_cache.restype = C.c_int64
    L.asgart_prepare_data.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, C.c_int64, C.POINTER(C.c_int64),
                                      C.POINTER(vp)]
    L.asgart_prepare_data.restype = C.c_int32
    L.asgart_debug_dump_stacks.argtypes = []
    L.asgart_debug_dump_stacks.restype = C.c_int32
    L.asgart_post_process.argtypes = [vp, vp, C.c_uint64, vp, C.c_int32, C.POINTER(vp)]
    L.asgart_post_process.restype = C.c_int32
    L.asgart_search_duplications_multi.argtypes = [C.POINTER(vp), C.c_int32, vp, C.c_int64, C.POINTER(_Settings), vp,
                                                   C.POINTER(vp)]
    L.asgart_search_duplications_multi.restype = C.c_int32
    L.asgart_index_destroy.argtypes = [vp]
    L.asgart_index_destroy.restype = None
    L.asgart_index_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.asgart_index_set_option.restype = C.c_int32
    L.asgart_index_check_sa.argtypes = [vp]
    L.asgart_index_check_sa.restype = C.c_int64
    L.asgart_index_prepare.argtypes = [vp, C.c_uint64]
    L.asgart_index_prepare.restype = C.c_int32
    L.asgart_search_duplications.argtypes = [vp, vp, C.c_int64, C.POINTER(_Settings), vp,
                                             C.POINTER(vp)]
    L.asgart_search_duplications.restype = C.c_int32
    L.asgart_search_duplications_shard.argtypes = [vp, vp, C.c_int64, C.POINTER(_Settings),
                                                   C.c_int32, C.c_int32, C.POINTER(vp)]
    L.asgart_search_duplications_shard.restype = C.c_int32
    L.asgart_search_duplications_ex.argtypes = [vp, vp, C.c_int64, C.POINTER(_Settings), C.c_int32, C.c_int32, vp,
                                                C.POINTER(vp)]
    L.asgart_search_duplications_ex.restype = C.c_int32
    L.asgart_search_duplications_passes.argtypes = [vp, vp, C.c_int64, C.POINTER(_Settings), C.c_int32, C.POINTER(vp)]
    L.asgart_search_duplications_passes.restype = C.c_int32
    L.asgart_families_keys.argtypes = [vp, vp]
    L.asgart_families_keys.restype = None
    L.asgart_search_duplications_passes_shard.argtypes = [vp, vp, C.c_int64, C.POINTER(_Settings), C.c_int32, C.c_int32,
                                                          C.c_int32, C.POINTER(vp)]
    L.asgart_search_duplications_passes_shard.restype = C.c_int32
    L.asgart_families_counts.argtypes = [vp, u64p, u64p]
    L.asgart_families_counts.restype = None
    L.asgart_families_copy.argtypes = [vp, vp, vp]
    L.asgart_families_copy.restype = None
    L.asgart_families_free.argtypes = [vp]
    L.asgart_families_free.restype = None
    L.asgart_searcher_cache_get.argtypes = [vp, vp, C.c_int64, vp, vp]
    L.asgart_searcher_cache_get.restype = C.c_int32
    L.asgart_searcher_search.argtypes = [vp, vp, C.c_int64, C.c_uint64, vp, vp]
    L.asgart_searcher_search.restype = C.c_int32
    L.asgart_sa_read.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
    L.asgart_sa_read.restype = C.c_int32
    L.asgart_compute_scores.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
    L.asgart_compute_scores.restype = C.c_int32
    L.asgart_probe_hits.argtypes = [vp, vp, C.c_int64, C.POINTER(_Settings), vp, vp, vp, u64p]
    L.asgart_probe_hits.restype = C.c_int64
    L.asgart_get_stats.argtypes = [vp, C.c_uint32, C.POINTER(Stats)]
    L.asgart_get_stats.restype = C.c_int32
    L.asgart_last_error.argtypes = []
    L.asgart_last_error.restype = C.c_char_p
    L.asgart_version.argtypes = []
    L.asgart_version.restype = C.c_char_p
    _lib = L
    return L


def _check(rc: int):
    if rc < 0:
        raise AsgartError(int(rc), load_library().asgart_last_error().decode(errors="replace"))


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as_u8(seq) -> np.ndarray:
    if isinstance(seq, np.ndarray):
        if seq.dtype != np.uint8:
            raise TypeError("text must be uint8")
        return np.ascontiguousarray(seq)
    if isinstance(seq, str):
        seq = seq.encode()
    return np.frombuffer(bytes(seq), dtype=np.uint8).copy()


@dataclass
class RunSettings:
    """reference src/structs.rs:36-58 (fields that reach the search path)."""

    probe_size: int = 20
    max_gap_size: int = 120  # gap_size + probe_size, src/bin/asgart.rs:681
    min_duplication_length: int = 1000
    max_cardinality: int = 500
    reverse: bool = False
    complement: bool = False
    skip_masked: bool = False
    trim: Optional[Tuple[int, int]] = None

    @classmethod
    def from_cli(cls, k=20, gap=100, min_length=1000, max_cardinality=500, reverse=False,
                 complement=False, skip_masked=False) -> "RunSettings":
        """`asgart -k K -g G --min-length M --max-cardinality C [-R] [-C] [-S]`
        (reference src/bin/asgart.rs:564-631,677-693)."""
        return cls(k, gap + k, min_length, max_cardinality, reverse, complement, skip_masked)

    def _c(self) -> _Settings:
        return _Settings(self.probe_size, self.max_gap_size, self.min_duplication_length,
                         self.max_cardinality, int(self.reverse), int(self.complement))


@dataclass
class ProtoSD:
    """reference src/structs.rs:418-429"""

    left: int
    right: int
    left_length: int
    right_length: int
    identity: float = 0.0
    reversed: bool = False
    complemented: bool = False

    def as_tuple(self):
        return (self.left, self.right, self.left_length, self.right_length)


ProtoSDsFamily = List[ProtoSD]


@dataclass
class Strand:
    """reference src/bin/asgart.rs:267-271 (`data` ends with '$', :430)."""

    file_names: str
    data: np.ndarray
    map: list = field(default_factory=list)


class Index:
    """Device-resident text + suffix array + search structures (one per GPU).

    Stands for what SearchDuplications::run builds before its timed part:
    `r_divsufsort(&strand.data)` and `Searcher::new(&strand.data, &sa, 0)`
    (reference src/bin/asgart.rs:141-155).
    """

    def __init__(self, text, sa: Optional[np.ndarray] = None, device: int = 0,
                 trim: Optional[Tuple[int, int]] = None):
        """trim=(start, end): the `--trim` variant (reference src/bin/asgart.rs:142-148): the suffix array
        covers data[start..end] + '$' only (entries shifted by +start) and the whole text is searched
        against it.  `sa` is then that shifted array (end - start + 1 entries) or None."""
        L = load_library()
        self.text = _as_u8(text)
        self.n = len(self.text)
        self._h = C.c_void_p()
        self.trim = trim
        sa_arr = None
        want = len(self.text) if trim is None else trim[1] - trim[0] + 1
        if sa is not None:
            sa_arr = np.ascontiguousarray(sa, dtype=np.int64)
            if len(sa_arr) != want:
                raise ValueError("suffix array length != text length (or the trimmed window + 1)")
        if trim is None:
            _check(L.asgart_index_create(_ptr(self.text), len(self.text), _ptr(sa_arr),
                                         want if sa_arr is not None else 0, device, C.byref(self._h)))
        else:
            _check(L.asgart_index_create_trim(_ptr(self.text), len(self.text), _ptr(sa_arr),
                                              want if sa_arr is not None else 0, int(trim[0]), int(trim[1]),
                                              device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load_library().asgart_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def clone(self, device: int = 0) -> "Index":
        """A replica of this index on `device` (text + suffix array copied device to device)."""
        other = Index.__new__(Index)
        other.text = self.text
        other.n = self.n
        other.trim = self.trim
        other._h = C.c_void_p()
        _check(load_library().asgart_index_clone(self._h, device, C.byref(other._h)))
        return other

    def export(self) -> Tuple[int, int, int]:
        """(device address of the text, of the suffix array, bytes per suffix-array entry): what a one-process-per-GPU
        host broadcasts to the other ranks (asgart_index_export; multi.replicate_index)."""
        t, a, w = C.c_void_p(), C.c_void_p(), C.c_int32()
        _check(load_library().asgart_index_export(self._h, C.byref(t), C.byref(a), C.byref(w)))
        return int(t.value), int(a.value), int(w.value)

    @classmethod
    def from_device(cls, d_text: int, n: int, d_sa: int, sa_entry_bytes: int, device: int = 0, text=None) -> "Index":
        """A replica from text and suffix array already in this GPU's memory (asgart_index_create_device: copied)."""
        self = cls.__new__(cls)
        self.text = text
        self.n = int(n)
        self.trim = None
        self._h = C.c_void_p()
        _check(load_library().asgart_index_create_device(C.c_void_p(d_text), n, C.c_void_p(d_sa), n, sa_entry_bytes,
                                                         device, C.byref(self._h)))
        return self

    def set_option(self, name: str, value: int):
        """Tuning / test option (include/asgart_hip.h: asgart_index_set_option)."""
        _check(load_library().asgart_index_set_option(self._h, name.encode(), int(value)))

    def check_sa(self) -> int:
        """GPU verifier of the resident suffix array: number of violating slots (0 = valid)."""
        r = int(load_library().asgart_index_check_sa(self._h))
        _check(r)
        return r

    def prepare(self, probe_size: int):
        _check(load_library().asgart_index_prepare(self._h, probe_size))

    def sa_read(self, lo: int, hi: int) -> np.ndarray:
        out = np.empty(max(0, hi - lo), dtype=np.int64)
        _check(load_library().asgart_sa_read(self._h, lo, hi, _ptr(out)))
        return out

    def compute_scores(self, sds: np.ndarray, reversed_: bool = False, complemented: bool = False) -> np.ndarray:
        """ComputeScore of reference src/bin/asgart.rs:98-112 for an (n, 4) uint64 array of
        (left, right, left_length, right_length): Levenshtein identities as float32."""
        sds = np.ascontiguousarray(sds, dtype=np.uint64).reshape(-1, 4)
        out = np.empty(len(sds), dtype=np.float32)
        _check(load_library().asgart_compute_scores(self._h, _ptr(sds), len(sds), int(reversed_),
                                                    int(complemented), _ptr(out)))
        return out

    def post_process(self, offs: np.ndarray, sds: np.ndarray, threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """FilterNs -> ReOrder -> ReduceOverlap -> Sort (reference src/bin/asgart.rs:738-747) on raw family arrays as
        search_duplications_raw returns them -> the same form (asgart_post_process: N counts on the GPU, the reduction
        on host threads)."""
        L = load_library()
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        sds = np.ascontiguousarray(sds, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        _check(L.asgart_post_process(self._h, _ptr(offs), len(offs) - 1, _ptr(sds), threads, C.byref(h)))
        try:
            nf, ns = C.c_uint64(), C.c_uint64()
            L.asgart_families_counts(h, C.byref(nf), C.byref(ns))
            o = np.zeros(nf.value + 1, dtype=np.uint64)
            d = np.zeros((ns.value, 4), dtype=np.uint64)
            L.asgart_families_copy(h, _ptr(o), _ptr(d))
        finally:
            L.asgart_families_free(h)
        return o, d

    def stats(self, flags: int = 2) -> Stats:
        """Statistics of the last search call.  flags: 1 = with the yardstick / accounting passes, 2 (default) = with raw_hits
        (an untimed pass over the call's probes), 0 = what the call itself recorded; (i + 1) << 8 selects call context i."""
        st = Stats()
        _check(load_library().asgart_get_stats(self._h, flags, C.byref(st)))
        return st

    # -- SearchDuplications::run body --------------------------------------
    def search_duplications_raw(self, chunks: Sequence[Tuple[int, int]], settings: RunSettings,
                                shard: int = 0, n_shards: int = 1, progress: Optional[np.ndarray] = None,
                                with_keys: bool = False):
        """-> (fam_offsets[n_fam+1], sds[n_sd,4]) as uint64 arrays.  progress: optional uint64[n_chunks]
        the library writes each chunk's needle offset into when the call's search phases are over (see the header)."""
        L = load_library()
        ch = np.array(chunks, dtype=np.uint64).reshape(-1)
        st = settings._c()
        h = C.c_void_p()
        if progress is not None:
            assert progress.dtype == np.uint64 and len(progress) >= len(chunks)
        if n_shards == 1:
            _check(L.asgart_search_duplications(self._h, _ptr(ch), len(chunks), C.byref(st), _ptr(progress),
                                                C.byref(h)))
        elif progress is None:
            _check(L.asgart_search_duplications_shard(self._h, _ptr(ch), len(chunks), C.byref(st),
                                                      shard, n_shards, C.byref(h)))
        else:
            _check(L.asgart_search_duplications_ex(self._h, _ptr(ch), len(chunks), C.byref(st), shard, n_shards,
                                                   _ptr(progress), C.byref(h)))
        try:
            nf, ns = C.c_uint64(), C.c_uint64()
            L.asgart_families_counts(h, C.byref(nf), C.byref(ns))
            offs = np.zeros(nf.value + 1, dtype=np.uint64)
            sds = np.zeros((ns.value, 4), dtype=np.uint64)
            L.asgart_families_copy(h, _ptr(offs), _ptr(sds))
            keys = np.zeros(nf.value, dtype=np.uint64)
            if with_keys:
                L.asgart_families_keys(h, _ptr(keys))
        finally:
            L.asgart_families_free(h)
        return (offs, sds, keys) if with_keys else (offs, sds)

    def search_duplications_passes(self, chunks: Sequence[Tuple[int, int]], settings: Sequence[RunSettings],
                                   shard: int = 0, n_shards: int = 1, with_keys: bool = False) -> List[tuple]:
        """Several passes (one RunSettings each, e.g. the direct and the -RC run) in one call; the library pipelines
        them itself.  -> [(fam_offsets, sds)] in the order of `settings`, each as search_duplications_raw returns it."""
        L = load_library()
        ch = np.array(chunks, dtype=np.uint64).reshape(-1)
        n = len(settings)
        sts = (_Settings * max(n, 1))(*[s._c() for s in settings])
        hs = (C.c_void_p * max(n, 1))()
        if n_shards == 1:
            _check(L.asgart_search_duplications_passes(self._h, _ptr(ch), len(chunks), sts, n, hs))
        else:
            _check(L.asgart_search_duplications_passes_shard(self._h, _ptr(ch), len(chunks), sts, n, shard, n_shards, hs))
        out = []
        try:
            for j in range(n):
                nf, ns = C.c_uint64(), C.c_uint64()
                L.asgart_families_counts(hs[j], C.byref(nf), C.byref(ns))
                offs = np.zeros(nf.value + 1, dtype=np.uint64)
                sds = np.zeros((ns.value, 4), dtype=np.uint64)
                L.asgart_families_copy(hs[j], _ptr(offs), _ptr(sds))
                keys = np.zeros(nf.value, dtype=np.uint64)
                if with_keys:
                    L.asgart_families_keys(hs[j], _ptr(keys))
                out.append((offs, sds, keys) if with_keys else (offs, sds))
        finally:
            for j in range(n):
                L.asgart_families_free(hs[j])
        return out

    def probe_hits(self, chunks: Sequence[Tuple[int, int]], settings: RunSettings):
        """Per-probe filtered hits for all chunks: (status, row_offsets, hits)."""
        L = load_library()
        ch = np.array(chunks, dtype=np.uint64).reshape(-1)
        st = settings._c()
        nh = C.c_uint64()
        n_probes = L.asgart_probe_hits(self._h, _ptr(ch), len(chunks), C.byref(st), None, None,
                                       None, C.byref(nh))
        _check(n_probes)
        status = np.zeros(n_probes, dtype=np.uint8)
        offs = np.zeros(n_probes + 1, dtype=np.uint64)
        hits = np.zeros(nh.value, dtype=np.uint64)
        _check(L.asgart_probe_hits(self._h, _ptr(ch), len(chunks), C.byref(st), _ptr(status),
                                   _ptr(offs), _ptr(hits), C.byref(nh)))
        return status, offs, hits


def search_duplications_multi(indices: Sequence[Index], chunks: Sequence[Tuple[int, int]],
                              settings: RunSettings) -> Tuple[np.ndarray, np.ndarray]:
    """asgart_search_duplications_multi: one shard per index replica (one per GPU), one host thread each,
    results concatenated in shard order -> (fam_offsets, sds) like Index.search_duplications_raw."""
    L = load_library()
    ch = np.array(chunks, dtype=np.uint64).reshape(-1)
    st = settings._c()
    arr = (C.c_void_p * len(indices))(*[i._h for i in indices])
    h = C.c_void_p()
    _check(L.asgart_search_duplications_multi(arr, len(indices), _ptr(ch), len(chunks), C.byref(st), None,
                                              C.byref(h)))
    try:
        nf, ns = C.c_uint64(), C.c_uint64()
        L.asgart_families_counts(h, C.byref(nf), C.byref(ns))
        offs = np.zeros(nf.value + 1, dtype=np.uint64)
        sds = np.zeros((ns.value, 4), dtype=np.uint64)
        L.asgart_families_copy(h, _ptr(offs), _ptr(sds))
    finally:
        L.asgart_families_free(h)
    return offs, sds


def trim_cache(device: int = 0) -> int:
    """Device memory the library holds for reuse goes back to the device (asgart_trim_cache); -> bytes released."""
    r = int(load_library().asgart_trim_cache(device))
    _check(r)
    return r


def sa_build64(text) -> np.ndarray:
    """`r_divsufsort` (reference src/bin/asgart.rs:473-479) on the GPU: the suffix array of
    `text` as int64, via asgart_sa_build64 (signature-identical to divsufsort64)."""
    t = _as_u8(text)
    sa = np.empty(len(t), dtype=np.int64)
    _check(load_library().asgart_sa_build64(_ptr(t), _ptr(sa), len(t)))
    return sa


class Searcher:
    """reference src/searcher.rs:94-180: `Searcher::new(dna, sa, offset)` and `search`."""

    def __init__(self, index: Index, offset: int = 0):
        self.index = index
        self.offset = offset

    @classmethod
    def new(cls, dna, sa: np.ndarray, offset: int = 0, device: int = 0) -> "Searcher":
        return cls(Index(dna, sa, device), offset)

    def cache(self, patterns8: Sequence[bytes]) -> List[Tuple[int, int]]:
        """8-mer -> (start, end) SA interval, the entries of Searcher.cache."""
        pats = np.frombuffer(b"".join(bytes(p) for p in patterns8), dtype=np.uint8).copy()
        n = len(patterns8)
        lo = np.zeros(n, dtype=np.uint64)
        hi = np.zeros(n, dtype=np.uint64)
        _check(load_library().asgart_searcher_cache_get(self.index._h, _ptr(pats), n, _ptr(lo),
                                                        _ptr(hi)))
        return [(int(a), int(b)) for a, b in zip(lo, hi)]

    def search_ranges(self, patterns: Sequence[bytes]) -> List[Tuple[int, int]]:
        k = len(patterns[0])
        pats = np.frombuffer(b"".join(bytes(p) for p in patterns), dtype=np.uint8).copy()
        n = len(patterns)
        lo = np.zeros(n, dtype=np.uint64)
        hi = np.zeros(n, dtype=np.uint64)
        _check(load_library().asgart_searcher_search(self.index._h, _ptr(pats), n, k, _ptr(lo),
                                                     _ptr(hi)))
        return [(int(a), int(b)) for a, b in zip(lo, hi)]

    def search(self, pattern: bytes) -> List[Tuple[int, int]]:
        """-> Segments (start, end) in SA order, like Searcher::search."""
        (lo, hi), = self.search_ranges([pattern])
        starts = self.index.sa_read(lo, hi)
        k = len(pattern)
        return [(self.offset + int(x), self.offset + int(x) + k) for x in starts]


class SearchDuplications:
    """The `Step` of reference src/bin/asgart.rs:114-259.

    `run(input, strand)` ignores `input` (as the reference does, :137) and returns
    the proto-duplication families found in `chunks_to_process`.
    """

    def __init__(self, chunks_to_process: Sequence[Tuple[int, int]],
                 trim: Optional[Tuple[int, int]], settings: RunSettings,
                 suffix_array: Optional[np.ndarray] = None, device: int = 0,
                 index: Optional[Index] = None):
        self.trim = trim
        self.chunks_to_process = list(chunks_to_process)
        self.settings = settings
        self.suffix_array = suffix_array
        self.device = device
        self.index = index

    def name(self) -> str:
        return "Looking for proto-duplications"

    def run(self, _input: List[ProtoSDsFamily], strand: Strand) -> List[ProtoSDsFamily]:
        index = self.index or Index(strand.data, self.suffix_array, self.device, trim=self.trim)
        try:
            offs, sds = index.search_duplications_raw(self.chunks_to_process, self.settings)
        finally:
            if self.index is None:
                index.close()
        s = self.settings
        out: List[ProtoSDsFamily] = []
        for f in range(len(offs) - 1):
            out.append([ProtoSD(int(r[0]), int(r[1]), int(r[2]), int(r[3]), 0.0, s.reverse,
                                s.complement) for r in sds[int(offs[f]):int(offs[f + 1])]])
        return out
