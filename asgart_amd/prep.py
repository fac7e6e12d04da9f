"""Host-side input preparation: what `prepare_data` hands to the search step.

Mirrors reference src/bin/asgart.rs:273-471 for in-memory records (FASTA parsing
itself is `read_records`): per-record normalisation (:289-301), chunking at
N-runs longer than 5000 (:317-366), concatenation with per-record chunk offsets
(:375-395) and the final '$' (:430).  prepare_records is the numpy statement of it (host only: what the CPU tests
and the oracle comparisons use); prepare_records_gpu is the product path: the same step behind the C ABI
(asgart_prepare_data: normalisation and N-run detection as kernels over the uploaded bytes, the index built from
the same device buffer).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import numpy as np

N_RUN_THRESHOLD = 5000  # reference src/bin/asgart.rs:326

_UPPER = np.arange(256, dtype=np.uint8)
_UPPER[ord("a"):ord("z") + 1] -= 32
_KEEP = np.full(256, ord("N"), dtype=np.uint8)
for _c in b"ATGCN":
    _KEEP[_c] = _c
_NORM_PLAIN = _KEEP[_UPPER]  # upper-case first, then non-alphabet -> N
_NORM_MASKED = _KEEP.copy()  # lower-case (masked) letters are not in ALPHABET -> N


def normalise(seq: np.ndarray, skip_masked: bool) -> np.ndarray:
    """reference src/bin/asgart.rs:289-301."""
    return (_NORM_MASKED if skip_masked else _NORM_PLAIN)[seq]


def find_chunks_to_process(strand: np.ndarray) -> List[Tuple[int, int]]:
    """reference src/bin/asgart.rs:317-366: maximal pieces between N-runs > 5000."""
    n = len(strand)
    isn = (strand == ord("N")) | (strand == ord("n"))
    edges = np.diff(np.concatenate(([0], isn.view(np.int8), [0])))
    starts = np.flatnonzero(edges == 1)
    ends = np.flatnonzero(edges == -1)
    long_runs = (ends - starts) > N_RUN_THRESHOLD
    cut_s, cut_e = starts[long_runs], ends[long_runs]
    piece_s = np.concatenate(([0], cut_e))
    piece_e = np.concatenate((cut_s, [n]))
    chunks = [(int(a), int(b - a)) for a, b in zip(piece_s, piece_e) if b > a]
    if not chunks:
        chunks = [(0, n)]
    return chunks


@dataclass
class Start:
    """reference src/structs.rs:60-65"""

    name: str
    position: int
    length: int


@dataclass
class Prepared:
    data: np.ndarray                      # concatenated, normalised, '$'-terminated
    chunks: List[Tuple[int, int]]         # global (start, len)
    map: List[Start]


def prepare_records(records: Sequence[Tuple[str, np.ndarray]], skip_masked: bool = False) -> Prepared:
    """prepare_data for records already in memory (one or several files' worth)."""
    parts, chunks, starts = [], [], []
    offset = 0
    for name, seq in records:
        seq = normalise(np.asarray(seq, dtype=np.uint8), skip_masked)
        chunks.extend((offset + s, l) for s, l in find_chunks_to_process(seq))
        starts.append(Start(name, offset, len(seq)))
        offset += len(seq)
        parts.append(seq)
    parts.append(np.frombuffer(b"$", dtype=np.uint8))
    return Prepared(np.concatenate(parts), chunks, starts)


def prepare_records_gpu(records: Sequence[Tuple[str, np.ndarray]], skip_masked: bool = False, device: int = 0,
                        want_text: bool = True, want_index: bool = True):
    """prepare_data through the library (asgart_prepare_data): -> (Prepared, Index or None).  The raw records are
    uploaded once; normalisation, chunking and the suffix sort run on the GPU; with want_text = False the prepared strand
    stays on the device (Prepared.data is None) -- the search, the post-processing and the JSON need only the index, the
    chunks and the map."""
    import ctypes as C

    from . import Index, _check, _ptr, load_library

    L = load_library()
    seqs = [np.ascontiguousarray(np.asarray(seq, dtype=np.uint8)) for _, seq in records]
    n_rec = len(seqs)
    ptrs = (C.c_void_p * max(n_rec, 1))(*[s_.ctypes.data for s_ in seqs])
    lens = np.array([len(s_) for s_ in seqs], dtype=np.uint64)
    total = int(lens.sum())
    text = np.empty(total + 1, dtype=np.uint8) if want_text else None
    cap = 1 << 16
    h = C.c_void_p()
    while True:
        chunks = np.zeros((cap, 2), dtype=np.uint64)
        nc = C.c_int64()
        rc = L.asgart_prepare_data(ptrs, _ptr(lens), n_rec, 1 if skip_masked else 0, device, _ptr(text), _ptr(chunks), cap,
                                   C.byref(nc), C.byref(h) if want_index else None)
        if rc == -4 and nc.value > cap:   # ASGART_E_CAP: more chunks than room
            cap = int(nc.value)
            continue
        _check(rc)
        break
    starts, offset = [], 0
    for (name, _), ln in zip(records, lens.tolist()):
        starts.append(Start(name, offset, int(ln)))
        offset += int(ln)
    pr = Prepared(text, [(int(a), int(b)) for a, b in chunks[:nc.value]], starts)
    idx = None
    if want_index:
        idx = Index.__new__(Index)
        idx.text, idx.n, idx.trim, idx._h = text, total + 1, None, h
    return pr, idx


def validate_trim(trim, strand_len: int):
    """The --trim checks of prepare_data, reference src/bin/asgart.rs:432-463 (`strand_len` counts the
    final '$'): a stop past the data is clamped to the '$', an empty or out-of-range window disables
    trimming.  -> (start, stop) or None."""
    if trim is None:
        return None
    shift, stop = int(trim[0]), int(trim[1])
    if stop >= strand_len:
        stop = strand_len - 1
    if stop <= shift or shift >= strand_len:
        return None
    return (shift, stop)


def read_records(path: str) -> Iterable[Tuple[str, np.ndarray]]:
    """Minimal FASTA reader (id = header up to the first whitespace), standing in
    for bio::io::fasta::Reader at reference src/bin/asgart.rs:282-288."""
    name, buf = None, []
    with open(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    yield name, np.frombuffer(b"".join(buf), dtype=np.uint8)
                hdr = line[1:].split()
                name = hdr[0].decode() if hdr else ""
                buf = []
            elif name is not None:
                buf.append(line)
    if name is not None:
        yield name, np.frombuffer(b"".join(buf), dtype=np.uint8)
