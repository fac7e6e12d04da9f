"""Post-processing steps that follow the search step, and the RunResult JSON surface.

Host code mirroring the reference's `Step`s in the order fixed at src/bin/asgart.rs:738-747:
FilterNs (:81-96, ProtoSD::n_content src/structs.rs:454-467), ReOrder (:33-51), ReduceOverlap
(:67-79, helpers :481-562), Sort (:53-65); then ProtoSD -> SD with chromosome lookup (:770-821) and
the JSON exporter (src/exporters.rs:12-25, struct field order of src/structs.rs:36-58,60-98,471-493).
ComputeScore (--compute-score, src/bin/asgart.rs:98-112) is the one step that is not cheap: it runs on
the GPU (`asgart_compute_scores`, exact Levenshtein by anti-diagonal DP) and needs the index.

These steps are cheap host work in the reference as well; nothing here
touches the GPU.  All quirks are kept (see the comments), tests/test_postprocess.py checks the chain
against the oracle's C restatement.
"""
from __future__ import annotations

import json
from typing import List, Optional, Sequence

import numpy as np

from . import ProtoSD, ProtoSDsFamily, RunSettings
from .prep import Start


class FilterNs:
    def name(self) -> str:
        return "Filtering uncertain duplications"

    def run(self, families: List[ProtoSDsFamily], strand) -> List[ProtoSDsFamily]:
        data = strand.data
        is_n = ((data == ord("N")) | (data == ord("n")))
        csum = np.concatenate(([0], np.cumsum(is_n, dtype=np.int64)))

        def n_content(sd: ProtoSD) -> np.float32:
            # inclusive ranges [p ..= p+len], divided by len, in f32 (src/structs.rs:454-467)
            cl = csum[sd.left + sd.left_length + 1] - csum[sd.left]
            cr = csum[sd.right + sd.right_length + 1] - csum[sd.right]
            a = np.float32(cl) / np.float32(sd.left_length)
            b = np.float32(cr) / np.float32(sd.right_length)
            return max(a, b)

        out = []
        for fam in families:
            kept = [sd for sd in fam if n_content(sd) <= np.float32(0.2)]
            if kept:
                out.append(kept)
        return out


class ReOrder:
    def name(self) -> str:
        return "Re-ordering"

    def run(self, families, _strand):
        for fam in families:
            for sd in fam:
                if sd.left > sd.right:  # positions are swapped, the lengths are NOT (:42-46)
                    sd.left, sd.right = sd.right, sd.left
        return families


def _subsegment(xs, xl, ys, yl) -> bool:
    return xs >= ys and xs + xl <= ys + yl


def _overlap(xs, xl, ys, yl) -> bool:
    xe, ye = xs + xl, ys + yl
    return (ys <= xs <= ye and xe >= ye) or (xs <= ys <= xe and ye >= xe)


def _merge(x: ProtoSD, y: ProtoSD) -> ProtoSD:
    # src/bin/asgart.rs:497-513: x contributes left_length to BOTH arms, y right_length to both
    nl = min(x.left, y.left)
    ll = max(x.left + x.left_length, y.left + y.right_length) - nl
    nr = min(x.right, y.right)
    rl = max(x.right + x.left_length, y.right + y.right_length) - nr
    return ProtoSD(nl, nr, ll, rl, 0.0, x.reversed, x.complemented)


def _reduce(result: Sequence[ProtoSD]) -> List[ProtoSD]:
    news: List[ProtoSD] = []
    for x in result:
        for y in news:
            if (_subsegment(x.left, x.left_length, y.left, y.left_length)
                    and _subsegment(x.right, x.right_length, y.right, y.right_length)):
                break
            if (_subsegment(y.left, y.left_length, x.left, x.left_length)
                    and _subsegment(y.right, y.right_length, x.right, x.right_length)):
                y.left, y.right, y.left_length, y.right_length = x.left, x.right, x.left_length, x.right_length
                break
            if (_overlap(x.left, x.left_length, y.left, y.left_length)
                    and _overlap(x.right, x.right_length, y.right, y.right_length)):
                z = _merge(x, y)
                y.left, y.right, y.left_length, y.right_length = z.left, z.right, z.left_length, z.right_length
                break
        else:
            news.append(ProtoSD(x.left, x.right, x.left_length, x.right_length, x.identity,
                                x.reversed, x.complemented))
    return news


class ReduceOverlap:
    def name(self) -> str:
        return "Reducing overlap"

    def run(self, families, _strand):
        out = []
        for fam in families:
            old = len(fam)
            news = _reduce(fam)
            while len(news) < old:
                old = len(news)
                news = _reduce(news)
            out.append(news)
        return out


class ComputeScore:
    """`sd.identity = sd.levenshtein(&strand.data) as f32` for every duplication (src/bin/asgart.rs:98-112,
    src/structs.rs:439-452), on the GPU.  `index` is the asgart_amd.Index of the strand."""

    def __init__(self, index):
        self.index = index

    def name(self) -> str:
        return "Computing Levenshtein distance"

    def run(self, families, _strand):
        flat = [sd for fam in families for sd in fam]
        # the flags are constant per run (src/bin/asgart.rs:245-247) but ReduceOverlap keeps x's: group
        for key in {(sd.reversed, sd.complemented) for sd in flat}:
            group = [sd for sd in flat if (sd.reversed, sd.complemented) == key]
            arr = np.array([sd.as_tuple() for sd in group], dtype=np.uint64).reshape(-1, 4)
            ident = self.index.compute_scores(arr, key[0], key[1])
            for sd, v in zip(group, ident):
                sd.identity = float(v)
        return families


class Sort:
    def name(self) -> str:
        return "Sorting"

    def run(self, families, _strand):
        for fam in families:
            fam.sort(key=lambda sd: sd.left)  # stable, like slice::sort_by
        return families


def post_process(families: List[ProtoSDsFamily], strand, index=None, compute_score: bool = False) -> List[ProtoSDsFamily]:
    """The steps after SearchDuplications, in the reference's order (src/bin/asgart.rs:738-747);
    ComputeScore only with compute_score (it needs the GPU index)."""
    steps = [FilterNs(), ReOrder(), ReduceOverlap()]
    if compute_score:
        if index is None:
            raise ValueError("compute_score needs the asgart_amd.Index of the strand")
        steps.append(ComputeScore(index))
    steps.append(Sort())
    for step in steps:
        families = step.run(families, strand)
    return families


# ---- result surface -------------------------------------------------------------------------
def _find_chr_by_pos(starts: Sequence[Start], pos: int) -> Optional[Start]:
    for c in starts:  # first match, linear scan (src/structs.rs:85-90)
        if c.position <= pos < c.position + c.length:
            return c
    return None


def run_result(families: List[ProtoSDsFamily], strand, settings: RunSettings) -> dict:
    """RunResult as a dict in serde field order (src/bin/asgart.rs:770-821, src/structs.rs)."""
    fams = []
    for fam in families:
        out = []
        for sd in fam:
            cl, cr = _find_chr_by_pos(strand.map, sd.left), _find_chr_by_pos(strand.map, sd.right)
            out.append({
                "chr_left": cl.name if cl else "unknown",
                "chr_right": cr.name if cr else "unknown",
                "global_left_position": sd.left,
                "global_right_position": sd.right,
                "chr_left_position": sd.left - (cl.position if cl else 0),
                "chr_right_position": sd.right - (cr.position if cr else 0),
                "left_length": sd.left_length,
                "right_length": sd.right_length,
                "left_seq": None,
                "right_seq": None,
                "identity": F32(sd.identity),
                "reversed": bool(sd.reversed),
                "complemented": bool(sd.complemented),
            })
        fams.append(out)
    return {
        "strand": {
            "name": strand.file_names,
            "length": sum(c.length for c in strand.map),
            "map": [{"name": c.name, "position": c.position, "length": c.length} for c in strand.map],
        },
        # reverse/complement/threads_count/compute_score are skip_serializing (src/structs.rs:44-57)
        "settings": {
            "probe_size": settings.probe_size,
            "max_gap_size": settings.max_gap_size,
            "min_duplication_length": settings.min_duplication_length,
            "max_cardinality": settings.max_cardinality,
            "trim": list(settings.trim) if settings.trim else None,
            "skip_masked": bool(settings.skip_masked),
        },
        "families": fams,
    }


def f32_repr(v) -> str:
    """An f32 the way serde_json prints it (ryu): the shortest decimal that reads back as the same
    f32, `97.3` not `97.30000305175781`; always with a fraction or an exponent (`0.0`, `100.0`,
    `1e-7`).  NaN and infinities serialise as `null` (serde_json)."""
    x = np.float32(v)
    if not np.isfinite(x):
        return "null"
    if x == 0:
        return "-0.0" if np.signbit(x) else "0.0"
    digits, exp = np.format_float_scientific(x, unique=True, trim="-", exp_digits=1).split("e")
    e10 = int(exp)
    mant = digits.replace(".", "").lstrip("-")
    kk = e10 + 1  # position of the decimal point relative to the first digit
    if -6 < kk <= 13:  # ryu's f32 pretty printer: plain decimal notation for 1e-6 <= |x| < 1e13
        return np.format_float_positional(x, unique=True, trim="0")
    sign = "-" if x < 0 else ""
    body = mant[0] + ("." + mant[1:] if len(mant) > 1 else "")
    return f"{sign}{body}e{kk - 1}"


class F32(float):
    """A float that `to_json` prints as an f32 (ProtoSD/SD.identity is `f32`, src/structs.rs:424,485)."""


def _dump(obj, ind: int, out: list):
    pad, pad_in = "  " * ind, "  " * (ind + 1)
    if isinstance(obj, dict):
        if not obj:
            out.append("{}")
            return
        out.append("{\n")
        for j, (key, val) in enumerate(obj.items()):
            out.append(pad_in + json.dumps(str(key), ensure_ascii=False) + ": ")
            _dump(val, ind + 1, out)
            out.append(",\n" if j + 1 < len(obj) else "\n")
        out.append(pad + "}")
    elif isinstance(obj, (list, tuple)):
        if not obj:
            out.append("[]")
            return
        out.append("[\n")
        for j, val in enumerate(obj):
            out.append(pad_in)
            _dump(val, ind + 1, out)
            out.append(",\n" if j + 1 < len(obj) else "\n")
        out.append(pad + "]")
    elif obj is None:
        out.append("null")
    elif isinstance(obj, bool):
        out.append("true" if obj else "false")
    elif isinstance(obj, (F32, np.float32)):
        out.append(f32_repr(obj))
    elif isinstance(obj, (int, np.integer)):
        out.append(str(int(obj)))
    elif isinstance(obj, float):
        out.append(json.dumps(obj))
    else:
        out.append(json.dumps(str(obj), ensure_ascii=False))


def to_json(result: dict) -> str:
    """`serde_json::to_string_pretty` text of a RunResult (src/exporters.rs:12-25): 2-space indent,
    fields in struct order, `null`, UTF-8 unescaped, `identity` as the shortest f32 decimal."""
    out: list = []
    _dump(result, 0, out)
    return "".join(out)


def post_process_arrays(index, offs: np.ndarray, sds: np.ndarray, threads: int = 0):
    """The step chain behind the search (FilterNs, ReOrder, ReduceOverlap, Sort) on raw family arrays, at the scale of
    a whole-genome run: asgart_post_process (N counts on the GPU, the quadratic reduction on host threads) instead of
    the per-object loops above, which remain the readable statement of the same steps (tests compare the two)."""
    return index.post_process(offs, sds, threads)


def to_json_arrays(offs: np.ndarray, sds: np.ndarray, strand, settings: RunSettings, identity: Optional[np.ndarray] = None) -> str:
    """`to_json(run_result(...))` for family arrays: the same bytes, one format operation per duplication (the
    chromosome of a position by bisection of the strand map, whose records follow one another: the reference's
    first-match scan, src/structs.rs:85-90, finds the same record)."""
    offs = np.asarray(offs, dtype=np.int64)
    sds = np.asarray(sds, dtype=np.uint64).reshape(-1, 4)
    starts = np.array([c.position for c in strand.map], dtype=np.uint64)
    ends = starts + np.array([c.length for c in strand.map], dtype=np.uint64)
    names = [json.dumps(c.name, ensure_ascii=False) for c in strand.map] + ['"unknown"']

    def locate(pos):
        i = np.searchsorted(starts, pos, side="right").astype(np.int64) - 1
        ok = (i >= 0) & (pos < ends[np.maximum(i, 0)])
        i = np.where(ok, i, len(strand.map))
        return i, np.where(ok, pos - starts[np.minimum(np.maximum(i, 0), len(starts) - 1)], pos)

    if len(strand.map):
        il, pl = locate(sds[:, 0])
        ir, pr_ = locate(sds[:, 1])
    else:
        il = ir = np.zeros(len(sds), dtype=np.int64)
        pl, pr_ = sds[:, 0], sds[:, 1]
    ident = [f32_repr(v) for v in identity] if identity is not None else None
    rev = "true" if settings.reverse else "false"
    comp = "true" if settings.complement else "false"
    sd_txt = []
    L, R, LL, RL = (sds[:, j].tolist() for j in range(4))
    il, ir, pl, pr_ = il.tolist(), ir.tolist(), pl.tolist(), pr_.tolist()
    for j in range(len(sds)):
        sd_txt.append(
            '      {\n'
            f'        "chr_left": {names[il[j]]},\n'
            f'        "chr_right": {names[ir[j]]},\n'
            f'        "global_left_position": {L[j]},\n'
            f'        "global_right_position": {R[j]},\n'
            f'        "chr_left_position": {pl[j]},\n'
            f'        "chr_right_position": {pr_[j]},\n'
            f'        "left_length": {LL[j]},\n'
            f'        "right_length": {RL[j]},\n'
            '        "left_seq": null,\n'
            '        "right_seq": null,\n'
            f'        "identity": {ident[j] if ident else "0.0"},\n'
            f'        "reversed": {rev},\n'
            f'        "complemented": {comp}\n'
            '      }')
    fams = []
    for f in range(len(offs) - 1):
        a, b = int(offs[f]), int(offs[f + 1])
        fams.append("    [\n" + ",\n".join(sd_txt[a:b]) + "\n    ]" if b > a else "    []")
    head = to_json({"strand": run_result([], strand, settings)["strand"], "settings": run_result([], strand, settings)["settings"]})
    # (head ends with "\n}"; the families array is appended as the third field)
    body = "[\n" + ",\n".join(fams) + "\n  ]" if fams else "[]"
    return head[:-2] + ',\n  "families": ' + body + "\n}"


def out_filename(files: Sequence[str], settings: RunSettings, prefix: str = "") -> str:
    """Default output name, src/bin/asgart.rs:642-714: {prefix}{stems joined '-'}[_][R][C][_a-b].json"""
    import os

    radix = "-".join(os.path.splitext(os.path.basename(f))[0] for f in files)
    rc = ("_" if settings.reverse or settings.complement else "") + \
         ("R" if settings.reverse else "") + ("C" if settings.complement else "")
    trim = f"_{settings.trim[0]}-{settings.trim[1]}" if settings.trim else ""
    return f"{prefix}{radix}{rc}{trim}.json"


def search_duplications(strands_files: Sequence[str], settings: RunSettings, device: int = 0,
                        compute_score: bool = False) -> dict:
    """`search_duplications()` of src/bin/asgart.rs:731-822: prepare_data, the step chain with the
    HIP search step first, then the RunResult.  Raises AsgartError without a GPU (no CPU fallback)."""
    from . import Index, SearchDuplications, Strand
    from .prep import prepare_records, read_records, validate_trim

    records = [rec for f in strands_files for rec in read_records(f)]
    pr = prepare_records(records, settings.skip_masked)
    strand = Strand(", ".join(strands_files), pr.data, pr.map)  # file_names, :437
    trim = validate_trim(settings.trim, len(pr.data))           # prepare_data's checks, :432-463
    with Index(strand.data, None, device, trim=trim) as index:
        families = SearchDuplications(pr.chunks, trim, settings, index=index).run([], strand)
        families = post_process(families, strand, index, compute_score)
    return run_result(families, strand, settings)
