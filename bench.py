#!/usr/bin/env python3
"""bench.py -- Mbp/s of the probe -> SA search -> extend hot path on MI355X.

A "step" is one pass of the hot path over the whole synthetic genome: the direct run
plus the reverse-complement run (`asgart` and `asgart -RC`, BASELINE.json "direct+RC"),
i.e. the equivalent of reference src/bin/asgart.rs:201-253 executed twice over the same
index.  The index (text, suffix array, search keys, presence filters) is resident in HBM
before the timed region; results (families of ProtoSD) are back on the host when it ends.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg1|cfg2|cfg3|cfg4|cfg5|tiny]

cfg1..cfg5 are BASELINE.json configs[0..4] (cfg1: direct pass only; cfg3: --skip-masked).
Prints ONE JSON line on rank 0 (driver contract), with two extra objects:

"roofline" -- the dominant HBM-bound kernels, probe_count_kernel + big_count_kernel + rank_count_kernel:
    achieved  = kernel_algorithmic_bytes / avg_launch_ms: the bytes THIS kernel's design loads and
                stores per launch (window staging, filter word, prefix-table entries, keys the
                bisection reads, suffix-array entries read, outputs), counted exactly by the
                library's accounting pass over the same probes, over the kernels' duration
                measured live with HIP events on the library's stream in the timed region;
    frac      = achieved / peak (8 TB/s);
    traffic   = HBM-side bytes per launch from the rocprofv3 PMC passes kept in profiles/
                (FETCH_SIZE + WRITE_SIZE, tools/profile_round.sh), traffic_ms the kernels'
                duration in those passes, traffic_frac = traffic / traffic_ms / peak,
                waste = traffic / kernel_algorithmic_bytes;
    reference_algorithm_bytes = SURVEY.md section 8d yardstick (the REFERENCE's algorithm at its
                widths) -- context only, never used for frac.
"cpu_baseline" -- the CPU oracle (OpenMP over chunks like the reference's rayon par_iter) timed on
    this host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# The extension tiers of a call run on six HIP streams (twelve with both passes of a step in flight).  ROCm maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of streams that share a queue run one after
# the other.  Must be set before the process's first HIP call; the library sets the same default when it is loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import asgart_amd  # noqa: E402
from asgart_amd import multi, prep, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
GATHER_CEILING_GBS = 3120.0  # measured: random 4-/8-byte gathers, 64 B counted each (profiles/r02_ubench_gather.txt)
DIRECT_RC = ((False, False), (True, True))  # direct, then -RC
WORKLOADS = {
    # name: (synth config id, scale, skip_masked, modes, description)
    "tiny": (2, 0.05, False, DIRECT_RC, "S. cerevisiae-shaped synthetic x0.05 (0.6 Mb), direct+RC, k=20 g=100"),
    "cfg1": (1, 1.0, False, ((False, False),), "E. coli K-12 MG1655-shaped synthetic (4.6 Mb), direct, k=20 g=100"),
    "cfg2": (2, 1.0, False, DIRECT_RC, "S. cerevisiae S288C-shaped synthetic (12.2 Mb, 17 records), direct+RC, k=20 g=100"),
    "cfg3": (3, 1.0, True, DIRECT_RC, "human chr1-shaped synthetic (249 Mb), direct+RC, --skip-masked, k=20 g=100"),
    "cfg4": (4, 1.0, False, DIRECT_RC, "GRCh38-shaped synthetic (3.1 Gb, 25 records), direct+RC, k=20 g=100"),
    "cfg5": (5, 1.0, False, DIRECT_RC, "GRCh38-shaped + 1.2 %-diverged second genome (two files, 6.1 Gb), direct+RC, k=20 g=100"),
    # not a BASELINE.json config: a realism check (young interspersed repeats: two probes in five pass the filter)
    "cfg3r": (6, 1.0, False, DIRECT_RC, "human chr1-sized, repeat-rich synthetic (249 Mb; 27 % SINE-like + 15 % LINE-like families at 1-5 % divergence), direct+RC, k=20 g=100"),
}


def fasta_inputs(args):
    """--fasta a.fa [b.fa ...], or every FASTA file of $ASGART_DATA_DIR (real assemblies are
    used instead of the synthetic stand-ins when present).  Several files are concatenated record by record, as the
    reference does (src/bin/asgart.rs:375-395)."""
    files = list(args.fasta or [])
    d = os.environ.get("ASGART_DATA_DIR")
    if not files and d and os.path.isdir(d):
        files = sorted(os.path.join(d, f) for f in os.listdir(d)
                       if f.lower().endswith((".fa", ".fasta", ".fna", ".fa.gz", ".fasta.gz", ".fna.gz")))
    return files


def read_fasta_files(files):
    import gzip
    import shutil
    import tempfile
    recs = []
    for f in files:
        if f.endswith(".gz"):
            with gzip.open(f, "rb") as src, tempfile.NamedTemporaryFile(suffix=".fa") as tmp:
                shutil.copyfileobj(src, tmp)
                tmp.flush()
                recs.extend(prep.read_records(tmp.name))
        else:
            recs.extend(prep.read_records(f))
    return recs


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def reference_algorithm_bytes(st: dict, k: int, W: int = 8) -> int:
    """B_p = k + 16 + 2*ceil(log2(b_p+1))*(W+k) + W*h_p summed over the
    searched probes (reference algorithm at reference widths, W = 8-byte SA entries)."""
    return (st["probes_searched"] * (k + 16) + 2 * st["bisect_steps"] * (W + k)
            + W * st["raw_hits"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("ASGART_BENCH_WORKLOAD", "cfg4"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fasta", nargs="+", default=None, help="real FASTA input(s) instead of the synthetic workload")
    ap.add_argument("--skip-masked", action="store_true", help="with --fasta: lower-case bases count as N (-S)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod

        # debugging knobs for 1-GPU boxes: all ranks on device 0 + gloo (RCCL refuses two ranks
        # on one device); the driver's multi-GPU runs use neither
        if os.environ.get("ASGART_BENCH_ONE_DEVICE"):
            local_rank = 0
        backend = os.environ.get("ASGART_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
        comm_device = f"cuda:{local_rank}" if backend == "nccl" else "cpu"

    cfg, scale, skip_masked, modes, desc = WORKLOADS[args.workload]
    k, gap = 20, 100
    t0 = time.time()
    files = fasta_inputs(args)
    data_kind = "synthetic"
    if files:
        recs = read_fasta_files(files)
        skip_masked, modes = bool(args.skip_masked), DIRECT_RC
        desc = (f"FASTA {', '.join(os.path.basename(f) for f in files)} ({len(recs)} records), direct+RC"
                f"{', --skip-masked' if skip_masked else ''}, k=20 g=100")
        data_kind = "real"
    else:
        recs = synth.config_genome(cfg, scale)
    pr = prep.prepare_records(recs, skip_masked=skip_masked)
    del recs
    total_bp = sum(l for _, l in pr.chunks)
    t_gen = time.time() - t0
    if rank == 0:
        log(f"[bench] {desc}: {len(pr.data)} bytes, {len(pr.chunks)} chunks, gen {t_gen:.1f}s")

    # ---- index build (outside the timed region, reported separately) -------------
    t0 = time.time()
    idx = asgart_amd.Index(pr.data, None, device=local_rank)   # suffix array built on the GPU
    t_sa = time.time() - t0
    t0 = time.time()
    idx.prepare(k)
    t_index = time.time() - t0
    if rank == 0:
        log(f"[bench] upload + GPU suffix array {t_sa:.2f}s, search keys/tables {t_index:.2f}s")

    settings = [asgart_amd.RunSettings.from_cli(k=k, gap=gap, reverse=r, complement=c, skip_masked=skip_masked)
                for r, c in modes]

    def sync():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    def run_pass(st):
        if world > 1:
            return idx.search_duplications_raw(pr.chunks, st, shard=rank, n_shards=world, with_keys=True)
        return idx.search_duplications_raw(pr.chunks, st)

    def one_step():
        out = []
        for st in settings:
            r_ = run_pass(st)
            out.append(multi.gather_families(r_[0], r_[1], dist, device=comm_device, keys=r_[2]) if world > 1 else r_)
        return out

    t0 = time.time()
    for _ in range(args.warmup):   # the first call of an orientation also builds its presence filter
        one_step()
    t_warm = time.time() - t0
    # per-pass device timings + work counters + the accounting pass (one extra untimed call per
    # mode, the passes one after the other: "alone on the chip" kernel durations)
    pass_stats = []
    for st in settings:
        run_pass(st)
        pass_stats.append(idx.stats(1).as_dict())   # rank-local work counters

    # Timed region.  The library is re-entrant (one internal context per call), so the direct and
    # the RC pass of a step can be issued from two host threads and overlap on the GPU: while one
    # pass runs its serial extension chains the other one's search kernels use the idle CUs.
    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(max_workers=len(settings))

    def run_pipelined(order):
        """Both passes of a step, pipelined: the second call is issued when the first one reports (through
        the progress array of the C ABI, polled like the reference's progress bar polls its counters) that
        its probes are searched; its own search phases then run beside the extension of the first, whose
        tail is a few serial segments.  Returns the results in `settings` order."""
        first, second = order
        prog = np.zeros(len(pr.chunks), dtype=np.uint64)
        fut = pool.submit(idx.search_duplications_raw, pr.chunks, settings[first], rank if world > 1 else 0,
                          world if world > 1 else 1, prog, world > 1)
        while not fut.done() and not prog.any():
            time.sleep(0.0005)
        res = {second: run_pass(settings[second]), first: fut.result()}
        return [res[j] for j in range(len(settings))]

    # Four ways to issue the two passes of a step; which is fastest depends on how much of a pass is its
    # serial extension tail (pipelining hides it) and how much is chip-wide work the passes would only
    # steal from each other.  One untimed step of each decides (ASGART_BENCH_MODE forces one).
    MODES_OF_ISSUE = ("library", "back_to_back", "overlapped", "pipelined_0_first", "pipelined_1_first")

    def issue(mode):
        """-> (results in `settings` order, the library's per-call stats)"""
        if mode == "library" and len(settings) > 1:
            # ONE call for the passes of a step (asgart_search_duplications_passes): the library issues pass j+1 when
            # pass j's probes are searched, longest extension first -- no host threads, no polling here
            results = idx.search_duplications_passes(pr.chunks, settings, rank if world > 1 else 0,
                                                     world if world > 1 else 1, with_keys=world > 1)
            return results, [idx.stats((ci + 1) << 8) for ci in range(len(settings))]
        if mode == "back_to_back" or len(settings) == 1:
            results, stats = [], []
            for st in settings:
                results.append(run_pass(st))
                stats.append(idx.stats(0))          # the call just finished
            return results, stats
        if mode == "overlapped":
            results = list(pool.map(run_pass, settings))
        else:
            results = run_pipelined((0, 1) if mode == "pipelined_0_first" else (1, 0))
        return results, [idx.stats((ci + 1) << 8) for ci in range(len(settings))]  # one context per call

    mode_probe_ms = None
    mode = os.environ.get("ASGART_BENCH_MODE", "")
    if os.environ.get("ASGART_BENCH_OVERLAP") in ("0", "1"):   # (older switch)
        mode = "back_to_back" if os.environ["ASGART_BENCH_OVERLAP"] == "0" else "overlapped"
    if len(settings) == 1:
        mode = "back_to_back"
    elif mode not in MODES_OF_ISSUE:
        # default: the library's own pipelining.  One untimed step of each way of issuing is recorded for
        # information (ASGART_BENCH_MODE=auto picks the fastest of them instead).
        mode_probe_ms = {}
        for m in MODES_OF_ISSUE if (mode == "auto" or os.environ.get("ASGART_BENCH_PROBE_MODES")) else ("back_to_back", "library"):
            sync()
            t_a = time.perf_counter()
            issue(m)
            dt = time.perf_counter() - t_a
            if dist is not None:   # a step lasts as long as its slowest rank; every rank must pick the same mode
                import torch

                v = torch.tensor([dt], device=comm_device)
                dist.all_reduce(v, op=dist.ReduceOp.MAX)
                dt = float(v.item())
            mode_probe_ms[m] = round(dt * 1e3, 2)
        mode = min(mode_probe_ms, key=mode_probe_ms.get) if mode == "auto" else "library"
    elif mode != "back_to_back":
        issue(mode)   # untimed: a forced concurrent mode allocates its second call context here, not in the timed region
    sequential = mode == "back_to_back"

    sync()
    t0 = time.perf_counter()
    search_ms = 0.0
    probe_count_ms = 0.0
    phase_ms = {"search": 0.0, "scan": 0.0, "fill": 0.0, "extend": 0.0, "extend_tier2": 0.0}
    for _ in range(args.steps):
        results, per_call = issue(mode)
        if world > 1:
            # the only exchange of the path: duplicon lists -> rank 0 over RCCL
            results = [multi.gather_families(r_[0], r_[1], dist, device=comm_device, keys=r_[2]) for r_ in results]
        for s in per_call:
            search_ms += s.ms_search
            probe_count_ms += s.ms_probe_count
            for ph in phase_ms:
                phase_ms[ph] += getattr(s, "ms_" + ph)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    passes = len(modes)
    value = total_bp * passes * args.steps / elapsed / 1e6
    n_launch = args.steps * passes
    alg_bytes = sum(s["search_bytes"] for s in pass_stats) / passes          # per launch, this design
    ref_bytes = sum(reference_algorithm_bytes(s, k) for s in pass_stats) / passes
    avg_launch_ms = search_ms / n_launch if n_launch else 0.0
    achieved = alg_bytes / (avg_launch_ms / 1e3) / 1e9 if avg_launch_ms > 0 else 0.0
    alone_ms = sum(s["ms_search"] for s in pass_stats) / passes
    prof = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            prof = json.load(open(tpath)).get(args.workload) or {}
            if not isinstance(prof, dict):
                prof = {}
        except Exception:
            prof = {}
    traffic = prof.get("traffic_bytes_per_launch")
    traffic_ms = prof.get("kernel_ms_per_launch")
    roofline = {
        "bound": "hbm", "kernel": "probe_count_kernel + big_count_kernel + rank_count_kernel (one launch = one pass)",
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "kernel_algorithmic_bytes": int(alg_bytes),
        "avg_launch_ms": round(avg_launch_ms, 5),
        "launch_timing": "HIP events on the library's stream, timed region" +
                         ("" if sequential else " (the other pass's kernels share the chip)"),
        # first kernel of the pair alone (rocprofv3 lists the two kernels separately)
        "probe_count_kernel_ms": round(probe_count_ms / n_launch, 5) if n_launch else 0.0,
        "alone_launch_ms": round(alone_ms, 5),
        "achieved_alone": round(alg_bytes / (alone_ms / 1e3) / 1e9, 2) if alone_ms > 0 else 0.0,
        "traffic": traffic,
        "traffic_ms": traffic_ms,
        "traffic_frac": (round(traffic / (traffic_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
                         if traffic and traffic_ms else None),
        "traffic_x2_applied": False,
        "traffic_source": prof.get("source"),
        "waste": round(traffic / alg_bytes, 3) if traffic and alg_bytes else None,
        # The kernels are random-gather bound: an 8- or 4-byte gather moves (and FETCH_SIZE counts) one 64-byte
        # sector, so `traffic` exceeds the algorithmic bytes by design (`waste` is sector granularity, not re-reads),
        # and the ceiling of this access pattern is the measured gather rate, not the 8 TB/s stream peak:
        # tools/ubench_gather.hip, profiles/r02_ubench_gather.txt: 48.8 G gathers/s x 64 B.
        "gather_ceiling": GATHER_CEILING_GBS,
        "traffic_frac_of_gather_ceiling": (round(traffic / (traffic_ms / 1e3) / 1e9 / GATHER_CEILING_GBS, 4)
                                           if traffic and traffic_ms else None),
        "reference_algorithm_bytes": int(ref_bytes),
        "filter_rejected_frac": round(sum(s["probes_filter_rejected"] for s in pass_stats) /
                                      max(1, sum(s["probes_searched"] for s in pass_stats)), 4),
    }

    out = {
        "metric": "Mbp/s probe+extend (direct+RC, k=20 g=100)",
        "value": round(value, 3),
        "unit": "Mbp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64 keys / u32 SA" if len(pr.data) < 0xFFFFFF00 else "u64 keys / u64 SA",
        "data": data_kind,
        "config": {"workload": desc, "bp_per_pass": total_bp, "passes": passes,
                   "skip_masked": skip_masked,
                   "text_bytes": int(len(pr.data)), "chunks": len(pr.chunks),
                   "parallelism": f"probe-shard x{world}" if world > 1 else "1 GPU",
                   "passes_issued": mode, "mode_probe_ms": mode_probe_ms},
        "roofline": roofline,
        "phases_ms_per_step": {ph: round(v / args.steps, 4) for ph, v in phase_ms.items()},
        "index_build_s": {"upload_and_suffix_array": round(t_sa, 2), "keys_and_tables": round(t_index, 3),
                          "first_calls_incl_presence_filters": round(t_warm, 3),
                          "sa_builder": "GPU prefix doubling (asgart_sa_build64 path)"},
        "work_per_step": {key: sum(s[key] for s in pass_stats) for key in
                          ("probes_total", "probes_searched", "probes_card_skipped", "probes_filter_rejected",
                           "raw_hits", "filtered_hits", "segments", "overflow_segments", "heavy_segments",
                           "families", "proto_sds")},
    }

    if rank == 0 and not args.no_cpu_baseline:
        # CPU reference: the oracle with the reference's parallel structure (OpenMP over
        # chunks == rayon par_iter, src/bin/asgart.rs:201-205), same index, same chunks.
        import oracle  # the CPU checker, used here ONLY as the timed CPU baseline

        cores = min(os.cpu_count() or 1, len(pr.chunks))  # threads that can be busy: one per chunk
        n_text = len(pr.data)
        sa = np.empty(n_text, dtype=np.int64)   # (read back in slabs: sa_read returns a fresh array per call)
        slab = 1 << 28
        for o in range(0, n_text, slab):
            sa[o:min(n_text, o + slab)] = idx.sa_read(o, min(n_text, o + slab))
        oidx = oracle.Index.build(pr.data, sa)
        # bounded sample: the first `per_chunk` bases of EVERY chunk, so the CPU leg keeps the
        # reference's chunk-level parallelism (one thread per chunk) without its skew
        budget_bp = float(os.environ.get("ASGART_CPU_SAMPLE_BP", 200e6))
        per_chunk = max(100_000, int(budget_bp / max(1, len(pr.chunks))))
        sample_chunks = [(s0, min(l0, per_chunk)) for s0, l0 in pr.chunks]
        sample_bp = sum(l0 for _, l0 in sample_chunks)
        t0 = time.perf_counter()
        for r, c in modes:
            oidx.run_raw(sample_chunks, oracle.make_settings(k=k, gap=gap, reverse=r, complement=c),
                         threads=cores)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(sample_bp * passes / t_cpu / 1e6, 3), "unit": "Mbp/s", "cores": cores,
            "kind": "port",
            "sample": (f"first <= {per_chunk} bp of each of the {len(pr.chunks)} chunks "
                       f"({sample_bp} bp) x {passes} passes against the full index, "
                       f"OpenMP over chunks, {t_cpu:.2f} s"),
        }
    if rank == 0:
        print(json.dumps(out), flush=True)
    idx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
