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
    "cfg4r": (7, 1.0, False, DIRECT_RC, "GRCh38-sized, repeat-rich synthetic (3.1 Gb, 25 records; 27 % SINE-like + 15 % LINE-like families at 1-5 % divergence; higher-order satellite arrays: 6-12 monomers of 171 bp at 18-32 % from one another, unit copies at 1-2 %), direct+RC, k=20 g=100"),
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


def end_to_end(recs, files, skip_masked, modes, k, gap, device):
    """The host chain around the path, once, per stage in seconds (reference src/bin/asgart.rs:731-822: prepare_data,
    the Step chain -- SearchDuplications, FilterNs, ReOrder, ReduceOverlap, Sort -- ProtoSD -> SD, JSON export), for
    every pass of the workload.  Untimed in `value`; SURVEY.md section 8d lists these stages as reported separately."""
    from asgart_amd import Strand, postprocess

    out = {}
    t0 = time.perf_counter()
    if files:
        recs = read_fasta_files(files)
        out["read_fasta"] = round(time.perf_counter() - t0, 3)
        t0 = time.perf_counter()
    # prepare_data behind the C ABI (asgart_prepare_data): normalisation + chunking on the GPU, the index built from the
    # same device buffer; the prepared strand stays on the device (nothing downstream of it needs the bytes on the host)
    pr, idx_e2e = prep.prepare_records_gpu(recs, skip_masked=skip_masked, device=device, want_text=False)
    out["prepare_data_and_suffix_array"] = round(time.perf_counter() - t0, 3)
    strand = Strand(", ".join(files) if files else "synthetic", None, pr.map)
    t0 = time.perf_counter()
    with idx_e2e as idx:
        idx.prepare(k)
        out["search_keys_and_tables"] = round(time.perf_counter() - t0, 3)
        sts = [asgart_amd.RunSettings.from_cli(k=k, gap=gap, reverse=r, complement=c, skip_masked=skip_masked)
               for r, c in modes]
        t0 = time.perf_counter()
        raw = idx.search_duplications_passes(pr.chunks, sts) if len(sts) > 1 else \
            [idx.search_duplications_raw(pr.chunks, sts[0])]
        out["search_all_passes"] = round(time.perf_counter() - t0, 3)
        post, export, n_sd = 0.0, 0.0, 0
        for st, (offs, sds) in zip(sts, raw):
            t0 = time.perf_counter()
            fo, fs = postprocess.post_process_arrays(idx, offs, sds)
            post += time.perf_counter() - t0
            t0 = time.perf_counter()
            text = postprocess.to_json_arrays(fo, fs, strand, st)
            export += time.perf_counter() - t0
            n_sd += len(fs)
            out.setdefault("json_bytes", 0)
            out["json_bytes"] += len(text)
        out["post_processing"] = round(post, 3)
        out["run_result_json"] = round(export, 3)
        out["sds_after_post_processing"] = n_sd
    return out


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N ranks (fresh interpreters, one per GPU, the same
    command line) and wait for them.  This parent never touches a GPU -- no HIP call, no torch.cuda -- so the children
    are ordinary child processes of a process without device state (nothing is re-executed).  Rank 0 prints the JSON
    line on the stdout the children inherit.  Exit code: the first non-zero one of the ranks."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ASGART_BENCH_LAUNCHED="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("ASGART_BENCH_ONE_DEVICE") and not os.environ.get("ASGART_BENCH_KEEP_HW_QUEUES"):
            # all ranks on ONE device share its hardware queues: with 8 per process the tiers' streams of a rank are multiplexed
            # onto what is left and run one after the other (measured: per-rank extension 329 / 155 ms instead of 49 / 65); 4
            # per rank keeps them side by side, two streams to a queue (set whatever the environment says: the pool's boxes
            # come with GPU_MAX_HW_QUEUES=8)
            env["GPU_MAX_HW_QUEUES"] = str(max(2, 8 // n_ranks))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p_ in list(live):
            code = p_.poll()
            if code is None:
                continue
            live.remove(p_)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q_ in live:   # a rank died: the others would wait in a collective forever (exact PIDs, ours)
                    q_.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    # (default 5: a passes call that finds one segment to be its extension is followed by calls timed as one job and
    # pipelined in turn -- asgart_amd/csrc/pipeline.hip, fuse_passes --: after five calls the library has settled)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("ASGART_BENCH_WORKLOAD", "cfg4"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fasta", nargs="+", default=None, help="real FASTA input(s) instead of the synthetic workload")
    ap.add_argument("--skip-masked", action="store_true", help="with --fasta: lower-case bases count as N (-S)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time the host chain around the path once (records -> prepare_data -> index -> search -> "
                         "post-processing -> RunResult JSON; reference src/bin/asgart.rs:731-822), untimed in `value`")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")
    dist = None
    comm_device = "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist_mod

        # debugging knob for 1-GPU boxes: all ranks on device 0 over gloo (RCCL refuses two ranks on one device);
        # the driver's multi-GPU runs do not use it
        one_device = bool(os.environ.get("ASGART_BENCH_ONE_DEVICE"))
        if one_device:
            local_rank = 0
        backend = os.environ.get("ASGART_BENCH_BACKEND", "gloo" if one_device else "nccl")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
        comm_device = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        assert dist.get_world_size() == world, "the process group is not the world the launcher announced"

    cfg, scale, skip_masked, modes, desc = WORKLOADS[args.workload]
    k, gap = 20, 100
    files = fasta_inputs(args)
    data_kind = "real" if files else "synthetic"
    if files:
        skip_masked, modes = bool(args.skip_masked), DIRECT_RC

    # ---- input: rank 0 reads / generates it; the other ranks only need the chunk table (their index arrives by
    # broadcast, device to device) --------------------------------------------------------------------------------
    t0 = time.time()
    pr = None
    recs = None
    if rank == 0:
        if files:
            recs = read_fasta_files(files)
            desc = (f"FASTA {', '.join(os.path.basename(f) for f in files)} ({len(recs)} records), direct+RC"
                    f"{', --skip-masked' if skip_masked else ''}, k=20 g=100")
        else:
            recs = synth.config_genome(cfg, scale)
        pr = prep.prepare_records(recs, skip_masked=skip_masked)
        if not args.end_to_end:
            recs = None
        log(f"[bench] {desc}: {len(pr.data)} bytes, {len(pr.chunks)} chunks, gen {time.time() - t0:.1f}s")
    if world > 1:
        box = [(desc, [tuple(map(int, c)) for c in pr.chunks], int(len(pr.data)))] if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        desc, chunks, n_text = box[0]
    else:
        chunks, n_text = pr.chunks, int(len(pr.data))
    total_bp = sum(l for _, l in chunks)
    settings = [asgart_amd.RunSettings.from_cli(k=k, gap=gap, reverse=r, complement=c, skip_masked=skip_masked)
                for r, c in modes]
    shard_args = (rank, world) if world > 1 else (0, 1)

    def sync():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    def gather(r_):
        return multi.gather_families(r_[0], r_[1], dist, device=comm_device, keys=r_[2]) if world > 1 else r_

    def passes_call():
        if len(settings) > 1:
            return idx.search_duplications_passes(chunks, settings, *shard_args, with_keys=world > 1)
        return [idx.search_duplications_raw(chunks, settings[0], *shard_args, with_keys=world > 1)]

    # ---- the cold run: host text -> families of every pass on the host, in this fresh process ----------------------
    # (upload, GPU suffix array [+ its broadcast to the other ranks], search keys and tables, presence filters, the
    # first call with its allocations).  Reported as cold_s, never part of `value`.
    sync()
    t_c0 = time.perf_counter()
    idx = asgart_amd.Index(pr.data, None, device=local_rank) if rank == 0 else None   # suffix array built on the GPU
    t_sa = time.perf_counter() - t_c0
    t_bc = 0.0
    if world > 1:
        t1 = time.perf_counter()
        idx = multi.replicate_index(idx, dist, local_rank)
        t_bc = time.perf_counter() - t1
    t1 = time.perf_counter()
    idx.prepare(k)
    t_index = time.perf_counter() - t1
    t1 = time.perf_counter()
    first = [gather(r_) for r_ in passes_call()]
    sync()
    t_first = time.perf_counter() - t1
    cold_s = time.perf_counter() - t_c0
    if rank == 0:
        log(f"[bench] cold run {cold_s:.2f}s: upload + GPU suffix array {t_sa:.2f}s, "
            f"{'broadcast %.2fs, ' % t_bc if world > 1 else ''}search keys/tables {t_index:.2f}s, "
            f"first passes call (filters, allocations) {t_first:.2f}s")
    del first

    def run_pass(st):
        return idx.search_duplications_raw(chunks, st, *shard_args, with_keys=world > 1)

    def one_step():
        return [gather(r_) for r_ in passes_call()]

    t0 = time.time()
    # (An unsharded passes call that finds one segment to be its extension makes the library TIME the calls that follow as
    # one job and as pipelined single-pass calls in turn and keep the faster way -- asgart_hip.h, fuse_passes; calls that
    # still refuse cuts do not count as samples.  The timed region must lie behind that: after the warm-up steps asked for,
    # untimed steps go on -- twelve at most -- until four calls in a row have run the same way.)
    ways = []
    settling_steps = 0
    for _ in range(args.warmup):
        one_step()
        ways.append(int(idx.stats(0).passes))
    while len(settings) > 1 and world == 1 and settling_steps < 12 and (len(ways) < 4 or len(set(ways[-4:])) > 1):
        one_step()
        ways.append(int(idx.stats(0).passes))
        settling_steps += 1
    t_warm = time.time() - t0
    # per-pass device timings + work counters + the accounting pass (one extra untimed call per
    # mode, the passes one after the other: "alone on the chip" kernel durations)
    # (the passes call runs passes that differ in orientation only as ONE job -- one launch of every kernel over all
    # their probes, asgart_stats.passes says how many; sharded calls and ASGART_FUSE_PASSES=0 pipeline them as calls)
    pass_stats = []
    fused = False
    if len(settings) > 1 and os.environ.get("ASGART_BENCH_MODE", "") in ("", "library", "auto"):
        passes_call()
        st_f = idx.stats(1)
        if st_f.passes == len(settings):
            fused = True
            pass_stats.append(st_f.as_dict())
    if not fused:
        for st in settings:
            run_pass(st)
            pass_stats.append(idx.stats(1).as_dict())   # rank-local work counters
    passes_per_launch = len(settings) if fused else 1
    # what the scaling model is made of: ONE GPU's front (probe search, scans, hit rows), extension and longest tier
    # per pass.  With several ranks, rank 0 runs the passes unsharded once while the others wait.
    if world > 1:
        whole = []
        if rank == 0:
            if len(settings) > 1:
                idx.search_duplications_passes(chunks, settings)   # unsharded: one fused job
                if idx.stats(0).passes == len(settings):
                    whole.append(idx.stats(0).as_dict())
            if not whole:
                for st in settings:
                    idx.search_duplications_raw(chunks, st)
                    whole.append(idx.stats(0).as_dict())
        sync()
    else:
        whole = pass_stats

    # Timed region.  The library is re-entrant (one internal context per call), so the direct and
    # the RC pass of a step can be issued from two host threads and overlap on the GPU: while one
    # pass runs its serial extension chains the other one's search kernels use the idle CUs.
    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(max_workers=len(settings))

    def run_pipelined(order):
        """Both passes of a step, pipelined by the HOST (the older way; the library's passes call does the same
        inside): the second call is issued when the first one reports, through the progress array of the C ABI, that
        its probes are searched.  Returns the results in `settings` order."""
        first_, second = order
        prog = np.zeros(len(chunks), dtype=np.uint64)
        fut = pool.submit(idx.search_duplications_raw, chunks, settings[first_], shard_args[0], shard_args[1], prog,
                          world > 1)
        while not fut.done() and not prog.any():
            time.sleep(0.0005)
        res = {second: run_pass(settings[second]), first_: fut.result()}
        return [res[j] for j in range(len(settings))]

    # Ways to issue the two passes of a step; the default is the library's own pipelining (ONE C call per step).
    MODES_OF_ISSUE = ("library", "back_to_back", "overlapped", "pipelined_0_first", "pipelined_1_first")

    def issue(mode):
        """-> (results in `settings` order, the library's per-call stats)"""
        if mode == "library" and len(settings) > 1:
            results = passes_call()
            last = idx.stats(0)
            if last.passes == len(settings):   # one fused job: one set of statistics
                return results, [last]
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
    if len(settings) == 1:
        mode = "back_to_back"
    elif mode not in MODES_OF_ISSUE:
        # One untimed step of each way of issuing is recorded for information (ASGART_BENCH_MODE=auto picks the
        # fastest of them instead of the library's).
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
        issue(mode)
    sequential = mode == "back_to_back"

    # All ranks on ONE device (ASGART_BENCH_ONE_DEVICE, a 1-GPU box): the ranks take turns -- rank r runs its shard of a
    # step while the others wait at a barrier -- so that per_rank_ms is what each rank's shard costs ALONE on a GPU (what
    # its own GPU would see); ms_per_step is then the SUM of the turns, not a multi-GPU time, and the line says so.
    take_turns = world > 1 and bool(os.environ.get("ASGART_BENCH_ONE_DEVICE"))
    sync()
    t0 = time.perf_counter()
    search_ms = 0.0
    probe_count_ms = 0.0
    n_launch = 0
    own_elapsed = 0.0   # this rank's search calls ...
    own_gather = 0.0    # ... and its part of the gather of the result lists to rank 0
    phase_ms = {"search": 0.0, "scan": 0.0, "fill": 0.0, "extend": 0.0, "extend_tier2": 0.0, "longest_tier": 0.0,
                "longest_segment": 0.0}
    for _ in range(args.steps):
        for turn in (range(world) if take_turns else (rank,)):
            if turn == rank:
                t_own = time.perf_counter()
                results, per_call = issue(mode)
                own_elapsed += time.perf_counter() - t_own   # this rank's share, before it waits for the others
            if take_turns:
                sync()
        if world > 1:
            # the only exchange of the path: duplicon lists -> rank 0 over RCCL
            t_own = time.perf_counter()
            results = [gather(r_) for r_ in results]
            own_gather += time.perf_counter() - t_own
        n_launch += len(per_call)
        for s in per_call:
            search_ms += s.ms_search
            probe_count_ms += s.ms_probe_count
            for ph in phase_ms:
                phase_ms[ph] += getattr(s, "ms_" + ph)
    sync()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [round(own_elapsed / args.steps * 1e3, 3)]        # the search calls of a step: what the scaling model predicts
    per_rank_gather_ms = [round(own_gather / args.steps * 1e3, 3)]   # (RCCL over xGMI on a multi-GPU node; gloo over loopback on one device)
    per_rank_phase = [{ph: round(v / args.steps, 3) for ph, v in phase_ms.items()}]
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        box = [None] * world
        dist.all_gather_object(box, (per_rank_ms[0], per_rank_phase[0], per_rank_gather_ms[0]))
        per_rank_ms = [b[0] for b in box]
        per_rank_phase = [b[1] for b in box]
        per_rank_gather_ms = [b[2] for b in box]

    passes = len(modes)
    value = total_bp * passes * args.steps / elapsed / 1e6
    # one launch of the probe-search kernels covers passes_per_launch passes (all of them when the passes call fuses)
    n_ls = len(pass_stats)
    alg_bytes = sum(s["search_bytes"] for s in pass_stats) / n_ls          # per launch, this design
    ref_bytes = sum(reference_algorithm_bytes(s, k) for s in pass_stats) / n_ls
    avg_launch_ms = search_ms / n_launch if n_launch else 0.0
    achieved = alg_bytes / (avg_launch_ms / 1e3) / 1e9 if avg_launch_ms > 0 else 0.0
    alone_ms = sum(s["ms_search"] for s in pass_stats) / n_ls
    prof = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            prof = json.load(open(tpath)).get(args.workload) or {}
            if not isinstance(prof, dict):
                prof = {}
        except Exception:
            prof = {}
    traffic_counted = prof.get("traffic_bytes_per_launch")
    traffic_ms = prof.get("kernel_ms_per_launch")
    # FETCH_SIZE tallies exactly half the bytes of whole-wave 16-byte-per-lane loads on gfx950 (MI355X_MICROARCH.md, HBM):
    # the text windows and filter bitmaps of the probe search are such loads -- the accounting pass counts them apart
    wide_bytes = sum(s.get("search_bytes_wide_loads", 0) for s in pass_stats) / n_ls
    if traffic_counted and prof.get("passes_per_launch", 1) != passes_per_launch:
        # (the counter passes were collected with a different number of passes per launch: rescale to this run's launch)
        traffic_counted = traffic_counted * passes_per_launch / prof.get("passes_per_launch", 1)
        traffic_ms = traffic_ms * passes_per_launch / prof.get("passes_per_launch", 1) if traffic_ms else traffic_ms
    traffic = int(traffic_counted + wide_bytes / 2) if traffic_counted else None
    import hashlib
    lib_hash = hashlib.sha256(open(asgart_amd.library_path(), "rb").read()).hexdigest()[:12]
    roofline = {
        "bound": "hbm",
        "kernel": "probe_count_kernel + collect_pending_kernel + big_count_kernel + rank_count_kernel (one launch = "
                  + ("one pass)" if passes_per_launch == 1 else f"the {passes_per_launch} passes of a step as one job)"),
        "passes_per_launch": passes_per_launch,
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "kernel_algorithmic_bytes": int(alg_bytes),
        "avg_launch_ms": round(avg_launch_ms, 5),
        "launch_timing": "HIP events on the library's stream, timed region" +
                         ("" if sequential or fused else " (the other pass's kernels share the chip)"),
        # first kernel of the group alone (rocprofv3 lists the kernels separately)
        "probe_count_kernel_ms": round(probe_count_ms / n_launch, 5) if n_launch else 0.0,
        "alone_launch_ms": round(alone_ms, 5),
        "achieved_alone": round(alg_bytes / (alone_ms / 1e3) / 1e9, 2) if alone_ms > 0 else 0.0,
        "frac_alone": round(alg_bytes / (alone_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5) if alone_ms > 0 else 0.0,
        "traffic": traffic,
        "traffic_as_counted": traffic_counted,
        "wide_load_bytes": int(wide_bytes),
        "traffic_ms": traffic_ms,
        "traffic_frac": (round(traffic / (traffic_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
                         if traffic and traffic_ms else None),
        "traffic_x2_applied": "to the wide coalesced loads only: traffic = FETCH_SIZE + WRITE_SIZE as counted + wide_load_bytes / 2" if traffic else False,
        "traffic_source": prof.get("source"),
        "traffic_build": prof.get("build"),
        "library_build": lib_hash,
        "traffic_from_this_build": prof.get("build") == lib_hash if prof.get("build") else None,
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

    # Scaling model.  Rank r of N runs shard r -- the r-th slice of every pass, all passes as one job --
    # alone on its own GPU, and a step lasts as long as its slowest rank: model_ms(N) = max_r shard_ms(N, r), the shards
    # timed ONE AFTER THE OTHER ON ONE GPU by tools/shard_check.py (profiles/r06_<workload>_shards.json, committed with the
    # library hash it was measured on).  Not in it: the gather of the result lists to rank 0 (tens of MB over xGMI) and the
    # barrier.  NO 1 -> N curve has been measured on N GPUs by the builder (one GPU per gpurun call).
    model = None
    spath = os.path.join(ROOT, "profiles", f"r06_{args.workload}_shards.json")
    if rank == 0 and os.path.exists(spath):
        try:
            sh = json.load(open(spath))
            model = {"source": os.path.relpath(spath, ROOT), "measured_on_build": sh.get("library_build"),
                     "from_this_build": sh.get("library_build") == lib_hash,
                     "formula": "model_ms(N) = max_r shard_ms(N, r): every shard of a step timed alone on one GPU "
                                "(tools/shard_check.py); gather and barrier not included",
                     "curve_measured_on_n_gpus": False}
            for n_, e_ in sorted(sh.get("n", {}).items(), key=lambda kv: int(kv[0])):
                model[f"n{n_}_ms"] = e_["max_ms"]
                model[f"n{n_}_shard_ms"] = [s_["ms"] for s_ in e_["shards"]]
        except Exception as exc:   # (a damaged file must not cost the bench line)
            model = {"error": f"{spath}: {exc}"}
    if rank == 0 and whole:
        fronts = [w_["ms_search"] + w_["ms_scan"] + w_["ms_fill"] for w_ in whole]
        floor = [w_.get("ms_longest_segment") or w_["ms_longest_tier"] for w_ in whole]
        if not model or "error" in model:
            # no shard timings committed for this workload: the coarse formula of the earlier rounds, from ONE unsharded run --
            # sharding divides a job's front and the throughput part of its extension, never its longest single segment
            model = dict(model or {}, source="formula", curve_measured_on_n_gpus=False,
                         formula="sum_j front_j / N + max_j max(longest_segment_j, extend_j / N)   (j: the jobs of a step)")
            for n_ in sorted({1, 2, 4, 8, world}):
                model[f"n{n_}_ms"] = round(sum(fronts) / n_ + max(max(fl_, w_["ms_extend"] / n_) for w_, fl_ in zip(whole, floor)), 2)
        model["one_gpu_phases_ms"] = [{"front": round(f_, 2), "extend": round(w_["ms_extend"], 2), "longest_segment": round(fl_, 2)}
                                      for f_, w_, fl_ in zip(fronts, whole, floor)]

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
        "dtype": "u64 keys / u32 SA" if n_text < 0xFFFFFF00 else "u64 keys / u64 SA",
        "data": data_kind,
        "config": {"workload": desc, "bp_per_pass": total_bp, "passes": passes,
                   "skip_masked": skip_masked,
                   "text_bytes": n_text, "chunks": len(chunks),
                   "parallelism": f"probe-shard x{world}" if world > 1 else "1 GPU",
                   "ranks_launched_by": os.environ.get("ASGART_BENCH_LAUNCHED", "torchrun" if world > 1 else "none"),
                   "collective_backend": (dist.get_backend() if dist is not None else None),
                   "hw_queues_per_process": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "passes_issued": mode + (" (one fused job)" if fused and mode == "library" else ""),
                   "mode_probe_ms": mode_probe_ms},
        "roofline": roofline,
        "phases_ms_per_step": per_rank_phase[0],
        "per_rank_ms": per_rank_ms,
        "per_rank_gather_ms": per_rank_gather_ms if world > 1 else None,
        "model_vs_slowest_rank": (round((model or {}).get(f"n{world}_ms") / max(per_rank_ms), 3)
                                  if world > 1 and (model or {}).get(f"n{world}_ms") else None),
        "per_rank_phases_ms_per_step": per_rank_phase if world > 1 else None,
        "model_ms": (model or {}).get(f"n{world}_ms"),
        "scaling_model": model,
        "ranks_take_turns_on_one_device": take_turns or None,
        # a host that runs every orientation ONCE per index (the reference: src/bin/asgart.rs:738-757) sees the first call
        "first_call_ms": round(t_first * 1e3, 2),
        "first_call_mbps": round(total_bp * passes / t_first / 1e6, 1),
        "cold_s": round(cold_s, 3),
        "index_build_s": {"upload_and_suffix_array": round(t_sa, 3), "broadcast_to_ranks": round(t_bc, 3),
                          "keys_and_tables": round(t_index, 3),
                          "first_passes_call_incl_presence_filters": round(t_first, 3),
                          "warmup_steps": round(t_warm, 3),
                          "settling_steps_after_warmup": settling_steps,
                          "sa_builder": "GPU prefix doubling (asgart_sa_build64 path)"},
        "work_per_step": {key: sum(s[key] for s in pass_stats) for key in
                          ("probes_total", "probes_searched", "probes_card_skipped", "probes_filter_rejected",
                           "raw_hits", "filtered_hits", "segments", "overflow_segments", "heavy_segments",
                           "families", "proto_sds", "split_segments", "split_refused")},
        "work_per_step_scope": "rank 0's shard" if world > 1 else "whole job",
    }

    if rank == 0 and args.end_to_end:
        out["end_to_end_s"] = end_to_end(recs, files, skip_masked, modes, k, gap, local_rank)

    if rank == 0 and not args.no_cpu_baseline:
        # CPU reference: the oracle with the reference's parallel structure (OpenMP over
        # chunks == rayon par_iter, src/bin/asgart.rs:201-205), same index, same chunks.
        import oracle  # the CPU checker, used here ONLY as the timed CPU baseline

        cores = min(os.cpu_count() or 1, len(pr.chunks))  # threads that can be busy: one per chunk
        sa = np.empty(n_text, dtype=np.int64)   # (read back in slabs: sa_read returns a fresh array per call)
        slab = 1 << 28
        for o in range(0, n_text, slab):
            sa[o:min(n_text, o + slab)] = idx.sa_read(o, min(n_text, o + slab))
        oidx = oracle.Index.build(pr.data, sa)
        # bounded sample: the first `per_chunk` bases of EVERY chunk, so the CPU leg keeps the
        # reference's chunk-level parallelism (one thread per chunk) without its skew
        # ... sized for about 20 s of CPU work: a first, twenty times smaller sample tells how fast this input goes (a
        # repeat-rich genome costs the CPU path ten times more per base than the GRCh38-shaped one)
        budget_bp = float(os.environ.get("ASGART_CPU_SAMPLE_BP", 200e6))

        def cpu_sample(per_chunk_):
            chunks_ = [(s0, min(l0, per_chunk_)) for s0, l0 in pr.chunks]
            t_ = time.perf_counter()
            for r, c in modes:
                oidx.run_raw(chunks_, oracle.make_settings(k=k, gap=gap, reverse=r, complement=c), threads=cores)
            return sum(l0 for _, l0 in chunks_), time.perf_counter() - t_

        per_chunk = max(100_000, int(budget_bp / max(1, len(pr.chunks))))
        if "ASGART_CPU_SAMPLE_BP" not in os.environ and per_chunk > 2_000_000:
            # (grown at most four-fold at a time: the cost per base is far from uniform along a chunk -- the first
            # 200 kb of a record say nothing about the satellite array 3 Mb further on)
            target_s = float(os.environ.get("ASGART_CPU_SAMPLE_S", 20.0))
            cap, per_chunk = per_chunk, per_chunk // 20
            while True:
                sample_bp, t_cpu = cpu_sample(per_chunk)
                if per_chunk >= cap or t_cpu >= target_s / 4:
                    break
                per_chunk = min(cap, int(per_chunk * min(4.0 if t_cpu < 2.0 else 2.0, max(1.5, target_s / max(t_cpu, 1e-3)))))
        else:
            sample_bp, t_cpu = cpu_sample(per_chunk)
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
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
