"""World-size-2/3 tests of the multi-GPU host path on CPU (gloo): each rank produces the results of its share
of the chunks with the oracle (a stand-in for asgart_search_duplications_shard) -- a contiguous share
(concatenated in rank order) or every world-th chunk with family keys (merged by key, the way the shards of
the HIP path own interleaved segments); gather_families must reassemble exactly the single-process result on
rank 0."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, keyed=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import oracle
    from asgart_amd import multi, prep, synth

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = synth.make_genome([90_000, 60_000, 50_000, 40_000], seed=5, sd_per_mb=60,
                             sd_len=(1000, 6000), alu_frac=0.03, l1_frac=0.0, sat_per_record=0)
    pr = prep.prepare_records(recs)
    idx = oracle.Index.build(pr.data)
    st = oracle.make_settings()
    n = len(pr.chunks)
    if keyed:   # interleaved ownership: chunk c belongs to rank c % world; key = (chunk, family ordinal in it)
        offs, sds, keys = [0], [], []
        for c in range(rank, n, world):
            o, s_ = idx.run_raw([pr.chunks[c]], st)
            offs.extend((o[1:] + offs[-1]).tolist())
            sds.append(s_)
            keys.extend((c << 32) | j for j in range(len(o) - 1))
        offs = np.array(offs, dtype=np.uint64)
        sds = np.concatenate(sds) if sds else np.zeros((0, 4), np.uint64)
        got = multi.gather_families(offs, sds, dist, keys=np.array(keys, dtype=np.uint64))
    else:
        mine = pr.chunks[rank * n // world:(rank + 1) * n // world]
        offs, sds = idx.run_raw(mine, st)
        got = multi.gather_families(offs, sds, dist)
    if rank == 0:
        full = idx.run_raw(pr.chunks, st)
        ok = np.array_equal(got[0], full[0]) and np.array_equal(got[1], full[1]) and len(full[1]) > 0
        q.put(bool(ok))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("keyed", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_gather_families_world(world, keyed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, keyed)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _gpu_worker(rank, world, port, q):
    """The real pair: asgart_search_duplications_shard on the GPU + gather_families (every rank on device 0:
    one-GPU boxes; the transport is gloo because RCCL refuses two ranks on one device)."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import asgart_amd
    import oracle
    from asgart_amd import multi, prep, synth

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = synth.make_genome([300_000, 200_000, 120_000], seed=9, sd_per_mb=50, sd_len=(1000, 8000),
                             alu_frac=0.05, l1_frac=0.01, sat_per_record=1, sat_copies=(30, 120))
    pr = prep.prepare_records(recs)
    ok = True
    with asgart_amd.Index(pr.data, None, device=0) as idx:
        for rc in (False, True):
            st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
            offs, sds, keys = idx.search_duplications_raw(pr.chunks, st, shard=rank, n_shards=world, with_keys=True)
            got = multi.gather_families(offs, sds, dist, keys=keys)
            if rank == 0:
                full = idx.search_duplications_raw(pr.chunks, st)
                oidx = oracle.Index.build(pr.data, idx.sa_read(0, len(pr.data)))
                exp = oidx.run_raw(pr.chunks, oracle.make_settings(reverse=rc, complement=rc), threads=4)
                ok = ok and np.array_equal(got[0], full[0]) and np.array_equal(got[1], full[1])
                ok = ok and np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and len(exp[1]) > 0
            else:
                assert got is None
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_shard_on_gpu_and_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _nccl_worker(rank, world, port, q):
    """gather_families over the RCCL backend with CUDA tensors (one rank: a 1-GPU box cannot hold two)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from asgart_amd import multi

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    offs = np.array([0, 2, 5], dtype=np.uint64)
    sds = np.arange(20, dtype=np.uint64).reshape(5, 4) + (1 << 40)   # values beyond 32 bits survive the int64 payload
    got = multi.gather_families(offs, sds, dist, device="cuda:0")
    empty = multi.gather_families(np.zeros(1, np.uint64), np.zeros((0, 4), np.uint64), dist, device="cuda:0")
    q.put(bool(np.array_equal(got[0], offs) and np.array_equal(got[1], sds) and len(empty[1]) == 0 and len(empty[0]) == 1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gather_families_rccl_backend():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(0, 1, _free_port(), q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _nccl_replicate_worker(rank, world, port, q):
    """replicate_index + a sharded passes call + gather_families with backend "nccl" (= RCCL) and a text of more than
    1 GiB, so that the zero-copy views of library memory (__cuda_array_interface__) and the slab-wise broadcasts have met
    RCCL before the first real N > 1 communicator does -- one rank: a 1-GPU box cannot hold two.  The receiving side is
    walked by hand (what every rank but the source does inside replicate_index)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import asgart_amd
    from asgart_amd import multi

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    rng = np.random.default_rng(77)
    n = (1 << 30) + (1 << 27)   # 1.125 GiB of text: two slabs of text, five of suffix array
    g = rng.integers(0, 4, size=n, dtype=np.uint8)
    g[900_000_000:900_060_000] = g[1_000_000:1_060_000]           # a 60-kb duplication far apart ...
    g[1_100_000_000:1_100_030_000] = 3 - g[5_030_000:5_000_000:-1]  # ... and an inverted one (found by the -RC pass)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[g]
    del g
    text = np.concatenate([text, np.frombuffer(b"$", dtype=np.uint8)])
    chunks = [(0, len(text) - 1)]
    sts = [asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc) for rc in (False, True)]
    idx = asgart_amd.Index(text, None, device=0)
    same = multi.replicate_index(idx, dist, 0)       # the source's side: broadcasts 1-GiB views of the library's buffers
    ok = same is idx
    t_ptr, sa_ptr, width = idx.export()
    text_t = torch.as_tensor(multi._DeviceBytes(t_ptr, idx.n), device="cuda:0").clone()
    sa_t = torch.as_tensor(multi._DeviceBytes(sa_ptr, idx.n * width), device="cuda:0").clone()
    for t in (text_t, sa_t):
        for o in range(0, t.numel(), 1 << 30):
            dist.broadcast(t[o:o + (1 << 30)], src=0)
    torch.cuda.synchronize()
    rep = asgart_amd.Index.from_device(text_t.data_ptr(), idx.n, sa_t.data_ptr(), width, device=0)
    del text_t, sa_t
    whole = idx.search_duplications_passes(chunks, sts)
    parts = rep.search_duplications_passes(chunks, sts, 0, 1, with_keys=True)
    for w_, p_ in zip(whole, parts):
        got = multi.gather_families(p_[0], p_[1], dist, device="cuda:0", keys=p_[2])
        ok = ok and np.array_equal(got[0], w_[0]) and np.array_equal(got[1], w_[1]) and len(w_[1]) > 0
    rep.close()
    idx.close()
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_replicate_index_and_gather_over_rccl_with_gib_slabs():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_replicate_worker, args=(0, 1, _free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (fresh interpreters, before any
    GPU call in the parent), replicates the index by broadcast, and reports n_gpus = the ranks the process group saw.
    On a one-GPU box both ranks share device 0 over gloo (ASGART_BENCH_ONE_DEVICE)."""
    import json
    import subprocess

    env = dict(os.environ, ASGART_BENCH_ONE_DEVICE="1", ASGART_FUSE_PASSES="2")   # (2: always one job, whatever a call measures)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and len(line["per_rank_ms"]) == 2
    assert line["config"]["ranks_launched_by"] == "self" and line["config"]["collective_backend"] == "gloo"
    assert line["index_build_s"]["broadcast_to_ranks"] > 0 and line["cold_s"] > 0 and line["model_ms"] > 0
    # one device: the ranks take turns, so that per_rank_ms is each shard's cost alone on a GPU; every rank ran its slice of
    # BOTH passes as one job (what rank r of an N-GPU run executes)
    assert line["ranks_take_turns_on_one_device"] is True
    assert "one fused job" in line["config"]["passes_issued"], line["config"]["passes_issued"]
    # the same workload on one rank: same work, same results size
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tiny", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert l1["n_gpus"] == 1 and l1["config"]["bp_per_pass"] == line["config"]["bp_per_pass"]
