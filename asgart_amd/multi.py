"""Multi-GPU host side (one process per GPU): replicate the index, gather the per-shard duplicon lists.

The search itself needs no collective: shard r owns the automaton segments that START in the
r-th contiguous slice of every pass's probe sequence and searches those slices plus halos.  The shards' families,
MERGED BY THEIR KEYS (segment start probe, family ordinal: asgart_families_keys), are the unsharded result; without
keys gather_families concatenates in rank order, which is the same thing for contiguous slices.
The two exchanges of the arrangement, both with torch.distributed (backend "nccl" == RCCL over xGMI on MI355X, "gloo"
in the CPU tests and on one-GPU boxes):
  replicate_index   text + suffix array broadcast from the rank that sorted the suffixes;
  gather_families   the small per-shard lists to one rank, merged there (section 8e (2)).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def gather_families(offs: np.ndarray, sds: np.ndarray, dist, device: Optional[str] = None,
                    dst: int = 0, keys: Optional[np.ndarray] = None) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """offs: uint64[n_fam+1], sds: uint64[n_sd,4], keys: uint64[n_fam] (asgart_families_keys) of this rank.
    Returns, on rank `dst` (None elsewhere), the families of all ranks merged by key -- or, without keys
    (contiguous shards), concatenated in rank order."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device) if device else torch.device("cpu")
    fam_sizes = np.diff(offs.astype(np.int64))
    counts = torch.tensor([len(fam_sizes), len(sds)], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = [c.cpu().numpy() for c in all_counts]
    max_f = max(int(c[0]) for c in all_counts)
    max_s = max(int(c[1]) for c in all_counts)
    # one padded int64 payload per rank: family sizes, family keys, then the 4 x n_sd coordinates
    payload = torch.zeros(2 * max_f + 4 * max_s + 1, dtype=torch.int64, device=dev)
    payload[:len(fam_sizes)] = torch.from_numpy(fam_sizes).to(dev)
    if keys is not None and len(keys):
        payload[max_f:max_f + len(keys)] = torch.from_numpy(keys.astype(np.uint64).view(np.int64)).to(dev)
    if len(sds):
        payload[2 * max_f:2 * max_f + 4 * len(sds)] = torch.from_numpy(
            sds.astype(np.int64).reshape(-1)).to(dev)
    bufs = [torch.zeros_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        nf, ns = int(all_counts[r][0]), int(all_counts[r][1])
        b = bufs[r].cpu().numpy()
        o = np.concatenate([[0], np.cumsum(b[:nf])]).astype(np.uint64)
        parts.append((o, b[2 * max_f:2 * max_f + 4 * ns].reshape(ns, 4).astype(np.uint64),
                      b[max_f:max_f + nf].view(np.uint64)))
    if keys is not None:
        from . import merge_shards
        return merge_shards(parts)
    sizes = np.concatenate([np.diff(p[0].astype(np.int64)) for p in parts]) if parts else np.zeros(0, np.int64)
    out_offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    out_sds = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 4), np.uint64)
    return out_offs, out_sds


class _DeviceBytes:
    """A device address range as an object torch.as_tensor understands (zero-copy view of library memory)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def replicate_index(idx, dist, device_index: int, src: int = 0, piece: int = 1 << 30):
    """Every rank ends up with an index of the same text on ITS GPU, the suffixes sorted once: rank `src` passes the
    index it built (the others pass None); text and suffix array travel device to device as broadcasts of `piece`-byte
    slabs straight out of the library's buffers (asgart_index_export) and every other rank copies what it received
    into a replica (asgart_index_create_device).  Returns the rank's index (rank src: `idx` itself)."""
    import torch

    from . import Index

    rank = dist.get_rank()
    dev = torch.device("cuda", device_index)
    nccl = dist.get_backend() == "nccl"
    meta = torch.zeros(2, dtype=torch.int64, device=dev if nccl else "cpu")
    if rank == src:
        t_ptr, sa_ptr, width = idx.export()
        n = idx.n
        meta[0], meta[1] = n, width
    dist.broadcast(meta, src=src)
    n, width = int(meta[0].item()), int(meta[1].item())
    if rank == src:
        text_t = torch.as_tensor(_DeviceBytes(t_ptr, n), device=dev)
        sa_t = torch.as_tensor(_DeviceBytes(sa_ptr, n * width), device=dev)
    else:
        text_t = torch.empty(n, dtype=torch.uint8, device=dev)
        sa_t = torch.empty(n * width, dtype=torch.uint8, device=dev)
    for t in (text_t, sa_t):
        for o in range(0, t.numel(), piece):
            dist.broadcast(t[o:o + piece], src=src)
    torch.cuda.synchronize(dev)
    if rank == src:
        return idx
    out = Index.from_device(text_t.data_ptr(), n, sa_t.data_ptr(), width, device=device_index)
    del text_t, sa_t
    torch.cuda.empty_cache()   # (torch's caching allocator would keep the 15 GB staging tensors)
    return out
