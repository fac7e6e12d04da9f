"""Multi-GPU host side: gather the per-shard duplicon lists (one process per GPU).

The search itself needs no collective: every shard computes the whole front and
owns every n-th segment of each extension tier's cost-sorted list; the shards' families, merged by their
keys (segment start probe, family ordinal), ARE the unsharded result.  This module only moves those small
lists to one rank with torch.distributed (backend "nccl" == RCCL over xGMI on MI355X, "gloo" in the CPU
tests) and merges them there.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def gather_families(offs: np.ndarray, sds: np.ndarray, dist, device: Optional[str] = None,
                    dst: int = 0, keys: Optional[np.ndarray] = None) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """offs: uint64[n_fam+1], sds: uint64[n_sd,4], keys: uint64[n_fam] (asgart_families_keys) of this rank.
    Returns, on rank `dst` (None elsewhere), the families of all ranks merged by key -- or, without keys
    (contiguous shards, option shard_lpt = 0), concatenated in rank order."""
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
