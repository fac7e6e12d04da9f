"""Multi-GPU host side: gather the per-shard duplicon lists (one process per GPU).

The search itself needs no collective: shard r returns the families of the
segments that start in its slice of the probe sequence, and the shards concatenated in rank order
ARE the unsharded result.  This module only moves those small lists to one rank with
torch.distributed (backend "nccl" == RCCL over xGMI on MI355X, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def gather_families(offs: np.ndarray, sds: np.ndarray, dist, device: Optional[str] = None,
                    dst: int = 0) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """offs: uint64[n_fam+1], sds: uint64[n_sd,4] of this rank.  Returns the concatenation over
    ranks (in rank order) on rank `dst`, None elsewhere."""
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
    # one padded int64 payload per rank: family sizes, then the 4 x n_sd coordinates
    payload = torch.zeros(max_f + 4 * max_s + 1, dtype=torch.int64, device=dev)
    payload[:len(fam_sizes)] = torch.from_numpy(fam_sizes).to(dev)
    if len(sds):
        payload[max_f:max_f + 4 * len(sds)] = torch.from_numpy(
            sds.astype(np.int64).reshape(-1)).to(dev)
    bufs = [torch.zeros_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None
    sizes, recs = [], []
    for r in range(world):
        nf, ns = int(all_counts[r][0]), int(all_counts[r][1])
        b = bufs[r].cpu().numpy()
        sizes.append(b[:nf])
        recs.append(b[max_f:max_f + 4 * ns].reshape(ns, 4))
    sizes = np.concatenate(sizes) if sizes else np.zeros(0, np.int64)
    out_offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    out_sds = (np.concatenate(recs) if recs else np.zeros((0, 4), np.int64)).astype(np.uint64)
    return out_offs, out_sds
