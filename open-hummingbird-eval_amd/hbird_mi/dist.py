"""Multi-GPU composition of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  The path shards by bank rows (faiss.IndexShards of the reference,
hbird/nn/search_faiss.py:53-63: contiguous row ranges, successive ids) and has ONE exchange step per search:
an all-gather of the per-rank top-k lists followed by a local k-way merge.

Message sizes at cfg-3 (21,904 queries, k = 30): 21,904 x 30 x (8 + 4) B = 7.9 MB per rank -- latency-bound on
7 x 153 GB/s xGMI links, so a single flat all-gather per tensor is used (no bucketing, no ring pipelining).
"""
from __future__ import annotations

from typing import Callable, Optional, List, Tuple

import torch
import torch.distributed as td


def rank_world() -> Tuple[int, int]:
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range of `rank` when n rows are split over `world` shards (successive ids)."""
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per)


def deal_round_robin(n_items: int, rank: int, world: int) -> List[int]:
    """Items (validation batches) handled by `rank`; all ranks step ceil(n/world) times together."""
    return list(range(rank, n_items, world))


def allgather_rows(x: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
    """All-gather of row blocks with ragged row counts: returns ([world, max_rows, ...] zero-padded, counts)."""
    rank, world = rank_world()
    n = torch.zeros(world, dtype=torch.int64, device=x.device)
    n[rank] = x.shape[0]
    if world > 1:
        td.all_reduce(n)
    counts = n.cpu().tolist()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[:x.shape[0]] = x
    # concatenated-along-dim-0 output: the one layout every backend (RCCL, gloo) accepts
    out = torch.empty((world * mx,) + tuple(pad.shape[1:]), dtype=x.dtype, device=x.device)
    if world > 1:
        td.all_gather_into_tensor(out, pad)
    else:
        out[:mx] = pad
    return out.view((world,) + tuple(pad.shape)), counts


def sharded_search(local_search: Callable, merge: Callable, q: torch.Tensor, k: int, id_base: int, metric: int,
                   finish: Optional[Callable] = None):
    """Every rank searches ALL queries on its shard, the per-rank lists are all-gathered and merged.

    local_search(q, k, id_base) -> (idx int64 [nq,k] global ids, dist fp32 [nq,k]);
    merge(dist_parts [world,nq,k], idx_parts [world,nq,k], metric) -> (idx, dist).
    With `finish`, local_search returns ORDERING scores (larger is better) instead of distances, the merge runs on them
    (metric 0 ordering) and finish(q, scores) -> distances converts the merged list: squared L2 distances round away
    score differences that the single-index search still orders by, so only this reproduces it bit for bit.
    Every rank ends up with the same merged result (faiss.IndexShards semantics)."""
    rank, world = rank_world()
    idx, dist = local_search(q, k, id_base)
    if world == 1:
        return idx, (finish(q, dist) if finish else dist)
    nq = idx.shape[0]
    pi = torch.empty((world * nq, k), dtype=idx.dtype, device=idx.device)
    pd = torch.empty((world * nq, k), dtype=dist.dtype, device=dist.device)
    td.all_gather_into_tensor(pi, idx.contiguous())
    td.all_gather_into_tensor(pd, dist.contiguous())
    mi, md = merge(pd.view(world, nq, k), pi.view(world, nq, k), 0 if finish else metric)
    return mi, (finish(q, md) if finish else md)
