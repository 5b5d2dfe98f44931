"""Multi-GPU composition of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  The path shards by bank rows (faiss.IndexShards of the reference,
hbird/nn/search_faiss.py:53-63: contiguous row ranges, successive ids) and has ONE exchange step per search:
an all-gather of the per-rank top-k lists followed by a local k-way merge.

Message sizes at cfg-3 (21,904 queries, k = 30): 21,904 x 30 x (8 + 4) B = 7.9 MB per rank -- latency-bound on
7 x 153 GB/s xGMI links, so ids and scores travel PACKED in one flat all-gather per search (`PackedTopK`: no
bucketing, no ring pipelining, one message per rank) and the merge kernel reads the gathered buffer in place.
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


def rank_batches(loader, keep: Callable[[int], bool]):
    """One pass over `loader`, yielding `(ordinal, batch)` for the batches whose ordinal `keep` accepts -- WITHOUT decoding the others
    when `loader` is a torch DataLoader over a map-style dataset (a sharded bank build must not decode ADE20K once per rank: rank r
    of N touches 1 / N of the images).  The epoch's batch order is drawn from the loader's own batch sampler exactly as iter(loader)
    draws it (same generator, same number and order of draws from the default CPU generator: ranks whose generators were aligned
    agree on the order, and the stream the reference's single process would see afterwards is the same), then a second DataLoader
    over the same dataset fetches only the kept index batches (same workers, collate_fn, pin_memory).  `loader` may offer its own
    `rank_batches(keep)` (tiling.WindowedLoader).  Any other iterable is iterated and filtered."""
    if hasattr(loader, "rank_batches"):
        yield from loader.rank_batches(keep)
        return
    from torch.utils.data import DataLoader, IterableDataset
    if (isinstance(loader, DataLoader) and loader.batch_sampler is not None and not isinstance(loader.dataset, IterableDataset)
            and getattr(loader, "_auto_collation", True)):
        # iter(loader) draws the iterator's base seed first, then (at the first fetch) whatever the sampler draws: same here
        base_seed = int(torch.empty((), dtype=torch.int64).random_(generator=loader.generator).item())
        index_batches = list(loader.batch_sampler)
        owned = [(i, b) for i, b in enumerate(index_batches) if keep(i)]
        if not owned:
            return
        g = torch.Generator()
        g.manual_seed(base_seed)           # the sub-loader seeds its workers from a private generator: the default stream is untouched
        kw = dict(num_workers=loader.num_workers, collate_fn=loader.collate_fn, pin_memory=loader.pin_memory,
                  worker_init_fn=loader.worker_init_fn, timeout=loader.timeout, generator=g)
        if loader.num_workers > 0:
            kw.update(prefetch_factor=loader.prefetch_factor, multiprocessing_context=loader.multiprocessing_context)
        sub = DataLoader(loader.dataset, batch_sampler=[b for _, b in owned], **kw)
        for (i, _), batch in zip(owned, sub):
            yield i, batch
        return
    for i, batch in enumerate(loader):
        if keep(i):
            yield i, batch


def allgather_rows(x: torch.Tensor, max_rows: Optional[int] = None) -> Tuple[torch.Tensor, List[int]]:
    """All-gather of row blocks with ragged row counts: returns ([world, max_rows, ...] zero-padded, counts).
    With `max_rows` (an upper bound every rank knows, e.g. batch size x tokens) the gathered shape is FIXED: the row counts ride
    in one extra trailing row of the payload, so a step costs ONE collective and no host round trip before it (the counts are read
    back after the gather, together with whatever the caller synchronises on anyway); without it: an all-reduce of the counts and a
    host read-back decide the padded shape first."""
    rank, world = rank_world()
    if max_rows is not None:
        assert x.shape[0] <= max_rows, f"allgather_rows: {x.shape[0]} rows exceed max_rows={max_rows}"
        pad = torch.zeros((max_rows + 1,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        pad[:x.shape[0]] = x
        pad[max_rows].view(-1)[0] = x.shape[0]             # exact in every dtype used here up to 2^24 rows (fp32) / 255 (uint8: not used)
        out = torch.empty((world * (max_rows + 1),) + tuple(pad.shape[1:]), dtype=x.dtype, device=x.device)
        if world > 1:
            td.all_gather_into_tensor(out, pad)
        else:
            out.copy_(pad)
        out = out.view((world, max_rows + 1) + tuple(pad.shape[1:]))
        counts = [int(v) for v in out[:, max_rows].reshape(world, -1)[:, 0].cpu().tolist()]
        return out[:, :max_rows], counts
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


class PackedTopK:
    """One rank's top-k lists as ONE message: [nq*k int64 ids][nq*k fp32 scores] (+ padding to 16 B; the layout of
    hb_packed_list_bytes / hb_merge_topk_packed), and the receive buffer of the all-gather ([world] such lists).
    `idx` / `dist` are [nq, k] views into the send buffer: a search may write them directly."""

    def __init__(self, nq: int, k: int, device, world: int):
        n = nq * k
        self.nq, self.k, self.world = nq, k, world
        self.part_bytes = (n * 12 + 15) // 16 * 16
        self.send = torch.zeros(self.part_bytes, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(world * self.part_bytes, dtype=torch.uint8, device=device)
        self.idx = self.send[:n * 8].view(torch.int64).view(nq, k)
        self.dist = self.send[n * 8:n * 12].view(torch.float32).view(nq, k)

    def gather(self, async_op: bool = False):
        """The exchange step: one all-gather of the packed lists (RCCL on the GPU box)."""
        if self.world == 1:
            self.recv.copy_(self.send)
            return None
        return td.all_gather_into_tensor(self.recv, self.send, async_op=async_op)

    def parts(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(dist_parts [world, nq, k], idx_parts [world, nq, k]) copies of the gathered lists (generic merges)."""
        n = self.nq * self.k
        r = self.recv.view(self.world, self.part_bytes)
        pi = r[:, :n * 8].contiguous().view(torch.int64).view(self.world, self.nq, self.k)
        pd = r[:, n * 8:n * 12].contiguous().view(torch.float32).view(self.world, self.nq, self.k)
        return pd, pi


def sharded_search(local_search: Callable, merge: Callable, q: torch.Tensor, k: int, id_base: int, metric: int,
                   finish: Optional[Callable] = None, merge_packed: Optional[Callable] = None,
                   exchange: Optional[PackedTopK] = None):
    """Every rank searches ALL queries on its shard, the per-rank lists are all-gathered and merged.

    local_search(q, k, id_base) -> (idx int64 [nq,k] global ids, dist fp32 [nq,k]);
    merge(dist_parts [world,nq,k], idx_parts [world,nq,k], metric) -> (idx, dist).
    With `finish`, local_search returns ORDERING scores (larger is better) instead of distances, the merge runs on them
    (metric 0 ordering) and finish(q, scores) -> distances converts the merged list: squared L2 distances round away
    score differences that the single-index search still orders by, so only this reproduces it bit for bit.
    merge_packed(recv, part_bytes, world, nq, k, metric) -> (idx, dist), when given, merges the gathered PackedTopK
    buffer in place (hb_merge_topk_packed); `exchange` re-uses a PackedTopK across calls of one shape.
    Every rank ends up with the same merged result (faiss.IndexShards semantics)."""
    rank, world = rank_world()
    idx, dist = local_search(q, k, id_base)
    if world == 1:
        return idx, (finish(q, dist) if finish else dist)
    nq = idx.shape[0]
    ex = exchange if exchange is not None and (exchange.nq, exchange.k, exchange.world) == (nq, k, world) \
        else PackedTopK(nq, k, idx.device, world)
    ex.idx.copy_(idx)
    ex.dist.copy_(dist)
    ex.gather()
    m = 0 if finish else metric
    if merge_packed is not None:
        mi, md = merge_packed(ex.recv, ex.part_bytes, world, nq, k, m)
    else:
        pd, pi = ex.parts()
        mi, md = merge(pd, pi, m)
    return mi, (finish(q, md) if finish else md)
