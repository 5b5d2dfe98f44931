"""world_size-2 gloo tests (CPU) of the multi-GPU composition in hbird_mi/dist.py.

The HIP kernels cannot run here, so the per-rank searcher is the CPU oracle and the merge is a numpy
restatement of hb_merge_topk's ordering -- both are test doubles; what is under test is the sharding
arithmetic, the id bases, the ragged all-gather and the fact that an all-gather of per-shard top-k lists
followed by a k-way merge reproduces the unsharded search exactly (ties included)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import golden_inputs as gi


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _np_merge(dist_parts, idx_parts, metric):
    d = dist_parts.numpy(); i = idx_parts.numpy()
    parts, nq, k = d.shape
    out_i = np.empty((nq, k), dtype=np.int64); out_d = np.empty((nq, k), dtype=np.float32)
    for q in range(nq):
        dd = d[:, q].reshape(-1); ii = i[:, q].reshape(-1)
        key = -dd if metric == 0 else dd
        missing = ii < 0
        order = np.lexsort((ii, key, missing))[:k]
        out_i[q] = ii[order]; out_d[q] = dd[order]
    return torch.from_numpy(out_i), torch.from_numpy(out_d)


def _worker(rank, world, port, metric_name, M, D, nq, k, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "open-hummingbird-eval_amd"), os.path.join(root, "tests")]
    import oracle
    from hbird_mi import dist as hdist
    bank = gi.unit_bank(M, D, seed=1)
    bank[M - 3] = bank[5]                      # a tie that straddles the shard boundary
    q = gi.vit_like_queries(nq, D, seed=2)
    lo, hi = hdist.shard_range(M, rank, world)
    assert (lo, hi) == ((0, (M + 1) // 2) if rank == 0 else ((M + 1) // 2, M))

    def local_search(qq, kk, id_base):
        i, d = oracle.knn_chain_f32(qq.numpy(), bank[lo:hi], kk, metric_name, id_base)
        return torch.from_numpy(i), torch.from_numpy(d)

    metric = 0 if metric_name == "dot_product" else 1
    idx, dist = hdist.sharded_search(local_search, _np_merge, torch.from_numpy(q), k, lo, metric)
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric_name)
    ok = np.array_equal(idx.numpy(), ridx) and np.array_equal(dist.numpy(), rdist)
    # the same search through the in-place merge hook: ONE packed all-gather message per rank, whose byte layout
    # ([nq*k int64 ids][nq*k fp32 scores], padded to 16 B) is what hb_merge_topk_packed reads on the GPU
    seen = {}

    def merge_packed(recv, part_bytes, parts, nq_, k_, m):
        raw = recv.numpy().reshape(parts, part_bytes)
        n = nq_ * k_
        pi = np.stack([raw[p, :n * 8].copy().view(np.int64).reshape(nq_, k_) for p in range(parts)])
        pd = np.stack([raw[p, n * 8:n * 12].copy().view(np.float32).reshape(nq_, k_) for p in range(parts)])
        seen["bytes"] = (part_bytes, recv.numel())
        return _np_merge(torch.from_numpy(pd), torch.from_numpy(pi), m)

    ex = hdist.PackedTopK(nq, k, "cpu", world)
    idx2, dist2 = hdist.sharded_search(local_search, None, torch.from_numpy(q), k, lo, metric, merge_packed=merge_packed, exchange=ex)
    ok = ok and np.array_equal(idx2.numpy(), ridx) and np.array_equal(dist2.numpy(), rdist)
    ok = ok and seen["bytes"] == ((nq * k * 12 + 15) // 16 * 16, world * ((nq * k * 12 + 15) // 16 * 16))
    # ragged all-gather
    rows = torch.full((3 + 2 * rank, 4), float(rank))
    allr, counts = hdist.allgather_rows(rows)
    ok = ok and counts == [3, 5] and allr.shape == (2, 5, 4) and float(allr[1, 4, 0]) == 1.0 and float(allr[0, 4, 0]) == 0.0
    ok = ok and hdist.deal_round_robin(5, rank, world) == ([0, 2, 4] if rank == 0 else [1, 3])
    ret[rank] = bool(ok)
    td.destroy_process_group()


@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_sharded_search_equals_unsharded_world2(metric):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, metric, 2001, 32, 37, 30, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def test_shard_range_properties():
    from hbird_mi import dist as hdist
    for n in (0, 1, 7, 100, 10_000_000):
        for world in (1, 2, 3, 8):
            ranges = [hdist.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
            assert sorted(sum((hdist.deal_round_robin(n % 50, r, world) for r in range(world)), [])) == list(range(n % 50))


def test_rank_batches_fetches_only_the_kept_index_batches():
    """hdist.rank_batches on a DataLoader: the kept batches, in the order and with the content the plain loader would deliver
    (shuffled loaders included), only their items fetched from the dataset, and the default CPU generator left exactly where a
    plain pass leaves it (a sharded build replays the reference's single random stream behind it)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "open-hummingbird-eval_amd")]
    from torch.utils.data import DataLoader, Dataset
    from hbird_mi import dist as hdist
    from hbird_mi.tiling import WindowedLoader

    class DS(Dataset):
        def __init__(self): self.calls = []
        def __len__(self): return 23
        def __getitem__(self, i):
            self.calls.append(i)
            return torch.full((3, 8, 8), float(i)), torch.full((1, 8, 8), float(i))

    for shuffle in (False, True):
        ds = DS(); L = DataLoader(ds, batch_size=4, shuffle=shuffle)
        torch.manual_seed(5); ref = [b[0][:, 0, 0, 0].tolist() for b in L]; after_ref = torch.rand(3)
        ds.calls.clear()
        torch.manual_seed(5); got = list(hdist.rank_batches(L, lambda i: i % 3 == 1)); after = torch.rand(3)
        assert [(i, b[0][:, 0, 0, 0].tolist()) for i, b in got] == [(i, ref[i]) for i in range(len(ref)) if i % 3 == 1]
        assert len(ds.calls) == 8 and torch.equal(after, after_ref)
    # any other iterable: iterate and filter; a WindowedLoader: only the frame batches with a kept window are fetched
    assert [i for i, _ in hdist.rank_batches([("a",), ("b",), ("c",)], lambda i: i != 1)] == [0, 2]
    ds = DS(); W = WindowedLoader(DataLoader(ds, batch_size=4), 4, 4, frame_hw=(8, 8))      # 4 windows per frame batch
    kept = list(hdist.rank_batches(W, lambda i: 8 <= i < 12))                                    # = frame batch 2, all its windows
    assert [i for i, _ in kept] == [8, 9, 10, 11] and sorted(set(ds.calls)) == [8, 9, 10, 11]
    assert all(b[0].shape[-2:] == (4, 4) for _, b in kept)


def _gather_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "open-hummingbird-eval_amd")]
    from hbird_mi import dist as hdist
    x = torch.arange((3 + 2 * rank) * 5, dtype=torch.float32).view(-1, 5) + 100 * rank       # 3 rows on rank 0, 5 on rank 1
    a, ca = hdist.allgather_rows(x)                       # ragged: count all-reduce + read-back, padded to the largest
    b, cb = hdist.allgather_rows(x, max_rows=7)           # fixed shape: ONE collective, the counts ride in the payload
    e, ce = hdist.allgather_rows(x[:0], max_rows=7)
    ok = ca == cb == [3, 5] and ce == [0, 0] and tuple(a.shape) == (2, 5, 5) and tuple(b.shape) == (2, 7, 5)
    for r in range(2):
        ok = ok and torch.equal(a[r, :ca[r]], b[r, :cb[r]]) and bool((b[r, cb[r]:] == 0).all())
    ret[rank] = bool(ok)
    td.destroy_process_group()


def test_allgather_rows_fixed_shape_equals_ragged():
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_gather_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] and ret[1]
