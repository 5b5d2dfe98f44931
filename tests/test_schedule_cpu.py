"""Invariants of the host-built kNN work list (hb_build_schedule), checked on the CPU through hb_schedule_plan:
every (query tile, bank tile) pair exactly once; every slot belongs to one block and one query tile and sees its bank
tiles in ascending order (the kernel's strict `score > threshold` tie rule relies on it); `first` marks exactly the
first segment of a slot; work is balanced over the blocks."""
import ctypes

import numpy as np
import pytest

from hbird_mi import _lib


def plan(nqt, nbt, G, panel, d=768, cluster=(1, 1)):
    """-> (rows {block, q_tile, b_tile0, n_tiles, slot, first, stride, tile0, next_tile0, progress word}, stats)."""
    stats = (ctypes.c_int64 * 8)()
    _lib.check(_lib.lib().hb_schedule_plan(nqt, nbt, G, panel, d, cluster[0], cluster[1], None, 0, stats))
    nseg = stats[1]
    buf = np.zeros((nseg, 10), dtype=np.int32)
    _lib.check(_lib.lib().hb_schedule_plan(nqt, nbt, G, panel, d, cluster[0], cluster[1], buf.ctypes.data_as(ctypes.c_void_p), nseg, stats))
    keys = ["workgroups", "segments", "slots", "panel_tiles", "max_slots_per_qtile", "query_tiles", "bank_tiles"]
    st = dict(zip(keys, list(stats)[:7]))
    st["cluster"] = (int(stats[7]) // 16, int(stats[7]) % 16)
    return buf, st


@pytest.mark.parametrize("nqt,nbt,G,panel", [
    (86, 39063, 256, 0),        # headline: 21,904 queries x 10 M rows
    (49, 8102, 256, 0),         # cfg-2
    (86, 4883, 256, 0),         # one of 8 shards of the 10 M bank
    (1, 1, 256, 0), (3, 4, 256, 0), (5, 7, 3, 2), (86, 500, 256, 32), (7, 1000, 64, 5), (2, 9, 256, 3), (86, 300, 104, 0),
])
def test_work_list_invariants(nqt, nbt, G, panel):
    segs, st = plan(nqt, nbt, G, panel)
    assert st["query_tiles"] == nqt and st["bank_tiles"] == nbt
    assert st["workgroups"] == min(G, nqt * nbt)
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    slot_q, slot_blk, slot_last, slot_first_seen = {}, {}, {}, set()
    per_block = np.zeros(st["workgroups"], dtype=np.int64)
    last_block = -1
    for blk, q, b0, n, slot, first, stride, tile0, next_tile0, member in segs.tolist():
        assert stride == 1 and member == -1
        assert 0 <= q < nqt and 0 <= b0 and n > 0 and b0 + n <= nbt
        assert blk >= last_block; last_block = blk              # segments are grouped by block
        cover[q, b0:b0 + n] += 1
        per_block[blk] += n
        if first:
            assert slot not in slot_first_seen
            slot_first_seen.add(slot); slot_q[slot] = q; slot_blk[slot] = blk
        else:
            assert slot in slot_first_seen, "a slot is continued before it was started"
        assert slot_q[slot] == q and slot_blk[slot] == blk     # one owner, one query tile per slot
        assert b0 > slot_last.get(slot, -1), "bank tiles of a slot must ascend"
        slot_last[slot] = b0 + n - 1
    assert (cover == 1).all(), "every (query tile, bank tile) pair exactly once"
    assert len(slot_first_seen) == st["slots"]
    # balance: automatic panels split evenly; forced ones within one pair per panel
    npanels = -(-nbt // st["panel_tiles"])
    assert per_block.max() - per_block.min() <= (npanels if panel else max(1, per_block.mean() * 0.03))
    # slots per query tile stay within the two-level merge's reach at k = 256 (24 groups of 24 lists)
    assert st["max_slots_per_qtile"] <= 24 * 24


def test_headline_plan_numbers():
    _, st = plan(86, 39063, 256, 0)
    assert st["panel_tiles"] == 128 and st["slots"] == 340 and st["max_slots_per_qtile"] == 4


@pytest.mark.parametrize("nqt,nbt,G,panel,cq,cb", [
    (86, 39063, 256, 0, 2, 2),      # headline, the automatic shape
    (86, 39063, 256, 0, -1, -1),    # ... selected automatically
    (86, 39063, 256, 0, 4, 2), (86, 39063, 256, 0, 2, 4), (86, 39063, 256, 0, 1, 4), (86, 39063, 256, 0, 8, 1),
    (49, 8102, 256, 0, 2, 2),       # cfg-2: odd query-tile count (one member pair idles for the last query group)
    (86, 4883, 256, 0, 2, 2),       # one of 8 shards
    (7, 1001, 64, 0, 2, 2), (5, 333, 32, 7, 2, 2), (86, 500, 256, 33, 2, 2), (3, 4, 256, 0, 2, 2), (86, 300, 104, 0, 2, 2),
])
def test_clustered_work_list_invariants(nqt, nbt, G, panel, cq, cb):
    """L2-sharing clusters: cq x cb workgroups walk (cq query tiles) x (cb interleaved bank tiles) units on a common
    clock.  Same correctness invariants as the plain list, plus: members of a cluster sit on one XCD (blocks equal
    mod 8), and at every clock tick the members that are busy hold pairs of ONE unit -- same bank-tile group for all,
    the same bank tile for equal bank way, the same query tile for equal query way."""
    segs, st = plan(nqt, nbt, G, panel, cluster=(cq, cb))
    a, b = st["cluster"]
    if cq < 0:
        assert (a, b) == (8, 1)            # 86 query tiles: 2 of 88 idle in the last group of eight (2.3 %)
    elif G % (8 * cq * cb) != 0 or nqt * nbt < G:
        assert (a, b) == (1, 1)                                  # falls back to the plain list
    else:
        assert (a, b) == (cq, cb)
    assert st["workgroups"] == min(G, nqt * nbt)
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    slot_q, slot_blk, slot_last, started = {}, {}, {}, set()
    per_block = np.zeros(st["workgroups"], dtype=np.int64)
    at = {}                       # (cluster, clock) -> list of (member, q, bank tile)
    blk_clock = {}
    last_block = -1
    for blk, q, b0, n, slot, first, stride, tile0, next_tile0, member in segs.tolist():
        assert blk >= last_block; last_block = blk
        assert stride == (b if a * b > 1 else 1) and n > 0
        tiles = b0 + stride * np.arange(n)
        assert 0 <= q < nqt and tiles[0] >= 0 and tiles[-1] < nbt
        cover[q, tiles] += 1
        per_block[blk] += n
        if first:
            assert slot not in started
            started.add(slot); slot_q[slot] = q; slot_blk[slot] = blk
        assert slot in started and slot_q[slot] == q and slot_blk[slot] == blk
        assert b0 > slot_last.get(slot, -1), "bank tiles of a slot must ascend"
        slot_last[slot] = int(tiles[-1])
        assert tile0 >= blk_clock.get(blk, 0), "a block's clock never runs backwards"
        blk_clock[blk] = tile0 + n
        assert next_tile0 >= tile0 + n                          # 0x7fffffff after the block's last segment
        if a * b > 1:
            cl, m = member // 32, member % 32
            assert 0 <= m < a * b and blk % 8 == (cl // (st["workgroups"] // (a * b) // 8))   # whole clusters per XCD
            if nqt * nbt <= 600_000 or cl % 16 == 3:            # per-tick check: all clusters of the small cases, a sample of the big
                for j in range(n):
                    at.setdefault((cl, tile0 + j), []).append((m, q, int(tiles[j])))
        else:
            assert member == -1
    assert (cover == 1).all(), "every (query tile, bank tile) pair exactly once"
    assert len(started) == st["slots"]
    for (cl, t), mem in at.items():
        assert len({m for m, _, _ in mem}) == len(mem) <= a * b
        P = st["panel_tiles"]
        assert len({(bt // P, (bt % P) // b) for _, _, bt in mem}) == 1, "members of a tick work on one bank-tile group"
        for m, q, bt in mem:
            for m2, q2, bt2 in mem:
                if m % b == m2 % b:
                    assert bt == bt2                             # equal bank way: the SAME bank tile (shared through L2)
                if m // b == m2 // b:
                    assert q == q2                               # equal query way: the SAME query tile
    if a * b > 1:
        # balance: whole units are dealt, so blocks differ by at most one tile per panel (plus idle members of a ragged
        # last query group / bank group)
        npanels = -(-nbt // st["panel_tiles"])
        busy = per_block[per_block > 0]
        assert busy.max() - np.median(busy) <= npanels + 1
        if nqt % a == 0:
            assert busy.max() - busy.min() <= 2 * npanels + 1


def plan_shared(nqt, nbt, G, panel, cluster, d=768, phased=False):
    """plan() for the work list of hb_index_set_cluster_sharing(ix, 2): every run of a query group split over all clusters of an XCD."""
    stats = (ctypes.c_int64 * 8)()
    L = _lib.lib()
    _lib.check(L.hb_schedule_plan_shared(nqt, nbt, G, panel, d, cluster[0], cluster[1], int(phased), None, 0, stats))
    buf = np.zeros((stats[1], 10), dtype=np.int32)
    _lib.check(L.hb_schedule_plan_shared(nqt, nbt, G, panel, d, cluster[0], cluster[1], int(phased), buf.ctypes.data_as(ctypes.c_void_p), stats[1], stats))
    return buf, dict(workgroups=stats[0], slots=stats[2], panel_tiles=stats[3], max_slots_per_qtile=stats[4],
                     cluster=(int(stats[7]) // 16, int(stats[7]) % 16))


@pytest.mark.parametrize("nqt,nbt,G,panel,cq,cb,phased", [
    (86, 39063, 256, 0, 4, 1, True), (86, 4883, 256, 256, 4, 2, True), (86, 4883, 256, 0, 8, 1, False),
    (86, 4883, 256, 0, 2, 1, False), (49, 8102, 256, 0, 4, 1, True), (7, 1001, 64, 0, 2, 2, False), (5, 333, 32, 7, 2, 2, True),
    (86, 300, 256, 33, 2, 2, False), (3, 4, 256, 0, 2, 2, False),
])
def test_xcd_shared_work_list_invariants(nqt, nbt, G, panel, cq, cb, phased):
    """XCD-level sharing of the query tiles: the correctness invariants of any work list (every pair once; a slot = one block, one
    query tile, ascending bank tiles; clocks never run backwards; members of a cluster on one XCD and on one unit per tick), plus
    what the dealing is for: inside a panel every cluster of an XCD works on the SAME query groups, in the same order."""
    segs, st = plan_shared(nqt, nbt, G, panel, (cq, cb), phased=phased)
    a, b = st["cluster"]
    if G % (8 * cq * cb) != 0 or nqt * nbt < G:
        assert (a, b) == (1, 1)
        return
    assert (a, b) == (cq, cb)
    g = st["workgroups"]
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    slot_q, slot_blk, slot_last, started = {}, {}, {}, set()
    per_block = np.zeros(g, dtype=np.int64)
    blk_clock = {}
    at = {}
    qg_seq = {}                   # (cluster, panel) -> query groups in the order the cluster meets them
    P = st["panel_tiles"]
    for blk, q, b0, n, slot, first, stride, tile0, next_tile0, member in segs.tolist():
        assert stride == b and n > 0
        tiles = b0 + stride * np.arange(n)
        assert 0 <= q < nqt and tiles[0] >= 0 and tiles[-1] < nbt
        cover[q, tiles] += 1
        per_block[blk] += n
        if first:
            assert slot not in started
            started.add(slot); slot_q[slot] = q; slot_blk[slot] = blk
        assert slot in started and slot_q[slot] == q and slot_blk[slot] == blk
        assert b0 > slot_last.get(slot, -1), "bank tiles of a slot must ascend"
        slot_last[slot] = int(tiles[-1])
        assert tile0 >= blk_clock.get(blk, 0)
        blk_clock[blk] = tile0 + n
        assert next_tile0 >= tile0 + n
        cl, m = member // 32, member % 32
        per_xcd = g // (a * b) // 8
        assert 0 <= m < a * b and blk % 8 == cl // per_xcd
        if m == 0:
            seq = qg_seq.setdefault((cl, b0 // P), [])
            if not seq or seq[-1] != q // a:
                seq.append(q // a)
        if nqt * nbt <= 600_000 or cl % 16 == 3:
            for j in range(n):
                at.setdefault((cl, tile0 + j), []).append((m, q, int(tiles[j])))
    assert (cover == 1).all(), "every (query tile, bank tile) pair exactly once"
    assert len(started) == st["slots"]
    for (cl, t), mem in at.items():
        assert len({m for m, _, _ in mem}) == len(mem) <= a * b
        assert len({(bt // P, (bt % P) // b) for _, _, bt in mem}) == 1
        for m, q, bt in mem:
            for m2, q2, bt2 in mem:
                if m % b == m2 % b:
                    assert bt == bt2
                if m // b == m2 // b:
                    assert q == q2
    # the point of the dealing: the clusters of an XCD meet the same query groups of a panel in the same order (a cluster may miss
    # a group whose run is shorter than the number of clusters)
    per_xcd = g // (a * b) // 8
    for (cl, pn), seq in qg_seq.items():
        assert seq == sorted(set(seq))
        union = sorted({qq for (c2, p2), s2 in qg_seq.items() if p2 == pn and c2 // per_xcd == cl // per_xcd for qq in s2})
        assert len(union) <= -(-((nqt + a - 1) // a) // 8) + 1, "an XCD's share of a panel spans few query groups"
        assert set(seq) <= set(union)
    busy = per_block[per_block > 0]
    npanels = -(-nbt // P)
    assert busy.max() - np.median(busy) <= max(3 * npanels, 0.02 * np.median(busy)) + 2, (busy.max(), np.median(busy))
    assert st["max_slots_per_qtile"] <= 24 * 24


def plan_phased(nqt, nbt, G, panel, d=384, cluster=(1, 1)):
    """-> (rows as plan(), stats, cut clocks, bounds [cut][block] relative to the block's first segment)."""
    L = _lib.lib()
    stats = (ctypes.c_int64 * 8)()
    n_cuts = ctypes.c_int(0)
    _lib.check(L.hb_schedule_plan_phased(nqt, nbt, G, panel, d, cluster[0], cluster[1], None, 0, stats, None, 0, ctypes.byref(n_cuts), None))
    nseg, nc, g = stats[1], n_cuts.value, stats[0]
    buf = np.zeros((nseg, 10), dtype=np.int32)
    clocks = np.zeros(max(nc, 1), dtype=np.int32)
    bounds = np.zeros((max(nc, 1), g), dtype=np.int32)
    _lib.check(L.hb_schedule_plan_phased(nqt, nbt, G, panel, d, cluster[0], cluster[1], buf.ctypes.data_as(ctypes.c_void_p), nseg, stats,
                                         clocks.ctypes.data_as(ctypes.c_void_p), nc, ctypes.byref(n_cuts),
                                         bounds.ctypes.data_as(ctypes.c_void_p)))
    return buf, dict(workgroups=g, slots=stats[2], cluster=(int(stats[7]) // 16, int(stats[7]) % 16)), clocks[:nc], bounds[:nc]


@pytest.mark.parametrize("nqt,nbt,G,panel,cq,cb", [
    (98, 8102, 256, 0, 1, 1),       # cfg-2, fp16 / k > 32
    (98, 196, 256, 0, 1, 1),        # cfg-1
    (172, 39063, 256, 0, 2, 4),     # 10 M rows, clustered
    (172, 39063, 256, 0, -1, -1),   # ... the automatic fp16 shape
    (7, 40, 256, 0, 1, 1), (1, 3, 256, 0, 1, 1), (16, 500, 64, 7, 2, 2), (5, 7, 3, 2, 1, 1),
])
def test_phased_work_list_invariants(nqt, nbt, G, panel, cq, cb):
    """Phased searches (pools): the same pairs as the unphased list, each once, with every block's segments cut at common clocks.
    A phase is a contiguous range of every block's segments; all its segments lie between the phase's two clocks (so the
    workgroups share every phase and the members of a cluster stay on one clock); a slot's `first` segment comes before its
    continuations, phases included; the last phase holds at least half of the tiles."""
    segs, st, clocks, bounds = plan_phased(nqt, nbt, G, panel, cluster=(cq, cb))
    ref, st0 = plan(nqt, nbt, G, panel, d=384, cluster=(cq, cb))
    g = st["workgroups"]
    assert st["slots"] == st0["slots"] and st["cluster"] == st0["cluster"]
    per_wg = (nqt * nbt) // g
    assert list(clocks) == sorted(set(clocks.tolist())) and all(2 * c <= per_wg for c in clocks)
    if per_wg >= 2:
        assert len(clocks) >= 1 and clocks[0] == 1
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    started = {}
    by_block = {}
    for row in segs.tolist():
        by_block.setdefault(row[0], []).append(row)
    edges = [0] + clocks.tolist() + [2 ** 31 - 1]
    tiles_in_phase = np.zeros(len(clocks) + 1, dtype=np.int64)
    for blk in range(g):
        rows = by_block.get(blk, [])
        cuts = [0] + [int(bounds[p, blk]) for p in range(len(clocks))] + [len(rows)]
        assert cuts == sorted(cuts)
        for p in range(len(cuts) - 1):
            for _, q, b0, n, slot, first, stride, tile0, next_tile0, member in rows[cuts[p]:cuts[p + 1]]:
                assert edges[p] <= tile0 and tile0 + n <= edges[p + 1], "a segment lies inside its phase's clock range"
                tiles_in_phase[p] += n
                tiles = b0 + stride * np.arange(n)
                cover[q, tiles] += 1
                if first:
                    assert slot not in started
                    started[slot] = (p, blk, q, int(tiles[-1]))
                else:
                    p0, blk0, q0, last = started[slot]
                    assert p0 <= p and blk0 == blk and q0 == q and b0 > last
                    started[slot] = (p0, blk, q, int(tiles[-1]))
    assert (cover == 1).all()
    assert len(started) == st["slots"]
    if len(clocks):
        assert tiles_in_phase[-1] * 2 >= tiles_in_phase.sum() * 0.95
        assert (tiles_in_phase[:-1] > 0).all()
    # the same pairs per slot as the unphased list (only cut finer)
    def pairs(rows):
        out = {}
        for _, q, b0, n, slot, first, stride, *_ in rows.tolist():
            out.setdefault(slot, []).extend((q, b0 + stride * j) for j in range(n))
        return out
    if nqt * nbt <= 1_000_000:
        assert pairs(segs) == pairs(ref)


def test_automatic_cluster_shape():
    """The widest query way that idles at most 1 / 16 of the pairs (fp16 candidate kernel; the fp32 kernel: 2.5 %); none for small
    searches or unsuitable grids."""
    for nqt, nbt, G, want in ((86, 39063, 256, (8, 1)), (49, 8102, 256, (4, 2)), (48, 8102, 256, (8, 1)), (52, 8102, 256, (4, 2)),
                              (51, 8102, 256, (4, 2)), (50, 8102, 256, (4, 2)), (65, 8102, 256, (4, 2)), (45, 8102, 256, (8, 1)), (1, 100000, 256, (1, 1)), (86, 100, 256, (1, 1)), (86, 39063, 104, (1, 1)), (86, 39063, 32, (2, 2))):
        assert plan(nqt, nbt, G, 0, cluster=(-1, -1))[1]["cluster"] == want, (nqt, nbt, G)
    # the fp32 kernel's automatic shape (cluster_q = -2): 2 x 4, else 2 x 2
    for nqt, nbt, G, want in ((86, 39063, 256, (2, 4)), (49, 8102, 256, (2, 4)), (86, 4883, 256, (2, 4)), (1, 100000, 256, (1, 1)),
                              (86, 100, 256, (1, 1)), (86, 39063, 32, (2, 2)), (86, 39063, 104, (1, 1))):
        assert plan(nqt, nbt, G, 0, cluster=(-2, -2))[1]["cluster"] == want, (nqt, nbt, G)


def test_headline_clustered_plan_numbers():
    _, st = plan(86, 39063, 256, 0, cluster=(2, 2))
    assert st["cluster"] == (2, 2) and st["panel_tiles"] == 128 and st["slots"] <= 440 and st["max_slots_per_qtile"] <= 8


def test_multi_index_row_planning_without_a_gpu(monkeypatch):
    """Host logic of HipMultiIndex (one process, several GPUs): reserve() plans equal contiguous row ranges, add() fills the shards
    one after the other and splits a batch that straddles a boundary, rows beyond the plan go to the last shard, replicas get
    every row; shard_bases are the successive-id offsets (faiss.IndexShards, search_faiss.py:56-63)."""
    import numpy as np
    from hbird_mi.nn import search_hip

    class FakeIndex:
        def __init__(self, d, metric, device):
            self.d, self.metric, self.device, self.rows, self.reserved = d, metric, device, [], 0
        ntotal = property(lambda self: sum(len(r) for r in self.rows))
        def reserve(self, n): self.reserved = n
        def add(self, x, normalize=False): self.rows.append(np.asarray(x)[:, 0].copy())
        def use_current_stream(self): pass
        def set_fp16(self, e): self.fp16 = e
        def close(self): pass

    class NoDevice:
        def __init__(self, *a): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False

    monkeypatch.setattr(search_hip, "HipFlatIndex", FakeIndex)
    monkeypatch.setattr(search_hip.torch.cuda, "device", NoDevice)
    m = search_hip.HipMultiIndex(4, 0, [0, 1, 2], shard=True)
    m.reserve(100)                                            # 34 + 34 + 32 planned
    assert [ix.reserved for ix in m.indexes] == [34, 34, 34]
    rows = np.arange(130, dtype=np.float32)[:, None].repeat(4, 1)
    for a, b in ((0, 20), (20, 50), (50, 51), (51, 130)):     # 20..50 straddles the first boundary, 51..130 overruns the plan
        m.add(rows[a:b])
    assert m.shard_rows == [34, 34, 62] and m.shard_bases == [0, 34, 68] and m.ntotal == 130
    got = np.concatenate([np.concatenate(ix.rows) for ix in m.indexes])
    assert np.array_equal(got, np.arange(130))                # successive ids in the order the rows arrived
    r = search_hip.HipMultiIndex(4, 0, [0, 0], shard=False)
    r.reserve(10); r.add(rows[:7]); r.add(rows[7:10])
    assert r.shard_rows == [10, 10] and r.shard_bases == [0, 0] and r.ntotal == 10
    u = search_hip.HipMultiIndex(4, 0, [0, 1], shard=True)    # no plan: everything into the first shard
    u.add(rows[:9])
    assert u.shard_rows == [9, 0]
    m.set_fp16(2)
    assert all(ix.fp16 == 2 for ix in m.indexes)
    # GPUs without peer access to the home device: one warning, cross-device tensors go through the host
    monkeypatch.setattr(search_hip.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(search_hip.torch.cuda, "can_device_access_peer", lambda a, b: {a, b} != {0, 2})
    with pytest.warns(RuntimeWarning, match="no peer access"):
        s = search_hip.HipMultiIndex(4, 0, [0, 1, 2], shard=True)
    assert s._staged == {2}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert search_hip.HipMultiIndex(4, 0, [0, 1], shard=True)._staged == set()


@pytest.mark.parametrize("nqt,nbt,G,panel,cq,cb,shared", [
    (86, 39063, 256, 0, 1, 1, 0),       # headline, plain list
    (86, 39063, 256, 0, 2, 4, 0),       # headline, the fp32 kernel's 2 x 4 clusters
    (86, 39063, 256, 0, 8, 1, 1),       # headline, the fp16 kernel's 8 x 1 clusters with XCD-level query sharing
    (49, 8102, 256, 0, 1, 1, 0), (7, 1000, 64, 5, 1, 1, 0), (86, 4883, 256, 0, 2, 2, 0),
])
def test_weighted_work_list_invariants_and_shares(nqt, nbt, G, panel, cq, cb, shared):
    """hb_index_set_xcd_weights: uneven work shares per XCD group (blocks equal mod 8).  Whatever the shares: every (query tile, bank tile)
    pair exactly once, one owner and one query tile per slot, ascending bank tiles per slot -- and the groups' totals follow the shares to
    within a fraction of a percent (the per-panel rounding is dithered, so it does not add up over 300 panels)."""
    w = [1.006, 0.987, 1.005, 0.989, 1.004, 0.995, 1.004, 0.997]
    stats = (ctypes.c_int64 * 8)()
    wa = (ctypes.c_double * 8)(*w)
    L = _lib.lib()
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 768, cq, cb, shared, wa, None, 0, stats))
    nseg = stats[1]
    buf = np.zeros((nseg, 10), dtype=np.int32)
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 768, cq, cb, shared, wa, buf.ctypes.data_as(ctypes.c_void_p), nseg, stats))
    Gs = int(stats[0])
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    per_block = np.zeros(Gs, dtype=np.int64)
    slot_q, slot_blk, slot_last = {}, {}, {}
    for blk, q, b0, n, slot, first, stride, tile0, next_tile0, member in buf.tolist():
        tiles = b0 + stride * np.arange(n)
        assert tiles[-1] < nbt
        cover[q, tiles] += 1
        per_block[blk] += n
        if first:
            assert slot not in slot_q
            slot_q[slot] = q; slot_blk[slot] = blk
        assert slot_q[slot] == q and slot_blk[slot] == blk
        assert b0 > slot_last.get(slot, -1)
        slot_last[slot] = int(tiles[-1])
    assert (cover == 1).all()
    # equal weights are the unweighted list, segment for segment
    stats2 = (ctypes.c_int64 * 8)()
    ones = (ctypes.c_double * 8)(*[1.0] * 8)
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 768, cq, cb, shared, ones, None, 0, stats2))
    b1 = np.zeros((stats2[1], 10), dtype=np.int32); b2 = np.zeros((stats2[1], 10), dtype=np.int32)
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 768, cq, cb, shared, ones, b1.ctypes.data_as(ctypes.c_void_p), stats2[1], stats2))
    if shared:
        _lib.check(L.hb_schedule_plan_shared(nqt, nbt, G, panel, 768, cq, cb, 0, b2.ctypes.data_as(ctypes.c_void_p), stats2[1], stats2))
    else:
        _lib.check(L.hb_schedule_plan(nqt, nbt, G, panel, 768, cq, cb, b2.ctypes.data_as(ctypes.c_void_p), stats2[1], stats2))
    assert np.array_equal(b1, b2)
    # the groups' totals follow the shares, measured against the equal-share list (whose ragged last query group idles some members)
    if Gs % 8 == 0 and nbt > 4000:      # (with under one pair per workgroup and panel the equal-share list itself is uneven)
        base = np.zeros(Gs, dtype=np.int64)
        np.add.at(base, b2[:, 0], b2[:, 3])
        share = np.array([per_block[x::8].sum() / base[x::8].sum() for x in range(8)], dtype=np.float64)
        want = np.array(w) / np.mean(w)
        assert np.abs(share / want - 1.0).max() < 0.006, (share / want).tolist()


def test_weighted_work_list_random_small_shapes():
    """Extreme shares (0.3 ... 3.5) on small and ragged shapes, every list form: still every pair exactly once, one owner and one query tile
    per slot, ascending bank tiles, no empty segment (explicit shares apply to fp32 searches of any size: hb_index_set_xcd_weights mode 2)."""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for _ in range(150):
        nqt = int(rng.integers(1, 60)); nbt = int(rng.integers(1, 900)); G = int(rng.choice([8, 16, 64, 256])); panel = int(rng.choice([0, 1, 4, 16]))
        cq, cb = [(1, 1), (2, 2), (2, 4), (1, 1), (1, 2)][int(rng.integers(0, 5))]
        shared = int(rng.integers(0, 2))
        wa = (ctypes.c_double * 8)(*rng.uniform(0.3, 3.5, size=8).tolist())
        stats = (ctypes.c_int64 * 8)()
        if L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 768, cq, cb, shared, wa, None, 0, stats) != 0:
            continue          # (a cluster shape the planner refuses for this G)
        nseg = stats[1]
        buf = np.zeros((nseg, 10), dtype=np.int32)
        _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 768, cq, cb, shared, wa, buf.ctypes.data_as(ctypes.c_void_p), nseg, stats))
        cover = np.zeros((nqt, nbt), dtype=np.int32)
        slot_q, slot_blk, slot_last = {}, {}, {}
        for blk, q, b0, n, slot, first, stride, tile0, next_tile0, member in buf.tolist():
            assert n > 0 and q < nqt
            tiles = b0 + stride * np.arange(n)
            assert tiles[-1] < nbt
            cover[q, tiles] += 1
            if first:
                assert slot not in slot_q
                slot_q[slot] = q; slot_blk[slot] = blk
            assert slot_q[slot] == q and slot_blk[slot] == blk and b0 > slot_last.get(slot, -1)
            slot_last[slot] = int(tiles[-1])
        assert (cover == 1).all(), (nqt, nbt, G, panel, cq, cb, shared)


@pytest.mark.parametrize("nqt,nbt,G,panel,cq,cb", [(49, 8102, 256, 0, 1, 1), (86, 39063, 256, 128, 1, 1), (86, 19532, 256, 128, 2, 4), (49, 196, 256, 0, 1, 1)])
def test_phased_list_with_uneven_shares(nqt, nbt, G, panel, cq, cb):
    """A phased list (pool searches) under uneven XCD shares: the cuts follow the shares (a worker with 1 % more pairs has its cuts 1 % later), and
    still every pair exactly once, one owner per slot, ascending bank tiles per slot, every workgroup's clock strictly increasing."""
    L = _lib.lib()
    w = [1.012, 0.985, 1.01, 0.988, 1.008, 0.992, 1.006, 0.999]
    wa = (ctypes.c_double * 8)(*w)
    stats = (ctypes.c_int64 * 8)()
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 384, cq, cb, 2, wa, None, 0, stats))
    nseg = stats[1]
    buf = np.zeros((nseg, 10), dtype=np.int32)
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 384, cq, cb, 2, wa, buf.ctypes.data_as(ctypes.c_void_p), nseg, stats))
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    slot_q, slot_blk, slot_last, clock = {}, {}, {}, {}
    for blk, q, b0, n, slot, first, stride, tile0, next_tile0, member in buf.tolist():
        assert n > 0
        tiles = b0 + stride * np.arange(n)
        cover[q, tiles] += 1
        if slot not in slot_q:
            assert first
            slot_q[slot] = q; slot_blk[slot] = blk
        assert slot_q[slot] == q and slot_blk[slot] == blk and b0 > slot_last.get(slot, -1)
        slot_last[slot] = int(tiles[-1])
        assert tile0 >= clock.get(blk, 0)
        clock[blk] = tile0 + n
    assert (cover == 1).all()
    # the unweighted phased list has (many) more segments than the unphased one: the cuts are there
    s0 = (ctypes.c_int64 * 8)(); s1 = (ctypes.c_int64 * 8)()
    _lib.check(L.hb_schedule_plan_weighted(nqt, nbt, G, panel, 384, cq, cb, 0, wa, None, 0, s0))
    assert stats[1] > s0[1]
