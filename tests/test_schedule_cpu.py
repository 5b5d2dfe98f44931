"""Invariants of the host-built kNN work list (hb_build_schedule), checked on the CPU through hb_schedule_plan:
every (query tile, bank tile) pair exactly once; every slot belongs to one block and one query tile and sees its bank
tiles in ascending order (the kernel's strict `score > threshold` tie rule relies on it); `first` marks exactly the
first segment of a slot; work is balanced over the blocks.  Both ways of dealing a panel (XCD grid rounds / linear
ranges), plus what the grid promises: workgroups of one XCD share bank ranges and query tiles in step."""
import ctypes
from collections import defaultdict

import numpy as np
import pytest

from hbird_mi import _lib

GRID, LINEAR = 0, 1


def plan(nqt, nbt, G, panel, d=768, mode=GRID):
    stats = (ctypes.c_int64 * 8)()
    _lib.check(_lib.lib().hb_schedule_plan(nqt, nbt, G, panel, d, mode, None, 0, stats))
    nseg = stats[1]
    buf = np.zeros((nseg, 7), dtype=np.int32)
    _lib.check(_lib.lib().hb_schedule_plan(nqt, nbt, G, panel, d, mode, buf.ctypes.data_as(ctypes.c_void_p), nseg, stats))
    keys = ["workgroups", "segments", "slots", "panel_tiles", "max_slots_per_qtile", "query_tiles", "bank_tiles", "mode"]
    return buf, dict(zip(keys, list(stats)[:8]))


CASES = [
    (86, 39063, 256, 0),        # headline: 21,904 queries x 10 M rows
    (49, 8102, 256, 0),         # cfg-2
    (86, 4883, 256, 0),         # one of 8 shards of the 10 M bank
    (86, 79474, 256, 0),        # cfg-4's bank
    (1, 1, 256, 0), (3, 4, 256, 0), (5, 7, 3, 2), (86, 500, 256, 32), (7, 1000, 64, 5), (2, 9, 256, 3), (86, 300, 104, 0),
    (1, 1172, 256, 0),          # one query tile against a big bank
    (9, 700, 16, 0), (13, 90, 8, 16),
]


@pytest.mark.parametrize("mode", [GRID, LINEAR])
@pytest.mark.parametrize("nqt,nbt,G,panel", CASES)
def test_work_list_invariants(nqt, nbt, G, panel, mode):
    segs, st = plan(nqt, nbt, G, panel, mode=mode)
    assert st["query_tiles"] == nqt and st["bank_tiles"] == nbt
    assert st["workgroups"] == min(G, nqt * nbt)
    if mode == LINEAR or st["workgroups"] % 8:
        assert st["mode"] == LINEAR
    cover = np.zeros((nqt, nbt), dtype=np.int32)
    slot_q, slot_blk, slot_last, slot_first_seen = {}, {}, {}, set()
    per_block = np.zeros(st["workgroups"], dtype=np.int64)
    last_block = -1
    syncs = defaultdict(list)
    for blk, q, b0, n, slot, first, sync in segs.tolist():
        assert blk >= last_block; last_block = blk              # segments are grouped by block
        if sync:
            syncs[blk].append(sync)
        if n == 0:
            assert sync > 0                                     # rendezvous-only entry of an idle block
            continue
        assert 0 <= q < nqt and 0 <= b0 and n > 0 and b0 + n <= nbt
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
    if st["mode"] == LINEAR:
        assert not syncs
    else:
        # every block of an XCD passes the same rendezvous 1, 2, 3, ... in order (a missing one would stall the others)
        for x in range(8):
            ref = syncs[x]
            assert ref == list(range(1, len(ref) + 1)) and len(ref) > 0
            for b in range(x, st["workgroups"], 8):
                assert syncs[b] == ref, (x, b)
    # slots per query tile stay within the two-level merge's reach at k = 256 (24 groups of 24 lists)
    assert st["max_slots_per_qtile"] <= 24 * 24
    npanels = -(-nbt // st["panel_tiles"])
    if st["mode"] == LINEAR:
        # automatic panels split evenly; forced ones within one pair per panel
        assert per_block.max() - per_block.min() <= (npanels if panel else max(1, per_block.mean() * 0.03))
    else:
        # XCD groups (blocks x, x+8, ...) carry the same load to within one (query tile x panel) unit plus rounding
        xcd = np.array([per_block[x::8].max() for x in range(8)], dtype=np.float64)
        L = st["workgroups"] // 8
        unit = st["panel_tiles"] / L + 1
        assert xcd.max() - xcd.min() <= 2 * unit + 0.02 * xcd.mean(), xcd


def test_headline_plan_numbers():
    _, lin = plan(86, 39063, 256, 0, mode=LINEAR)
    assert lin["panel_tiles"] == 128 and lin["slots"] == 340 and lin["max_slots_per_qtile"] == 4 and lin["mode"] == LINEAR
    segs, st = plan(86, 39063, 256, 0)
    assert st["mode"] == GRID and st["panel_tiles"] == 128
    assert st["max_slots_per_qtile"] <= 64
    # busiest block within 3 % of the ideal share (the 3 x 10 remainder rounds leave two workgroups idle)
    per_block = np.zeros(256, dtype=np.int64)
    np.add.at(per_block, segs[:, 0], segs[:, 3])
    assert per_block.max() <= 1.03 * 86 * 39063 / 256


def test_grid_rounds_share_bank_ranges_and_query_tiles_inside_an_xcd():
    """The point of the grid: at the same step of their lists, the workgroups of one XCD (blocks x, x+8, ...) work on at
    most 4 distinct query tiles, and those on different query tiles walk the very same bank ranges."""
    nqt, nbt = 86, 2048
    segs, st = plan(nqt, nbt, 256, 0)
    assert st["mode"] == GRID
    P = st["panel_tiles"]
    by_block = defaultdict(list)
    for blk, q, b0, n, slot, first, sync in segs.tolist():
        by_block[blk].append((q, b0, n, sync))
    for x in range(8):
        blocks = [b for b in range(x, 256, 8)]
        # a round = the segments that follow the same rendezvous
        for p0 in range(0, nbt, P):
            rounds = defaultdict(list)
            for b in blocks:
                for s in by_block[b]:
                    if s[2] > 0 and p0 <= s[1] < p0 + P:
                        rounds[s[3]].append(s[:3])
            for r, members in rounds.items():
                qs = {m[0] for m in members}
                assert len(qs) <= 4
                ranges = defaultdict(set)
                for q, b0, n in members:
                    ranges[(b0, n)].add(q)
                if len(qs) > 1:
                    # every bank range of the round is walked by all of the round's query tiles
                    assert all(v == qs for v in ranges.values()), (x, p0, r)
                # the ranges tile the panel exactly once per query tile
                for q in qs:
                    cov = sorted((b0, n) for (qq, b0, n) in members if qq == q)
                    assert cov[0][0] == p0 and cov[-1][0] + cov[-1][1] == min(p0 + P, nbt)
                    assert all(cov[i][0] + cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1))
