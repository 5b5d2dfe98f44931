"""Invariants of the host-built kNN work list (hb_build_schedule), checked on the CPU through hb_schedule_plan:
every (query tile, bank tile) pair exactly once; every slot belongs to one block and one query tile and sees its bank
tiles in ascending order (the kernel's strict `score > threshold` tie rule relies on it); `first` marks exactly the
first segment of a slot; work is balanced over the blocks."""
import ctypes

import numpy as np
import pytest

from hbird_mi import _lib


def plan(nqt, nbt, G, panel, d=768):
    stats = (ctypes.c_int64 * 8)()
    _lib.check(_lib.lib().hb_schedule_plan(nqt, nbt, G, panel, d, None, 0, stats))
    nseg = stats[1]
    buf = np.zeros((nseg, 6), dtype=np.int32)
    _lib.check(_lib.lib().hb_schedule_plan(nqt, nbt, G, panel, d, buf.ctypes.data_as(ctypes.c_void_p), nseg, stats))
    keys = ["workgroups", "segments", "slots", "panel_tiles", "max_slots_per_qtile", "query_tiles", "bank_tiles"]
    return buf, dict(zip(keys, list(stats)[:7]))


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
    for blk, q, b0, n, slot, first in segs.tolist():
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
