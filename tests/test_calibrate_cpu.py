"""The decisions the kNN launcher takes from its workgroups' time stamps -- per-XCD work shares, their guard, the cluster decision -- as plain
host code (csrc/hbird_calibrate.cpp), fed here with the stamp sets of IMAGINED launches on a chip whose XCDs run at known speeds: no GPU.
(Round 5's calibration bug -- shares walking to their clamps on stale memory, a silent slowdown the parity tests cannot see -- lived in
exactly this logic.)"""
import ctypes

import numpy as np
import pytest

from hbird_mi import _lib

G = 256


class Chip:
    """Eight XCDs of given relative speeds; a launch of `work` ticks per unit share; block b runs on XCD (b + first) % 8."""

    def __init__(self, speeds, first=0, seed=0, jitter=0.0):
        self.speeds = np.asarray(speeds, dtype=np.float64)
        self.first, self.rng, self.jitter = first, np.random.default_rng(seed), jitter

    def launch(self, shares, work=200_000_000.0, cluster_cost=0.0):
        """-> stamps [G][2][4] uint32 of a launch in which group g did shares[g] of the work on its XCD (+ a relative cost of the clusters)."""
        st = np.zeros((G, 2, 4), dtype=np.uint32)
        t0 = 1_000_000
        for b in range(G):
            g, x = b % 8, (b + self.first) % 8
            dur = work * shares[g] / self.speeds[x] * (1.0 + cluster_cost) * (1.0 + self.jitter * self.rng.standard_normal())
            start = t0 + int(self.rng.integers(0, 40))
            end = start + int(dur)
            cyc0 = 5_000_000_000 + 24 * start
            st[b, 0] = (start & 0xFFFFFFFF, x, cyc0 & 0xFFFFFFFF, cyc0 >> 32)
            cyc1 = cyc0 + 24 * (end - start)                     # 2.4 GHz: 24 cycles per 10 ns tick
            st[b, 1] = (end & 0xFFFFFFFF, x, cyc1 & 0xFFFFFFFF, cyc1 >> 32)
        return st


class Cal:
    def __init__(self, fp16=False):
        self.L = _lib.lib()
        self.h = ctypes.c_void_p(self.L.hb_calibration_new(int(fp16)))

    def __del__(self):
        self.L.hb_calibration_free(self.h)

    def state(self):
        w = (ctypes.c_double * 8)(); o = (ctypes.c_int64 * 12)()
        assert self.L.hb_calibration_state(self.h, w, o) == 0
        keys = ["rounds", "locked", "reverts", "samples", "rejected", "map_moves", "cl_state", "cl_choice", "cur_n", "xcd_of_block0", "n_on", "n_off"]
        return np.array(list(w)), dict(zip(keys, [int(v) for v in o]))

    def feed(self, stamps, shares, key=(86, 39063, G, 1, 30, 17), auto_cluster=0, frac=1.0):
        st = np.ascontiguousarray(stamps, dtype=np.uint32)
        return self.L.hb_calibration_feed(self.h, st.ctypes.data_as(ctypes.c_void_p), G, (ctypes.c_double * 8)(*shares), (ctypes.c_int * 6)(*key),
                                          int(auto_cluster), float(frac))


def span(st):
    return float(st[:, 1, 0].astype(np.int64).max() - st[:, 0, 0].astype(np.int64).min())


def test_shares_follow_the_xcds_speeds_and_the_launch_gets_shorter():
    """Odd XCDs 2 % slower: after the first stamp set the shares give them less work, the groups end together, the launch is shorter; later
    sets of the same speeds change nothing (and ask for no new work list)."""
    chip = Chip([1.0, 0.98, 1.0, 0.98, 1.0, 0.98, 1.0, 0.98])
    cal = Cal()
    w, _ = cal.state()
    assert np.allclose(w, 1.0)
    s0 = chip.launch(w)
    assert cal.feed(s0, w) & 1                                   # rebuild the list with the new shares
    w1, st = cal.state()
    assert st["rounds"] == 1 and st["rejected"] == 0 and abs(w1.mean() - 1.0) < 1e-12
    assert (w1[1::2] < 0.995).all() and (w1[0::2] > 1.005).all()
    s1 = chip.launch(w1)
    assert span(s1) < span(s0) * 0.992                           # 2 % imbalance -> about 1 % of the launch
    for _ in range(4):
        flags = cal.feed(chip.launch(cal.state()[0]), cal.state()[0])
    assert not flags & 1 and cal.state()[1]["locked"] == 0
    d = [np.median((s1[g::8, 1, 0].astype(np.int64) - s1[g::8, 0, 0].astype(np.int64))) for g in range(8)]
    assert max(d) / min(d) < 1.001                               # the groups end together


def test_garbage_stamps_are_rejected_and_change_nothing():
    """What round 5 calibrated on when a kernel did not stamp: zeros (the region is zeroed before a launch), blocks of one group on
    different XCDs, two groups on one XCD, a duration far from the others'.  Every such set is thrown away; the shares stay."""
    chip = Chip([1.0] * 8)
    cal = Cal()
    good = chip.launch([1.0] * 8)
    bad = []
    z = good.copy(); z[17] = 0; bad.append(z)                                   # a block that never stamped
    m = good.copy(); m[9, :, 1] = 5; bad.append(m)                              # block 9 (group 1) on XCD 5
    t = good.copy(); t[1::8, :, 1] = 0; bad.append(t)                           # groups 0 and 1 both on XCD 0
    o = good.copy(); o[3::8, 1, 0] = o[3::8, 0, 0] + 3 * (good[3, 1, 0] - good[3, 0, 0]); bad.append(o)     # a group three times as long
    e = good.copy(); e[40, 1, 0] = e[40, 0, 0]; bad.append(e)                   # start == end
    for st in bad:
        assert cal.feed(st, [1.0] * 8) == 8
    w, s = cal.state()
    assert np.allclose(w, 1.0) and s["rejected"] == len(bad) and s["samples"] == 0 and s["rounds"] == 0


def test_the_guard_drops_shares_that_measure_slower():
    """A box on which the shares do NOT help (the durations the calibration sees do not repeat: the slow XCDs of one launch are the fast ones
    of the next): the calibrated set's launches are longer than the equal-shares launch, the guard brings equal shares back and locks."""
    cal = Cal()
    a, b = Chip([1.0, 0.97] * 4), Chip([0.97, 1.0] * 4)
    w0, _ = cal.state()
    cal.feed(a.launch(w0), w0)                                   # calibrates for chip a ...
    w1, _ = cal.state()
    assert (w1[1::2] < 0.99).all()
    flags = 0
    for _ in range(3):                                           # ... but the chip now behaves like b: these shares make it worse
        flags = cal.feed(b.launch(w1), w1)
        if cal.state()[1]["locked"]:
            break
    w2, st = cal.state()
    assert st["locked"] == 1 and st["reverts"] == 1 and flags & 1 and np.allclose(w2, 1.0)
    assert cal.feed(b.launch(w2), w2) & 1 == 0 and np.allclose(cal.state()[0], 1.0)      # locked: no further calibration


def test_shares_belong_to_the_physical_xcds_whatever_block_0_lands_on():
    """Block 0 on XCD 3: the groups' shares are the physical shares through the observed map; a map that keeps moving ends the calibration
    with equal shares."""
    speeds = [1.0, 0.97, 1.0, 1.0, 1.02, 1.0, 0.99, 1.0]
    cal = Cal()
    chip = Chip(speeds, first=3)
    w, _ = cal.state()
    cal.feed(chip.launch(w), w)                                  # the map moves once: group g ran on XCD (g + 3) % 8
    w1, st = cal.state()
    assert st["map_moves"] == 1 and st["xcd_of_block0"] == 3
    assert w1[(1 - 3) % 8] == w1.min() and w1[(4 - 3) % 8] == w1.max()      # the group that runs on the slow XCD 1 / the fast XCD 4
    s1 = chip.launch(w1)
    assert span(s1) < span(chip.launch(w)) * 0.99
    cal2 = Cal()
    for first in (1, 2, 3):
        c = Chip(speeds, first=first)
        g = cal2.state()[0]
        cal2.feed(c.launch(g), g)
    w2, st2 = cal2.state()
    assert st2["locked"] == 2 and np.allclose(w2, 1.0)


@pytest.mark.parametrize("cluster_cost,kept", [(+0.006, 0), (-0.025, 1)])
def test_the_fp32_clusters_stay_only_where_they_measure_faster(cluster_cost, kept):
    """Two calibrated launches with clusters, two without, the faster form stays: a box at full clock pays 0.6 % for them (dropped), the
    power-limited box of round 5 gained 2.5 % (kept).  Launches of another shape in between do not count."""
    chip = Chip([1.0, 0.985] * 4, jitter=0.0002, seed=3)
    cal = Cal()
    on, off = (86, 39063, G, 1, 30, 2 * 16 + 4), (86, 39063, G, 1, 30, 17)
    trail = []
    for i in range(8):
        w, st = cal.state()
        clustered = st["cl_state"] == 0 or (st["cl_state"] == 2 and st["cl_choice"] == 1)
        if i == 3:                                               # a search of another shape (fewer query tiles) in between: ignored by the trial
            cal.feed(chip.launch(w, work=1e8, cluster_cost=cluster_cost if clustered else 0.0), w, key=(40, 39063, G, 1, 30, on[5] if clustered else 17), auto_cluster=1)
            continue
        cal.feed(chip.launch(w, cluster_cost=cluster_cost if clustered else 0.0), w, key=on if clustered else off, auto_cluster=1)
        trail.append(clustered)
    _, st = cal.state()
    assert st["cl_state"] == 2 and st["cl_choice"] == kept, st
    assert trail[:3] == [True, True, True] and trail[3:5] == [False, False] and trail[-1] == bool(kept), trail
    assert st["n_on"] == 2 and st["n_off"] == 2


def test_the_fp16_family_is_damped_and_its_guard_is_wider():
    """The fp16 candidate kernel's launches scatter by 0.5 %: its shares move by half steps after the second round, and a share set is only
    dropped after three launches that are 0.8 % slower -- a 0.4 % difference (what sent one box of round 6 back to equal shares) is not enough."""
    chip = Chip([1.0, 0.96, 1.0, 0.97, 1.0, 0.95, 1.0, 0.98])
    cal = Cal(fp16=True)
    key = (86, 39063, G, 18, 64, 8 * 16 + 1)
    for _ in range(3):
        w, _ = cal.state()
        cal.feed(chip.launch(w), w, key=key)
    w, st = cal.state()
    assert st["rounds"] == 3 and st["locked"] == 0 and w[5] == w.min()
    for _ in range(4):                                           # launches 0.4 % longer than the best seen: tolerated
        cal.feed(chip.launch(w, work=200_000_000.0 * 1.004), w, key=key)
    assert cal.state()[1]["locked"] == 0 and cal.state()[1]["reverts"] == 0


def _adapt(f1, f2, nq=21904):
    n = len(f1)
    how = (ctypes.c_int * n)()
    assert _lib.lib().hb_f16_adapt_replay(n, (ctypes.c_double * n)(*f1), (ctypes.c_double * n)(*f2), nq, how) == 0
    return list(how)


def test_use_fp16_adaptive_use_follows_what_the_certificates_say():
    """Mode 2 of use_fp16 (what the plugin's use_fp16=True selects): the chain while certificates pass; one k' = 256 pass for all queries when
    most first certificates fail but the wide pass settles them; the fp32 kernel right away when most queries end there anyway -- with a probe
    of the whole chain every 16th search, through which a bank (or a query stream) that becomes certifiable again is noticed."""
    CHAIN, WIDE, FP32 = 0, 1, 2
    assert _adapt([0.0] * 40, [0.0] * 40) == [CHAIN] * 40                        # N(0,1)-like banks: always the chain
    assert _adapt([0.05] * 40, [0.0] * 40) == [CHAIN] * 40                       # 5 % failing, settled by the second pass: still the chain
    how = _adapt([1.0] * 40, [0.0] * 40)                                         # all fail the first certificate, the wide pass settles them
    assert how[0] == CHAIN and how[1] == CHAIN and set(how[2:15]) == {WIDE} and how[15] == CHAIN and set(how[16:31]) == {WIDE}
    how = _adapt([1.0] * 40, [1.0] * 40)                                         # nothing can be certified: the fp32 kernel, probes at 15 and 31
    assert how[:2] == [CHAIN, CHAIN] and set(how[2:15]) == {FP32} and how[15] == CHAIN and set(how[16:31]) == {FP32} and how[31] == CHAIN
    # a stream that becomes certifiable at search 20: the probe at 31 sees it, one more chain search halves the averages below 1/2
    how = _adapt([1.0] * 20 + [0.0] * 30, [1.0] * 20 + [0.0] * 30)
    assert set(how[16:31]) == {FP32} and how[31] == CHAIN and set(how[33:]) == {CHAIN}
    # ... and one that stops being certifiable at search 10 is noticed at once (the chain observes every search)
    how = _adapt([0.0] * 10 + [1.0] * 20, [0.0] * 10 + [1.0] * 20)
    assert how[:11] == [CHAIN] * 11 and FP32 in how[11:14] and set(how[14:15]) == {FP32}
