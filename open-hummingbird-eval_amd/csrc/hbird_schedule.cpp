// Host side of the kNN kernels' work list: which workgroup computes which (query tile, bank tile) pairs, in which order, into which
// partial-list slot, cut into which phases -- and the C entry points that expose it without a GPU (hb_schedule_plan*, include/hbird_hip.h).
// Plain C++ (no HIP): compiled by hipcc into libhbird_hip.so and, host-only with -fsanitize=address,undefined -DHB_PLAN_STANDALONE, into
// lib/build/libhbird_plan_asan.so for the sanitizer leg of the CPU suite (tests/test_sanitizers_cpu.py).
#include "hbird_schedule.h"
#include <algorithm>
#include <cmath>
#include <map>

#ifdef HB_PLAN_STANDALONE
static thread_local std::string g_plan_err;
int hb_fail(const std::string& msg) { g_plan_err = msg; return -1; }
extern "C" const char* hb_last_error(void) { return g_plan_err.c_str(); }
#endif

// ---- host-side work list ------------------------------------------------------------------------------
// Pairs (query tile q, bank tile b) are processed panel by panel (a panel = `panel` consecutive bank
// tiles, sized to stay resident in the 256 MiB Infinity Cache together with the queries): inside a
// panel the q-major pair list is cut into G equal contiguous ranges, one per workgroup, so at any time
// all workgroups read the same panel (each bank byte leaves HBM about once per search) while every
// workgroup keeps working on the same <= 2 query tiles for the whole search.  Bank tiles are visited in
// ascending order for every slot, which the strict `score > threshold` filter relies on for ties.
//
// Tried and dropped (round 1, 10 M x 768): dealing each XCD's 32 workgroups a grid of 4 query tiles x 8 bank ranges
// per round, started together by an XCD-wide rendezvous, so that fragments are shared through the XCD's L2.  The
// sharing works (L2 hit rate 26 % -> 68 %, fabric reads -58 %, profiles/r01/README.md) but buys nothing: the fp32
// kernel is bound by the matrix pipe (2438 vs 2413 ms) and the fp16 candidate kernel by the latency of a stage's slowest
// line, which a 68 % hit rate does not shorten, while 3.4x more partial-list slots cost more (454 vs 413 ms).
static int hb_gcd(int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; }

int hb_default_panel(int nqt, int G, size_t tile_bytes, int cq, int cb) {
    // smallest panel for which the work of a panel divides evenly: nqt * panel pairs over G workgroups, or, with
    // clusters, ceil(nqt / cq) * (panel / cb) units over G / (cq cb) clusters
    int p0;
    if (cq * cb > 1) {
        const int NC = std::max(1, G / (cq * cb)), NQG = (nqt + cq - 1) / cq;
        p0 = cb * (NC / hb_gcd(NQG, NC));
    } else p0 = G / hb_gcd(nqt, G);
    size_t budget = (size_t)96 << 20;
    int j = (int)std::max<size_t>(1, budget / (tile_bytes * (size_t)p0));
    return p0 * j;
}

// L2-sharing clusters: cq x cb workgroups of ONE XCD walk the same unit list in lockstep -- a unit is (cq query tiles)
// x (cb consecutive bank tiles), member (ia, ib) takes the pair (query tile ia, bank tile ib) -- so that at any time the
// cq members with the same ib stream the same bank tile and the cb members with the same ia the same query tile: one L2
// fill serves cq (bank) or cb (query) consumers, fabric traffic per pair drops from Q + B to Q / cb + B / cq.  The
// members hold each other within a few stages through the progress words (soft sync in the kernels); placement and
// lockstep are speed only, any schedule gives the same result.
void hb_default_cluster(int nqt, int nbt, int G, bool fp32_kernel, int* cq, int* cb) {
    // The first shape whose ragged last group idles at most 2.5 % of the pairs (fp16 kernel: 1 / 16 -- 49 query tiles run 6 % faster as
    // 4 x 2 with three idle members than as 2 x 2).  fp16 candidate kernel: 8 x 1 (eight
    // workgroups stream the same bank tiles: measured best at 10 M x 768), else 4 x 2, else 2 x 2, else none.  fp32 kernel
    // (bound by the matrix pipe, so only the cheapest sharing pays): 2 x 4, else 2 x 2 -- at 10 M x 768 fabric reads
    // 4.79 -> 1.93 TB for +0.5 % time; 4 x 2 and 8 x 1 cost 3 %.
    *cq = 1; *cb = 1;
    if ((long long)nqt * nbt < 64LL * G) return;                 // enough work to share
    static const int shapes16[3][2] = {{8, 1}, {4, 2}, {2, 2}};
    static const int shapes32[3][2] = {{2, 4}, {2, 2}, {2, 2}};
    for (const auto& sh : fp32_kernel ? shapes32 : shapes16) {
        const int q = sh[0], b = sh[1];
        if (G % (8 * q * b) != 0 || nqt < q) continue;
        const int padded = (nqt + q - 1) / q * q;
        if ((padded - nqt) * (fp32_kernel ? 40 : 16) > padded) continue;
        *cq = q; *cb = b;
        return;
    }
}

static void hb_finish_schedule(hb_schedule& out, std::vector<std::vector<hb_seg>>& per_wg, const std::vector<int>& logical_of_block,
                               const std::vector<std::vector<int>>& slots_of_qt) {
    const int G = out.G;
    // Phased searches (hb_build_schedule): every workgroup's segment list is cut at the same CLOCK values (tiles dealt), so that all
    // workgroups work in every phase and the members of a cluster stay on their common clock; a segment that straddles a cut is
    // split (same slot, the second piece continues it)
    const int n_cuts = (int)out.phase_clock.size();
    std::vector<std::vector<int>> rel(n_cuts, std::vector<int>(G, 0));   // per cut and logical workgroup: index of the first segment at or beyond it
    if (n_cuts) {
        // Uneven shares (hb_cum_shares): a worker that is dealt 1 % more pairs has its cuts 1 % later, so that every phase -- not only the
        // last one -- is as long for the fast XCDs as for the slow ones.  A worker = CS consecutive logical workgroups (a cluster: one clock).
        const int CS = std::max(1, out.cq * out.cb);
        std::vector<double> scale(G, 1.0);
        if (!out.xcd_w.empty() && G % CS == 0) {
            std::vector<double> end(G / CS, 0.0);
            double mean = 0.0;
            for (int w = 0; w < G; ++w)
                if (!per_wg[w].empty()) end[w / CS] = std::max(end[w / CS], (double)(per_wg[w].back().tile0 + per_wg[w].back().n_tiles));
            for (double e : end) mean += e / (double)end.size();
            // (from 1,000 pairs per workgroup: below, a cut moved by one tile is a bigger error than the skew it mends -- 600 k x 768 x 12,544
            // queries, 448 pairs: 78.3 -> 78.6 ms with moved cuts; 2 M x 384, 1,551 pairs: 135.4 -> 134.9; 10 M x 768 k = 90: 2265 -> 2255)
            if (mean >= 1000.0) {
                for (int w = 0; w < G; ++w) scale[w] = end[w / CS] > 0.0 ? end[w / CS] / mean : 1.0;
                out.cuts_scaled = true;
            }
        }
        std::vector<int> cuts(n_cuts);
        for (int w = 0; w < G; ++w) {
            for (int p = 0; p < n_cuts; ++p) {
                cuts[p] = std::max(1, (int)std::lround((double)out.phase_clock[p] * scale[w]));
                if (p && cuts[p] <= cuts[p - 1]) cuts[p] = cuts[p - 1] + 1;
            }
            std::vector<hb_seg> v;
            v.reserve(per_wg[w].size() + n_cuts);
            for (hb_seg sg : per_wg[w]) {
                // (the cuts ascend: only those inside the segment are visited)
                for (auto ct = std::upper_bound(cuts.begin(), cuts.end(), sg.tile0); ct != cuts.end() && *ct < sg.tile0 + sg.n_tiles; ++ct) {
                    const int t = *ct;
                    {
                        hb_seg head = sg;
                        head.n_tiles = t - sg.tile0;
                        v.push_back(head);
                        sg.b_tile0 += sg.stride * head.n_tiles; sg.tile0 = t; sg.n_tiles -= head.n_tiles; sg.first = 0;
                    }
                }
                v.push_back(sg);
            }
            per_wg[w].swap(v);
            for (int p = 0, i = 0; p < n_cuts; ++p) {      // (cuts and clocks both ascend: one pass)
                while (i < (int)per_wg[w].size() && per_wg[w][i].tile0 < cuts[p]) ++i;
                rel[p][w] = i;
            }
        }
    }
    std::vector<std::pair<int, int>> ord_of((size_t)std::max(1, out.n_slots), std::make_pair(0, 0));   // slot -> (ordinal among its query tile's slots, their number)
    for (const auto& sl : slots_of_qt)
        for (size_t i = 0; i < sl.size(); ++i) ord_of[sl[i]] = {(int)i, (int)sl.size()};
    for (auto& v : per_wg)
        for (size_t i = 0; i < v.size(); ++i) {
            v[i].next_tile0 = i + 1 < v.size() ? v[i + 1].tile0 : 0x7FFFFFFF;
            v[i].ord = ord_of[v[i].slot].first; v[i].nsl = ord_of[v[i].slot].second;
        }
    out.wg_off.assign(G + 1, 0);
    out.phase_bounds.assign((size_t)n_cuts * G, 0);
    for (int b = 0; b < G; ++b) {
        const int w = logical_of_block[b];
        out.wg_off[b + 1] = out.wg_off[b] + (int)per_wg[w].size();
        out.segs.insert(out.segs.end(), per_wg[w].begin(), per_wg[w].end());
        for (int p = 0; p < n_cuts; ++p) out.phase_bounds[(size_t)p * G + b] = out.wg_off[b] + rel[p][w];
    }
    out.qt_off.assign(out.nqt + 1, 0);
    for (int q = 0; q < out.nqt; ++q) {
        out.qt_off[q + 1] = out.qt_off[q] + (int)slots_of_qt[q].size();
        out.qt_slots.insert(out.qt_slots.end(), slots_of_qt[q].begin(), slots_of_qt[q].end());
        out.max_slots_per_qt = std::max(out.max_slots_per_qt, (int)slots_of_qt[q].size());
    }
}

// Uneven shares (hb_index_set_xcd_weights).  The XCDs of one MI355X do not run at one speed: with equal work the workgroups of the odd XCDs
// finish 1-2 % after those of the even ones (profiles/r05/xcd_speed_stamps_headline.txt: +1.3 % / -0.6 % around the median), and a launch
// lasts as long as its slowest workgroup.  `cum[i]` = the share of a panel's units that the first i workers (logical workgroups, or clusters)
// take together; the boundary of worker i in panel p is floor(W cum[i] + phi_p) with a per-panel dither phi_p (the golden-ratio sequence),
// so that the rounding of one panel does not repeat in every panel (42.5 units per panel must not become 42 three hundred times).
static std::vector<double> hb_cum_shares(int workers, int per_xcd, const std::vector<double>& xcd_w) {
    std::vector<double> cum(workers + 1, 0.0);
    for (int i = 0; i < workers; ++i) cum[i + 1] = cum[i] + xcd_w[std::min(7, i / std::max(1, per_xcd))];
    for (int i = 0; i <= workers; ++i) cum[i] /= cum[workers];
    return cum;
}
static long long hb_share_bound(long long W, const std::vector<double>& cum, int i, int workers, int panel_index) {
    if (i <= 0) return 0;
    if (i >= workers) return W;
    const double phi = std::fmod(0.5 + 0.6180339887498949 * panel_index, 1.0);
    return std::min<long long>(W, std::max<long long>(0, (long long)std::floor((double)W * cum[i] + phi)));
}

static void hb_build_clustered(int nqt, int nbt, int G, int panel, int cq, int cb, hb_schedule& out, bool xcd_share) {
    const int CS = cq * cb, NC = G / CS, NQG = (nqt + cq - 1) / cq;
    out.cq = cq; out.cb = cb; out.n_clusters = NC; out.xcd_share = xcd_share;
    std::vector<std::vector<hb_seg>> per_wg(G);   // logical workgroup = cluster * CS + member
    std::vector<int> slot_of((size_t)G * nqt, -1);   // (logical wg, q_tile) -> slot
    std::vector<std::vector<int>> slots_of_qt(nqt);
    std::vector<int> clock(NC, 0);                // cluster clock: tiles each member has been dealt (idle ones included)
    // units [qg][j0, j0 + cnt) of the panel at bank tile b0 (pp tiles) -> the members of cluster c
    auto deal = [&](int c, int b0, int pp, int qg, int j0, int cnt) {
        for (int m = 0; m < CS; ++m) {
            const int ia = m / cb, ib = m % cb, q = qg * cq + ia, w = c * CS + m;
            int n = cnt;
            if ((j0 + cnt - 1) * cb + ib >= pp) --n;      // the partial last group has no tile for this member
            if (q >= nqt || n <= 0) continue;             // idle for these units (its clock still advances)
            hb_seg sg;
            sg.q_tile = q; sg.b_tile0 = b0 + j0 * cb + ib; sg.n_tiles = n; sg.stride = cb; sg.tile0 = clock[c]; sg.next_tile0 = 0;
            int& known = slot_of[(size_t)w * nqt + q];
            if (known < 0) {
                sg.slot = out.n_slots++; sg.first = 1;
                known = sg.slot;
                slots_of_qt[q].push_back(sg.slot);
            } else { sg.slot = known; sg.first = 0; }
            per_wg[w].push_back(sg);
        }
        clock[c] += cnt;
    };
    const int per_xcd_c = NC / 8;
    int rot = 0;
    // uneven shares: per cluster (plain clustered list) or per XCD range (XCD-level query sharing)
    const std::vector<double> cum = (!out.xcd_w.empty() && !xcd_share && NC % 8 == 0) ? hb_cum_shares(NC, NC / 8, out.xcd_w) : std::vector<double>();
    const std::vector<double> cumx = (!out.xcd_w.empty() && xcd_share) ? hb_cum_shares(8, 1, out.xcd_w) : std::vector<double>();
    for (int b0 = 0; b0 < nbt; b0 += panel) {
        const int pp = std::min(panel, nbt - b0);
        const int UB = (pp + cb - 1) / cb;        // bank groups of the panel (the last one may be partial)
        const long long U = (long long)NQG * UB;
        if (xcd_share) {
            // XCD-level sharing of the QUERY tiles: the q-major unit list is cut into eight ranges, one per XCD, and every run of
            // one query group inside a range is split over ALL clusters of that XCD (contiguous bank sub-ranges), so that at any
            // time the XCD's workgroups re-read the same cq query tiles -- which then stay in its L2 (cq x 384 KiB of fp16 at
            // D = 768 beside the bank streams) instead of being re-streamed through the fabric for every pair; the clusters among
            // themselves need no sync for that.  The remainders of a run rotate over the clusters (balance within a tile or two).
            for (int x = 0; x < 8; ++x) {
                long long e = (U * x) / 8;
                long long e1 = (U * (x + 1)) / 8;
                if (!cumx.empty()) { e = hb_share_bound(U, cumx, x, 8, b0 / panel); e1 = hb_share_bound(U, cumx, x + 1, 8, b0 / panel); }
                while (e < e1) {
                    const int qg = (int)(e / UB), j0 = (int)(e % UB);
                    const int L = (int)std::min<long long>(UB - j0, e1 - e);
                    for (int i = 0; i < per_xcd_c; ++i) {
                        const int a0 = (int)((long long)L * i / per_xcd_c), a1 = (int)((long long)L * (i + 1) / per_xcd_c);
                        if (a1 > a0) deal(x * per_xcd_c + (i + rot) % per_xcd_c, b0, pp, qg, j0 + a0, a1 - a0);
                    }
                    ++rot;
                    e += L;
                }
            }
            continue;
        }
        for (int c = 0; c < NC; ++c) {
            long long e0 = (U * c) / NC, e1 = (U * (c + 1)) / NC;
            if (!cum.empty()) { e0 = hb_share_bound(U, cum, c, NC, b0 / panel); e1 = hb_share_bound(U, cum, c + 1, NC, b0 / panel); }
            while (e0 < e1) {
                const int qg = (int)(e0 / UB), j0 = (int)(e0 % UB);
                const int cnt = (int)std::min<long long>(UB - j0, e1 - e0);
                deal(c, b0, pp, qg, j0, cnt);
                e0 += cnt;
            }
        }
    }
    // placement (speed only): blocks b, b + 8, ... share an XCD; every XCD gets NC / 8 consecutive clusters, whole
    std::vector<int> logical_of_block(G);
    out.wg_member.assign(G, -1);
    const int per_xcd = NC / 8;
    for (int c = 0; c < NC; ++c)
        for (int m = 0; m < CS; ++m) {
            const int block = c / per_xcd + 8 * ((c % per_xcd) * CS + m);
            logical_of_block[block] = c * CS + m;
            out.wg_member[block] = c * HB_CLUSTER_LINE + m;
        }
    hb_finish_schedule(out, per_wg, logical_of_block, slots_of_qt);
}

void hb_build_schedule(int nqt, int nbt, int G, int panel, hb_schedule& out, int cq, int cb, bool phased, bool xcd_share, const double* xcd_w) {
    out = hb_schedule();
    out.nqt = nqt; out.nbt = nbt; out.panel = panel; out.phased = phased;
    if (xcd_w) {
        bool ok = true, equal = true;
        for (int x = 0; x < 8; ++x) { ok = ok && xcd_w[x] > 0.25 && xcd_w[x] < 4.0; equal = equal && xcd_w[x] == xcd_w[0]; }
        if (ok && !equal) out.xcd_w.assign(xcd_w, xcd_w + 8);      // (equal shares keep the exact integer split below)
    }
    const long long total_pairs = (long long)nqt * nbt;
    if (total_pairs < G) G = (int)std::max<long long>(1, total_pairs);
    out.G = G;
    if (phased) {
        // Cut clocks (tiles dealt per workgroup): 1, 3, 6, 10, 16, 25, ... -- phases that grow by half, twofold beyond 128 tiles;
        // the last phase keeps at least half of the work.  With the floor fixed during a phase, a phase that multiplies the rows seen
        // by g appends about k (g - 1) candidates per query, k (g - 1) / ln g per e-fold: g = 2 is 1.44 x the continuous
        // bound k ln(N / n0), g = 1.5 1.23 x.  Measured (kernel ms, 2,074,072 x 384 fp16): first cut at 1 / 2 / 4 / 8 tiles
        // 23.6 / 24.0 / 24.0 / 24.4, growth 1.5 / 2 / 3 flat within 0.3; at 50,176 x 384 the first cut is what matters (fp32 k = 90:
        // 7.85 with cuts from 2 tiles, 6.2 from 1).
        // Short searches (at most 64 tiles per workgroup) grow threefold: every launch costs about 40 us beyond its tiles (cfg-1, fp32: 132 /
        // 224 / 321 / 410 / 579 / 1983 us for 1 / 2 / 3 / 4 / 6 / 21.5 tiles per workgroup), four launches instead of six there: 3.78 -> 3.72 ms.
        const long long per_wg_tiles = total_pairs / G;
        long long t = 0, step = 1;
        while (out.phase_clock.size() < HB_PHASE_CUTS) {
            t += step;
            if (t * 2 > per_wg_tiles) break;
            out.phase_clock.push_back((int)t);
            step = std::max<long long>(step + 1, per_wg_tiles <= 64 ? step * 3 : step < 128 ? step * 3 / 2 : step * 2);
        }
    }
    if (cq < 1 || cb < 1 || cq * cb > HB_CLUSTER_MAX || G % (8 * cq * cb) != 0) { cq = 1; cb = 1; }
    if (cq * cb > 1) { hb_build_clustered(nqt, nbt, G, panel, cq, cb, out, xcd_share); return; }
    std::vector<std::vector<hb_seg>> per_wg(G);
    std::vector<int> slot_of((size_t)G * nqt, -1);   // (wg, q_tile) -> slot
    std::vector<std::vector<int>> slots_of_qt(nqt);
    std::vector<int> clock(G, 0);
    const std::vector<double> cum = (!out.xcd_w.empty() && G % 8 == 0) ? hb_cum_shares(G, G / 8, out.xcd_w) : std::vector<double>();
    for (int b0 = 0; b0 < nbt; b0 += panel) {
        const int pp = std::min(panel, nbt - b0);
        const long long W = (long long)nqt * pp;
        for (int w = 0; w < G; ++w) {
            long long e0 = (W * w) / G, e1 = (W * (w + 1)) / G;
            if (!cum.empty()) { e0 = hb_share_bound(W, cum, w, G, b0 / panel); e1 = hb_share_bound(W, cum, w + 1, G, b0 / panel); }
            while (e0 < e1) {
                const int q = (int)(e0 / pp), b = (int)(e0 % pp);
                const int cnt = (int)std::min<long long>(pp - b, e1 - e0);
                int& known = slot_of[(size_t)w * nqt + q];
                hb_seg sg;
                sg.q_tile = q; sg.b_tile0 = b0 + b; sg.n_tiles = cnt; sg.stride = 1; sg.tile0 = clock[w]; sg.next_tile0 = 0;
                clock[w] += cnt;
                if (known < 0) {
                    sg.slot = out.n_slots++; sg.first = 1;
                    known = sg.slot;
                    slots_of_qt[q].push_back(sg.slot);
                } else { sg.slot = known; sg.first = 0; }
                // coalesce with the previous segment when it continues the same slot contiguously
                if (!per_wg[w].empty()) {
                    hb_seg& pv = per_wg[w].back();
                    if (pv.slot == sg.slot && pv.b_tile0 + pv.n_tiles == sg.b_tile0) { pv.n_tiles += cnt; e0 += cnt; continue; }
                }
                per_wg[w].push_back(sg);
                e0 += cnt;
            }
        }
    }
    // XCD-aware placement (speed only, never correctness): hardware deals workgroups round-robin over the 8 XCDs
    // (block b runs on XCD b % 8, checked with tools/ubench/xcc_map.hip; blocks b and b+8 share an L2), so logical
    // ranges v = 0..G-1 -- neighbours share a query tile -- are laid out so that each XCD gets a contiguous run of
    // them: block b runs logical range (b % 8) * (G / 8) + b / 8.
    std::vector<int> logical_of_block(G);
    for (int b = 0; b < G; ++b) {
        if (G % 8 == 0) logical_of_block[b] = (b % 8) * (G / 8) + b / 8;
        else {
            const int q = G / 8, r = G % 8, x = b % 8;   // bijective variant for G not a multiple of 8
            logical_of_block[b] = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
        }
    }
    out.wg_member.assign(G, -1);
    hb_finish_schedule(out, per_wg, logical_of_block, slots_of_qt);
}


// Host-only: build the kNN work list for a (query tiles x bank tiles) grid without touching a GPU, for inspection
// and tests.  segs_out receives up to max_segs rows of {block, q_tile, b_tile0, n_tiles, slot, first}.
static int schedule_plan(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, bool phased,
                         int* segs_out, int64_t max_segs, int64_t stats[8], hb_schedule& sc, bool xcd_share = false, const double* xcd_w = nullptr) {
    if (!stats) return hb_fail("hb_schedule_plan: stats is NULL");
    if (nqt <= 0 || nbt <= 0 || workgroups <= 0 || d <= 0) return hb_fail("hb_schedule_plan: bad arguments");
    const int dp = (d + HB_KC - 1) / HB_KC * HB_KC;
    const int G = (int)std::min<long long>(workgroups, (long long)nqt * nbt);
    int cq = cluster_q, cb = cluster_b;
    if (cq < 0 || cb < 0) hb_default_cluster(nqt, nbt, G, cq == -2, &cq, &cb);  // negative: the automatic shape (-1 fp16, -2 fp32 kernel)
    if (cq < 1 || cb < 1 || (long long)nqt * nbt < workgroups || cq * cb > HB_CLUSTER_MAX || G % (8 * cq * cb) != 0) { cq = 1; cb = 1; }
    const int panel = panel_tiles > 0 ? panel_tiles : hb_default_panel(nqt, G, (size_t)HB_BT * dp * 4, cq, cb);
    hb_build_schedule(nqt, nbt, workgroups, panel, sc, cq, cb, phased, xcd_share && cq * cb > 1, xcd_w);
    stats[0] = sc.G; stats[1] = (int64_t)sc.segs.size(); stats[2] = sc.n_slots; stats[3] = sc.panel;
    stats[4] = sc.max_slots_per_qt; stats[5] = sc.nqt; stats[6] = sc.nbt; stats[7] = sc.cq * 16 + sc.cb;
    if (segs_out) {
        int64_t n = 0;
        for (int b = 0; b < sc.G; ++b)
            for (int i = sc.wg_off[b]; i < sc.wg_off[b + 1] && n < max_segs; ++i, ++n) {
                const hb_seg& g = sc.segs[i];
                int* o = segs_out + n * 10;
                o[0] = b; o[1] = g.q_tile; o[2] = g.b_tile0; o[3] = g.n_tiles; o[4] = g.slot; o[5] = g.first;
                o[6] = g.stride; o[7] = g.tile0; o[8] = g.next_tile0; o[9] = sc.wg_member[b];
            }
    }
    return 0;
}

extern "C" int hb_schedule_plan(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b,
                                int* segs_out, int64_t max_segs, int64_t stats[8]) {
    hb_schedule sc;
    return schedule_plan(nqt, nbt, workgroups, panel_tiles, d, cluster_q, cluster_b, false, segs_out, max_segs, stats, sc);
}

// The same for a PHASED search (pools: k > 32 and the fp16 candidate pass): the work list with its segments cut at the phase
// clocks, the clocks, and per cut and block the position (within the block's own segments) of the first segment of the next phase.
extern "C" int hb_schedule_plan_phased(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b,
                                       int* segs_out, int64_t max_segs, int64_t stats[8], int* clocks_out, int max_cuts, int* n_cuts,
                                       int* bounds_out) {
    if (!n_cuts) return hb_fail("hb_schedule_plan_phased: n_cuts is NULL");
    hb_schedule sc;
    if (schedule_plan(nqt, nbt, workgroups, panel_tiles, d, cluster_q, cluster_b, true, segs_out, max_segs, stats, sc)) return -1;
    *n_cuts = (int)sc.phase_clock.size();
    for (int p = 0; p < *n_cuts && p < max_cuts; ++p) {
        if (clocks_out) clocks_out[p] = sc.phase_clock[p];
        if (bounds_out)
            for (int b = 0; b < sc.G; ++b) bounds_out[(size_t)p * sc.G + b] = sc.phase_bounds[(size_t)p * sc.G + b] - sc.wg_off[b];
    }
    return 0;
}

// hb_schedule_plan with per-XCD work shares (hb_index_set_xcd_weights): cluster_q / cluster_b as above, shared bit 0 = XCD-level query
// sharing, bit 1 = a phased list (its cuts follow the shares)
extern "C" int hb_schedule_plan_weighted(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, int shared,
                                         const double* xcd_w8, int* segs_out, int64_t max_segs, int64_t stats[8]) {
    hb_schedule sc;
    return schedule_plan(nqt, nbt, workgroups, panel_tiles, d, cluster_q, cluster_b, (shared & 2) != 0, segs_out, max_segs, stats, sc, (shared & 1) != 0, xcd_w8);
}

extern "C" int hb_schedule_plan_shared(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, int phased,
                                       int* segs_out, int64_t max_segs, int64_t stats[8]) {
    hb_schedule sc;
    return schedule_plan(nqt, nbt, workgroups, panel_tiles, d, cluster_q, cluster_b, phased != 0, segs_out, max_segs, stats, sc, true);
}

