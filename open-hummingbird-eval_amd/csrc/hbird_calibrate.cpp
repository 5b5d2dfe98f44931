// The calibration decisions of the kNN launcher as plain host code (see hbird_calibrate.h): compiled by hipcc into libhbird_hip.so and,
// host-only under sanitizers, into lib/build/libhbird_plan_asan.so beside the planner.
#include "hbird_calibrate.h"
#include <algorithm>
#include <cmath>
#include <vector>

bool hb_stamps_summarise(const unsigned* st, int G, hb_stamp_summary& o) {
    if (G < 8 || G % 8 != 0) return false;
    std::vector<double> dur[8], every, ghz;
    const unsigned s0 = st[0];
    long long first = 0, last = 0;
    for (int b = 0; b < G; ++b) {
        const unsigned* sb = st + 8 * (size_t)b;
        const unsigned t0 = sb[0], t1 = sb[4], xcc = sb[1];
        if (t0 == 0u && t1 == 0u) return false;
        if (xcc > 7u || sb[5] != xcc) return false;
        if (b < 8) o.xcc[b] = (int)xcc;
        else if ((int)xcc != o.xcc[b & 7]) return false;
        const unsigned d = t1 - t0;                       // (mod 2^32: a launch is far shorter than 43 s)
        if (d == 0u || d > 0x7FFFFFFFu) return false;
        dur[b & 7].push_back((double)d); every.push_back((double)d);
        const long long rs = (long long)(int)(t0 - s0), re = rs + (long long)d;
        if (b == 0) { first = rs; last = re; }
        first = std::min(first, rs); last = std::max(last, re);
        const unsigned long long c0 = ((unsigned long long)sb[3] << 32) | sb[2], c1 = ((unsigned long long)sb[7] << 32) | sb[6];
        if (c1 > c0 && d >= 100u) ghz.push_back((double)(c1 - c0) / (double)d * 0.1);       // cycles per 10 ns tick -> GHz
    }
    for (int x = 0, seen = 0; x < 8; ++x) { if (seen & (1 << o.xcc[x])) return false; seen |= 1 << o.xcc[x]; }
    std::nth_element(every.begin(), every.begin() + every.size() / 2, every.end());
    const double m_all = every[every.size() / 2];
    o.all = 0.0;
    for (int x = 0; x < 8; ++x) {
        if (dur[x].empty()) return false;
        std::nth_element(dur[x].begin(), dur[x].begin() + dur[x].size() / 2, dur[x].end());
        o.med[x] = dur[x][dur[x].size() / 2];
        if (!(o.med[x] > 0.5 * m_all && o.med[x] < 1.5 * m_all)) return false;
        o.all += o.med[x] / 8.0;
    }
    o.span_ticks = (double)(last - first);
    o.ghz_med = o.ghz_min = o.ghz_max = 0.0;
    if (!ghz.empty()) {
        std::sort(ghz.begin(), ghz.end());
        o.ghz_med = ghz[ghz.size() / 2]; o.ghz_min = ghz.front(); o.ghz_max = ghz.back();
    }
    return true;
}

int hb_xcd_step(hb_xcd_state& c, int fam, const hb_stamp_set& s) {
    int flags = 0;
    hb_stamp_summary sm;
    if (!hb_stamps_summarise(s.stamps, s.G, sm)) { ++c.rejected; return HB_CAL_REJECTED; }
    ++c.samples;
    // Everything below is in terms of the PHYSICAL XCDs: group g of this launch ran on XCD sm.xcc[g] with the share stamp_w[g].  A launch
    // whose groups sat on other XCDs than the work list assumed (c.perm) moves the map (and the list is rebuilt for it); a map that keeps
    // moving makes shares meaningless: after three moves this index keeps equal shares.
    double run_w[8], med[8];
    for (int g = 0; g < 8; ++g) { run_w[sm.xcc[g]] = s.run_shares[g]; med[sm.xcc[g]] = sm.med[g]; }
    if (!std::equal(sm.xcc, sm.xcc + 8, c.perm)) {
        std::copy(sm.xcc, sm.xcc + 8, c.perm);
        flags |= HB_CAL_REBUILD;
        if (++c.perm_moves >= 3) { for (int x = 0; x < 8; ++x) c.w[x] = 1.0; c.locked = 2; }
    }
    // The fp32 kernel's automatic L2-sharing clusters (2 x 4 for the biggest searches) cut the L2-miss traffic by 60 % and cost cycles (the
    // soft sync, more slots, shorter segments).  Whether that pays is a property of the BOX: round 5's driver box -- held at 2.31 GHz by its
    // power budget -- ran 2.5 % FASTER with them (2314 vs 2374 ms), every box of rounds 5 and 6 that held 2.38-2.39 GHz ran 0.2-0.7 % slower at
    // the same clock.  So it is measured: launches with calibrated shares are timed with clusters (two), then without (two), by their
    // spans; the faster form stays for this index and is remembered for the device.  Five searches in all, then nothing changes any more.
    const std::array<int, 3> shape_now{{s.key[0], s.key[1], s.key[4]}};      // (query tiles, bank tiles, k: spans of one shape only)
    if (fam == 0 && s.auto_cluster && c.cl_state < 2 && c.rounds >= 1 && (c.cl_n_on + c.cl_n_off == 0 || shape_now == c.cl_shape)) {
        const bool clustered = s.key[5] != 17;      // (cluster shape q * 16 + b; 1 x 1 = 17)
        c.cl_shape = shape_now;
        if (c.cl_state == 0 && clustered) {
            c.cl_span_on = c.cl_n_on ? std::min(c.cl_span_on, sm.span_ticks) : sm.span_ticks;
            if (++c.cl_n_on >= 2) { c.cl_state = 1; flags |= HB_CAL_REBUILD; }
        } else if (c.cl_state == 1 && !clustered) {
            c.cl_span_off = c.cl_n_off ? std::min(c.cl_span_off, sm.span_ticks) : sm.span_ticks;
            if (++c.cl_n_off >= 2) {
                c.cl_state = 2; c.cl_choice = c.cl_span_on < c.cl_span_off ? 1 : 0;
                flags |= HB_CAL_REBUILD;
                flags |= HB_CAL_REMEMBER_CLUSTERS;
            }
        }
    }
    if (c.locked == 2) return flags;
    // The GUARD: shares are kept only while they measure faster.  Launches of one shape (the key) are compared by their span (first start to
    // last end, the minimum over a share set's launches: clock dips only ever lengthen one); a share set that has had two launches and is still
    // 0.15 % slower than the best set seen (fp16 family: three launches, 0.8 %) goes, the best set comes back, and this index stops calibrating that family (round 5's driver box
    // ran 1.6 % slower than the builder's boxes with shares spread +- 2.8 %, and its record could not say whether the shares were the reason).
    if (c.key != s.key) { c.key = s.key; c.best_span = 0.0; c.cur_n = 0; c.locked = 0; }
    if (c.cur_n > 0 && std::equal(run_w, run_w + 8, c.cur_w)) { c.cur_span = std::min(c.cur_span, sm.span_ticks); ++c.cur_n; }
    else {
        if (c.cur_n > 0 && (c.best_span == 0.0 || c.cur_span < c.best_span)) { c.best_span = c.cur_span; std::copy(c.cur_w, c.cur_w + 8, c.best_w); }
        std::copy(run_w, run_w + 8, c.cur_w); c.cur_span = sm.span_ticks; c.cur_n = 1;
    }
    // (the fp16 candidate kernel's launches scatter by +- 0.5 % from search to search and its stamps cover the last phase only: three launches and
    // 0.8 % there -- with the fp32 rule one box of round 6 went back to equal shares on a 0.4 % difference and kept them: 284 ms where shares give 275)
    const int guard_n = fam ? 3 : 2;
    const double guard_tol = fam ? 1.008 : 1.0015;
    if (c.best_span > 0.0 && c.cur_n >= guard_n && c.cur_span > c.best_span * guard_tol && !std::equal(c.cur_w, c.cur_w + 8, c.best_w)) {
        for (int x = 0; x < 8; ++x) c.w[x] = c.best_w[x];
        flags |= HB_CAL_REBUILD;
        c.locked = 1; ++c.reverts; ++c.rounds;
        flags |= HB_CAL_REMEMBER_SHARES;
        return flags;
    }
    if (c.locked) return flags;
    // (a duration that is off by e in a launch holding the part f of the work is mended by e x f of the whole share)
    double w[8], mean = 0.0, change = 0.0;
    for (int x = 0; x < 8; ++x) { w[x] = run_w[x] * (1.0 + s.frac * (sm.all / med[x] - 1.0)); mean += w[x] / 8.0; }
    for (int x = 0; x < 8; ++x) {
        w[x] = std::min(1.25, std::max(0.8, w[x] / mean));
        if (fam && c.rounds >= 2) w[x] = 0.5 * (w[x] + c.w[x]);      // the fp16 kernel's durations scatter by +- 0.5 % from search to search: damped ...
        change = std::max(change, std::fabs(w[x] / c.w[x] - 1.0));
    }
    // ... and a new work list (10 M x 768: 8 ms of host time) only for a change that is worth it
    const double worth = fam ? (c.rounds < 2 ? 0.003 : c.rounds < 4 ? 0.006 : 0.012) : (c.rounds < 2 ? 0.0015 : c.rounds < 6 ? 0.003 : 0.006);   // (fp16: 0.5 % / 0.8 % kept the shares moving: slower; fp32, round 6: 0.3 % for ever re-planned four times in twenty steps)
    if (change > worth) {
        for (int x = 0; x < 8; ++x) c.w[x] = w[x];
        flags |= HB_CAL_REBUILD;                            // rebuilt with the new shares by the caller
        flags |= HB_CAL_REMEMBER_SHARES;
    }
    ++c.rounds;
    return flags;
}

int hb_f16_choose(hb_f16_adapt& a) {
    const bool probe = (a.searches++ & 15) == 15;
    if (!probe && a.r12 > 0.5) return HB_F16_FP32;
    if (!probe && a.r1 > 0.5) return HB_F16_WIDE_FIRST;
    return HB_F16_CHAIN;
}
void hb_f16_observe(hb_f16_adapt& a, int how, int64_t nq, int64_t first_failed, int64_t reached_fp32) {
    if (nq <= 0 || how == HB_F16_FP32) return;                 // (the fp32 kernel right away: nothing was observed)
    if (how == HB_F16_CHAIN) a.r1 = 0.5 * a.r1 + 0.5 * (double)first_failed / (double)nq;      // (a wide-first search does not run the first pass)
    a.r12 = first_failed == 0 ? 0.5 * a.r12 : 0.5 * a.r12 + 0.5 * (double)reached_fp32 / (double)nq;
}

// ---- test hooks (no GPU): a calibration state fed with synthetic stamp sets -----------------------------------------------------------
struct hb_calibration { hb_xcd_state st; int fam; };
extern "C" void* hb_calibration_new(int fp16_kernel) { hb_calibration* h = new hb_calibration(); h->fam = fp16_kernel ? 1 : 0; return h; }
extern "C" void hb_calibration_free(void* h) { delete static_cast<hb_calibration*>(h); }
// the GROUP shares the next launch would run with (the physical shares through the group -> XCD map, divided by their mean), and the state
extern "C" int hb_calibration_state(const void* hv, double shares8[8], int64_t out[12]) {
    if (!hv || !shares8 || !out) return -1;
    const hb_xcd_state& c = static_cast<const hb_calibration*>(hv)->st;
    double mean = 0.0;
    for (int x = 0; x < 8; ++x) mean += c.w[x] / 8.0;
    for (int g = 0; g < 8; ++g) shares8[g] = c.w[c.perm[g]] / mean;
    out[0] = c.rounds; out[1] = c.locked; out[2] = c.reverts; out[3] = c.samples; out[4] = c.rejected; out[5] = c.perm_moves;
    out[6] = c.cl_state; out[7] = c.cl_choice; out[8] = c.cur_n; out[9] = c.perm[0]; out[10] = c.cl_n_on; out[11] = c.cl_n_off;
    return 0;
}
// one launch's stamps [G][2][4] (wg_stamp layout), the group shares it ran with, its shape key [6], whether its cluster shape was automatic
extern "C" int hb_calibration_feed(void* hv, const uint32_t* stamps, int G, const double run_shares8[8], const int key6[6], int auto_cluster, double frac) {
    if (!hv || !stamps || !run_shares8 || !key6) return -1;
    hb_calibration* h = static_cast<hb_calibration*>(hv);
    hb_stamp_set s;
    s.stamps = stamps; s.G = G; s.frac = frac; s.auto_cluster = auto_cluster;
    for (int x = 0; x < 8; ++x) s.run_shares[x] = run_shares8[x];
    for (int x = 0; x < 6; ++x) s.key[x] = key6[x];
    return hb_xcd_step(h->st, h->fam, s);
}
// the adaptive use of use_fp16 (mode 2): a fresh state walked through a stream of searches -- per search how it would run (out_how[i]) given
// what the searches before it saw: failing shares of the first certificate f1[i] and of the second pass f2[i] (of those that entered it)
extern "C" int hb_f16_adapt_replay(int n, const double* f1, const double* f2, int64_t nq, int* out_how) {
    if (n < 0 || !f1 || !f2 || !out_how || nq <= 0) return -1;
    hb_f16_adapt a;
    for (int i = 0; i < n; ++i) {
        const int how = hb_f16_choose(a);
        out_how[i] = how;
        const int64_t failed1 = (int64_t)std::llround(f1[i] * (double)nq);
        // chain: the second pass takes the first's failures, its own failures reach fp32; wide-first: the k' = 256 pass runs on all queries and
        // fails where BOTH would have (f1 x f2 of them)
        const int64_t first_failed = how == HB_F16_WIDE_FIRST ? (int64_t)std::llround(f1[i] * f2[i] * (double)nq) : failed1;
        const int64_t fp32 = how == HB_F16_WIDE_FIRST ? first_failed : (int64_t)std::llround(f2[i] * (double)failed1);
        hb_f16_observe(a, how, nq, first_failed, fp32);
    }
    return 0;
}
