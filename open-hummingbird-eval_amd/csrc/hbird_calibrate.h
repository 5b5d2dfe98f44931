// The decisions the kNN launcher takes from its workgroups' own time stamps (host side, plain C++: no HIP types) -- per-XCD work shares with
// their guard, and whether the fp32 L2-sharing clusters stay.  Shared by hbird_knn.hip (which feeds it the stamps of real launches) and by the
// host-only test hooks hb_calibration_* (tests/test_calibrate_cpu.py; also built under sanitizers with the planner: make plan_asan).
#pragma once
#include <array>
#include <stdint.h>

// What one launch's stamps say (wg_stamp, hbird_knn_dev.h: per block, at its start and at its end, {100 MHz real-time counter (low word),
// XCC id, shader-cycle counter lo, hi}).
struct hb_stamp_summary {
    double med[8];            // median duration (ticks) of the blocks equal to g mod 8
    int xcc[8];               // the XCD that group ran on
    double all;               // mean of the eight medians
    double span_ticks;        // first start to last end
    double ghz_med, ghz_min, ghz_max;     // shader cycles per tick x 100 MHz, over the workgroups
};
// -> false when the sample cannot be trusted (the region is zeroed before a stamping launch: a block that never stamped reads 0 / 0; blocks
// equal mod 8 that did NOT share an XCD, or two such groups on one XCD, say the dispatch order is not what the share groups assume; a
// duration far from the others' is a wrap or a preempted block).  WHICH XCD a group ran on is an output: HIP promises no placement, block 0
// usually lands on XCD 0 but need not (MI355X_MICROARCH.md, "Workgroup dispatch"), and the shares belong to the physical XCDs.
bool hb_stamps_summarise(const unsigned* st, int G, hb_stamp_summary& o);

// The calibration state of one kernel family of one index (hb_index::xcd_cal adds the HIP side: pinned stamps, event, what is pending).
struct hb_xcd_state {
    double w[8] = {1, 1, 1, 1, 1, 1, 1, 1};          // shares in use, per PHYSICAL XCD
    int rounds = 0;
    int samples = 0, rejected = 0;                   // stamp sets read / thrown away (hb_stamps_summarise)
    // the guard: shares stay only while launches of the same shape measure faster with them
    std::array<int, 6> key{{0, 0, 0, 0, 0, 0}};      // shape of the launches being compared: query tiles, bank tiles, workgroups, phases, k, cluster shape
    double cur_w[8] = {1, 1, 1, 1, 1, 1, 1, 1}, best_w[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    double cur_span = 0.0, best_span = 0.0;          // shortest launch (100 MHz ticks, first start to last end) with the current / the best share set
    int cur_n = 0, locked = 0, reverts = 0;          // locked: 1 = by the guard, 2 = the group -> XCD map kept moving (equal shares)
    int perm[8] = {0, 1, 2, 3, 4, 5, 6, 7};          // XCD that group g (blocks equal to g mod 8) was last seen on
    int perm_moves = 0;
    // fp32 family only: the automatic L2-sharing clusters of the biggest searches are kept only where they measure faster
    int cl_state = 0;                                // 0 = measuring with clusters, 1 = measuring without, 2 = decided
    int cl_choice = 1;                               // decided: 1 = clusters, 0 = none
    int cl_n_on = 0, cl_n_off = 0;
    std::array<int, 3> cl_shape{{0, 0, 0}};          // (query tiles, bank tiles, k) of the launches being compared
    double cl_span_on = 0.0, cl_span_off = 0.0;      // shortest qualifying launch with / without clusters (ticks)
};
// One stamp set of a launch that ran with the GROUP shares `run_shares` (group g = blocks equal to g mod 8).
struct hb_stamp_set {
    const unsigned* stamps; int G;
    double run_shares[8];
    std::array<int, 6> key;
    double frac;              // the stamped launch's part of the search's work (a phased search stamps its last launch)
    int auto_cluster;         // the cluster shape of that search was the automatic choice
};
enum { HB_CAL_REBUILD = 1, HB_CAL_REMEMBER_SHARES = 2, HB_CAL_REMEMBER_CLUSTERS = 4, HB_CAL_REJECTED = 8 };
// Feed one stamp set to the state: -> flags (HB_CAL_*): the work list must be rebuilt / the shares (the cluster decision) are worth
// remembering for the device / the set was thrown away.  fam: 0 = the fp32 kernels, 1 = the fp16 candidate kernel.
int hb_xcd_step(hb_xcd_state& c, int fam, const hb_stamp_set& s);

// ---- use_fp16, adaptive use (mode 2: what the plugin's use_fp16=True selects) ---------------------------------------------------------
// On a bank whose neighbours sit closer together than fp16 can tell apart most certificates fail, and passes that certify nothing are pure
// overhead.  The index keeps moving averages of the share of queries that failed the first certificate (r1) and of the share that reached
// the fp32 kernel (r12): r12 > 1/2 -> the fp32 kernel right away; r1 > 1/2 -> the first pass is skipped, ONE pass with k' = 256 serves all
// queries; every 16th search walks the whole chain again, so that a bank (or a query stream) that changes is noticed.
struct hb_f16_adapt { double r1 = 0.0, r12 = 0.0; int searches = 0; };
enum { HB_F16_CHAIN = 0, HB_F16_WIDE_FIRST = 1, HB_F16_FP32 = 2 };
int hb_f16_choose(hb_f16_adapt& a);                                                   // how the next search runs (counts it)
// what that search saw: `first_failed` of `nq` queries failed the first certificate it ran, `reached_fp32` went to the fp32 kernel
void hb_f16_observe(hb_f16_adapt& a, int how, int64_t nq, int64_t first_failed, int64_t reached_fp32);
