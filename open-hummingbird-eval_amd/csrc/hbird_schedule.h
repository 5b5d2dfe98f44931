// The kNN work list (host side, plain C++: no HIP types) -- shared by the HIP translation units of libhbird_hip.so and by the host-only
// sanitizer build of the planner (make plan_asan: g++ -fsanitize=address,undefined over hbird_schedule.cpp, tests/test_sanitizers_cpu.py).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

#define HB_QT 256        // query rows per workgroup tile (8 waves x 32 query columns)
#define HB_BT 256        // bank rows per tile (8 MFMA row tiles of 32)
#define HB_KC 16         // k extent of one LDS stage (two 8-wide fragment groups)

struct hb_seg {
    int q_tile;    // query tile index (HB_QT rows)
    int b_tile0;   // first bank tile (HB_BT rows)
    int n_tiles;   // bank tiles b_tile0, b_tile0 + stride, ... (ascending)
    int slot;      // partial-list slot this segment accumulates into
    int first;     // 1: slot starts empty, 0: continue from the stored lists
    int stride;    // bank-tile stride (1, or the cluster's bank ways: the members interleave the tiles of a range)
    int tile0;     // cluster clock (tiles) at the segment's first tile; next_tile0 at its end (INT_MAX: no more work)
    int next_tile0;
    int ord;       // ordinal of `slot` among the slots of its query tile, and their number (quota floors, hbird_knn.hip)
    int nsl;
};

#define HB_PHASE_CUTS 24      // most phase boundaries of a pool search (hb_build_schedule: clocks 1, 3, 6, 10, 16, 25, ... growing by half, twofold beyond 128)
#define HB_CLUSTER_MAX 8      // workgroups per L2-sharing cluster
#define HB_CLUSTER_LINE 32    // ints per cluster in the progress array (one 128-B line)

struct hb_schedule {
    int nqt = 0, nbt = 0, G = 0, panel = 0;
    int cq = 1, cb = 1;              // cluster shape: cq query tiles x cb interleaved bank tiles (1 x 1: no clusters)
    bool xcd_share = false;          // clusters: all clusters of an XCD walk the same query group (hb_build_clustered)
    std::vector<double> xcd_w;       // work share per XCD group (blocks equal mod 8), empty = equal shares (hb_build_schedule)
    std::vector<hb_seg> segs;        // grouped by workgroup
    std::vector<int> wg_off;         // G+1 offsets into segs
    std::vector<int> wg_member;      // per block: cluster * HB_CLUSTER_LINE + member (progress word of the block)
    bool phased = false;             // pool searches: the segment lists are cut at phase_clock (hb_finish_schedule) ...
    std::vector<int> phase_clock;    // ... clock values (tiles dealt per workgroup; x the worker's share with uneven shares) where a phase ends ...
    bool cuts_scaled = false;        // ... (true: the cuts of this list follow the workers' shares, hb_finish_schedule) ...
    std::vector<int> phase_bounds;   // ... [cuts][G]: per block the first segment at or beyond each of them
    std::vector<int> qt_off;         // nqt+1 offsets into qt_slots
    std::vector<int> qt_slots;       // slots that hold partial lists of each query tile
    int n_slots = 0;
    int max_slots_per_qt = 0;
    int n_clusters = 0;
};

void hb_build_schedule(int nqt, int nbt, int G, int panel_tiles, hb_schedule& out, int cq = 1, int cb = 1, bool phased = false,
                       bool xcd_share = false, const double* xcd_w = nullptr);
int hb_default_panel(int nqt, int G, size_t tile_bytes, int cq = 1, int cb = 1);
// automatic cluster shape for a search (1 x 1 when clusters do not apply)
void hb_default_cluster(int nqt, int nbt, int G, bool fp32_kernel, int* cq, int* cb);

int hb_fail(const std::string& msg);
