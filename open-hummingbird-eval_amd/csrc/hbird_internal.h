// Internal declarations shared by the HIP translation units of libhbird_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <array>
#include <string>
#include <vector>
#include "hbird_schedule.h"
#include "hbird_calibrate.h"

// ---- tile geometry of the kNN kernel (see DESIGN.md "Data layout in HBM") -------------------
#define HB_RT 32         // rows per fragment tile (MFMA 32x32x2)
#define HB_POOL_MAX 512  // largest candidate pool per (slot, query) in global memory (k > HB_KL)
#define HB_KL 32         // per-query list capacity kept in LDS (k <= HB_KL on the fused path)
#define HB_THREADS 512
#define HB_WAVES 8
#define HB_BLK 256       // floats per fragment block (32 rows x 8 k) = 1 KiB

#define HB_STAGE_BYTES (16384 + 16384 + 1024)   // A fragments, B fragments, bank-row init values
#define HB_LDS_LISTS (2 * HB_STAGE_BYTES)
#define HB_LDS_SCRATCH (HB_LDS_LISTS + 2 * HB_QT * HB_KL * 4)
#define HB_LDS_TOTAL (HB_LDS_SCRATCH + HB_WAVES * 8 * 64 * 4)

#define HB_ID_NONE 0xFFFFFFFFu

// Optional ROCTx range around a host-side phase (shows up under `rocprofv3 --marker-trace`).  The marker library
// (librocprofiler-sdk-roctx.so, else libroctx64.so) is looked up at run time; without it the ranges are no-ops.
struct hb_range {
    explicit hb_range(const char* name);
    ~hb_range();
    hb_range(const hb_range&) = delete;
    hb_range& operator=(const hb_range&) = delete;
};

struct hb_index {
    int d = 0, dp = 0, g8 = 0, metric = 0, device = 0;
    hipStream_t stream = nullptr;
    int64_t ntotal = 0, cap_rows = 0;      // cap_rows is a multiple of HB_BT
    float* tiles = nullptr;                // fragment-tiled bank  [cap_rows/32][g8][256]
    float* binit = nullptr;                // per-row accumulator init [cap_rows]
    float* bnorm = nullptr;                // per-row L2 norm (fp32)  [cap_rows]
    float* labels = nullptr;               // [lab_cap][c] fp32 (label_P == 0) ...
    uint16_t* labels16 = nullptr;          // ... or [lab_cap][lab_stride()] uint16 counts j of values j / label_P (hb_index_set_label_denominator): rows padded to 8 counts = 16 B
    int label_P = 0;
    int* lab_flag = nullptr;               // sticky device flag: a label value was not a multiple of 1 / label_P
    int64_t lab_checked = 0;               // label rows whose conversion has been checked (one read-back after the table grew)
    int c = 0;
    int64_t nlabels = 0, lab_cap = 0;
    int lab_stride() const { return label_P ? (c + 7) & ~7 : c; }   // elements per stored label row (counts: 16-byte rows for K5's wide gather)
    int num_cu = 256;
    // optional borrowed tables covering a GLOBAL id range (multi-GPU: all-gathered labels / norms)
    const float* ext_labels = nullptr; const float* ext_bnorm = nullptr; int64_t ext_n = 0, ext_base = 0;
    const uint16_t* ext_labels16 = nullptr; int ext_P = 0;   // the borrowed label table as counts (hb_index_set_label_count_table)
    // search workspace (grown on demand, reused)
    float* q_tiles = nullptr; size_t q_tiles_bytes = 0;
    float* q_aux = nullptr; size_t q_aux_bytes = 0;      // qn2 (chain) and qnorm (fp32), 2*nq floats
    char* state = nullptr; size_t state_bytes = 0;
    char* sched_dev = nullptr; size_t sched_bytes = 0;
    char* tmp = nullptr; size_t tmp_bytes = 0;           // staging for host<->device convenience paths
    hb_schedule sched;                                   // cached for (nqt, nbt)
    int force_G = 0, force_panel = 0;                    // test/tuning overrides
    int force_cq = 0, force_cb = 0;                      // cluster shape override (0 = automatic)
    int sync_lag = -1;                                   // soft-sync lag in stages (-1 = automatic, 0 = no sync)
    int xcd_share = 0;                                   // clustered work lists: 0 = automatic, 1 = off, 2 = on (hb_index_set_cluster_sharing)
    // work shares per XCD group (blocks equal mod 8): hb_index_set_xcd_weights / calibrated from the workgroups' own time stamps.
    // Two families with shares of their own: [0] the fp32 kernels, [1] the fp16 candidate kernel (power-limited: its XCDs differ by other amounts)
    struct xcd_cal : hb_xcd_state {                      // (the decisions' state: hbird_calibrate.h; here the HIP side)
        unsigned* stamp_host = nullptr;                  // pinned copy of the last calibrating launch's per-block stamps ...
        hipEvent_t stamp_ev = nullptr;                   // ... complete when this event is
        int stamp_pending = 0;                           // blocks of that launch (0: nothing to read)
        double stamp_w[8] = {1, 1, 1, 1, 1, 1, 1, 1};    // the GROUP shares that launch ran with
        double stamp_frac = 1.0;                         // ... and its part of the search's work (phased searches stamp their LAST launch)
        std::array<int, 6> stamp_key{{0, 0, 0, 0, 0, 0}};   // ... its shape
        int stamp_auto_cluster = 0;                      // ... and whether its cluster shape was the automatic choice
    } xcal[2];
    int64_t sched_builds = 0;                            // work lists built for this index (a re-plan costs host time: 8 ms at 10 M x 768)
    int xcd_balance = 0;                                 // 0 = automatic (big fp32 searches calibrate the shares from their own workgroups' durations), 1 = equal shares, 2 = as set
    const int* cl_stats_dev = nullptr;                   // {checks, spins, timeouts} of the last clustered launch (in `state`)
    // fp16 candidate mode (use_fp16): fp16 copies of the bank / query fragment tiles, candidate buffers
    int fp16 = 0, dp16 = 0;
    void* tiles16 = nullptr; int64_t f16_cap_rows = 0, f16_rows = 0;
    int* f16_flag = nullptr; int f16_overflow = 0;       // a finite bank value overflowed fp16: the fp32 kernel serves this bank
    // ... and, where memory allows, the bank once more as plain fp32 rows [row][rows32_rs] for the exact re-rank (hbird_knn_f16.hip)
    float* rows32 = nullptr; int64_t rows32_cap_rows = 0, rows32_rows = 0; int rows32_rs = 0;
    int rerank_copy = 0;                                 // 0 = automatic, 1 = always, 2 = never (hb_index_set_rerank_copy)
    int64_t rows32_declined_cap = -1;                    // automatic mode found no room for the copy at this capacity: not asked again per search
    void* q16 = nullptr; size_t q16_bytes = 0;
    char* cand = nullptr; size_t cand_bytes = 0;
    float* bmax = nullptr;                               // device scalar: max bank-row norm
    char* mtmp = nullptr; size_t mtmp_bytes = 0;         // first-level lists of a two-level merge
    char* fb = nullptr; size_t fb_bytes = 0;             // workspace of the uncertified queries (a caller's search) ...
    char* fb1 = nullptr; size_t fb1_bytes = 0;           // ... and of their second fp16 pass
    int64_t last_fp16_fallbacks = 0;                     // queries of the last use_fp16 search that the fp32 kernel had to answer ...
    int64_t last_fp16_escalated = 0;                     // ... and queries whose first certificate failed (second fp16 pass, k' = 256, seeded floors)
    int fp16_escalation = 0;                             // 0 = on (automatic), 1 = off: uncertified queries go straight to the fp32 kernel (round 5)
    hb_f16_adapt f16_adapt;                              // adaptive use of use_fp16 in mode 2 (hbird_calibrate.h): moving averages of the failing shares
    int f16_skipped = 0;
    const float* ceil_s_dev = nullptr; const unsigned* ceil_i_dev = nullptr;    // a later pass of a search with k > 256 (hb_launch_knn_bigk)
    char* bigk = nullptr; size_t bigk_bytes = 0;         // its workspace: one pass's lists and the ceilings
    int esc_level = 0;                                   // inside hb_launch_knn: 0 = a caller's search, 1 = the second fp16 pass, 2 = the fp32 search of what is left
    const float* seed_dev = nullptr;                     // per-query floors (scores) a nested search starts from
    hb_schedule sched_esc; char* sched_esc_dev = nullptr; size_t sched_esc_bytes = 0;   // the nested searches' work list (the caller's stays cached)
    int score_output = 0;                                // 1: searches return ordering scores instead of distances
    int variant = 0;                                     // kernel selection for A/B runs and tests (hb_index_set_variant)
    int phases_on = 1;                                   // pool searches are launched in phases (hb_index_set_search_options)
    const unsigned* wg_stamp_dev = nullptr; int wg_stamp_blocks = 0;   // per-block stamps of the last timed kNN launch (hb_index_wg_stamps, hb_index_kernel_clock)
    long long small_limit = 0;                           // stages per workgroup below which a search counts as small (0 = default)
    double last_knn_ms = 0.0;                            // HIP-event time of the last knn kernel launch
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int time_kernels = 0;
};

void hb_set_error(const std::string& msg);
int hb_fail(const std::string& msg);
#define HB_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess)                                                                \
            return hb_fail(std::string(#call) + ": " + hipGetErrorString(_e));               \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) acts on the CURRENT device only: remembers the (kernel, device)
// pairs already configured (thread-safe), so an index on a second GPU of the same process gets its own call.
int hb_ensure_dyn_lds(const void* kernel, int bytes);

// kernels/launchers (each returns 0 or a negative status after hb_set_error)
int hb_launch_rows_to_tiles(const float* src, int64_t n_rows, int d, int dp, int64_t row0, float* tiles,
                            float* binit, float* bnorm, int metric, int normalize, int is_bank, hipStream_t s);
int hb_launch_scores_to_l2(const float* qn2, int64_t nq, int k, float* dist_inout, hipStream_t s);
int hb_launch_query_aux(const float* q, int64_t nq, int d, float* qn2, float* qnorm, hipStream_t s);
int hb_launch_tiles_to_rows(const float* tiles, int g8, int d, const int64_t* ids, int64_t n, int64_t id_base,
                            float* out, hipStream_t s);
int hb_launch_knn(hb_index* ix, const float* q_dev, int64_t nq, int k, int64_t id_base, int64_t* out_idx, float* out_dist);
int hb_launch_tiles_to_f16(const float* t32, int g8, _Float16* t16, int g16, int64_t n_row_tiles, int64_t rt0, int* overflow,
                           hipStream_t s);
int hb_launch_tiles_to_rows(const float* t32, int g8, float* rows, int rs, int64_t n_row_tiles, int64_t rt0, hipStream_t s);
int hb_launch_rerank_rows(const float* rows, int rs, const float* binit, int d, const float* q, const float* qn2,
                          const int64_t* cand, const float* cand_score, const float* qnorm, const float* bmax,
                          unsigned char* certified, int kc, int64_t nq, int k, int64_t id_base, int metric, int out_metric,
                          int64_t ntotal, int64_t* out_idx, float* out_dist, hipStream_t s, const float* seed_in = nullptr, float* kth_out = nullptr,
                          float* floor_out = nullptr);
int hb_launch_rerank(const float* tiles, const float* binit, int g8, int d, const float* q, const float* qn2,
                     const int64_t* cand, const float* cand_score, const float* qnorm, const float* bmax,
                     unsigned char* certified, int kc, int64_t nq, int k, int64_t id_base, int metric, int out_metric, int64_t ntotal, int64_t* out_idx,
                     float* out_dist, hipStream_t s, const float* seed_in = nullptr, float* kth_out = nullptr, float* floor_out = nullptr);
int hb_launch_bnorm_max(const float* bnorm, int64_t n, float* bmax, hipStream_t s);
int hb_launch_scatter_rows(const int64_t* rows, int64_t n, int k, const int64_t* src_idx, const float* src_dist,
                           int64_t* out_idx, float* out_dist, hipStream_t s);
struct knn16_args;
int hb_knn_f16_launch(const knn16_args& args, int grid, hipStream_t s);
// label storage: fp32 values, or uint16 counts of values j / P (exactly the fp32 value: K2 computes (float)j / (float)P)
int hb_launch_labels_to_counts(const float* src, int64_t rows, int c, int dst_stride, int P, uint16_t* dst, int* flag, hipStream_t s);
int hb_launch_gather_label_counts(const uint16_t* src, int64_t src_rows, int c, int src_stride, int P, const int64_t* ids, int64_t n, float* out, hipStream_t s);
int hb_labels_checked(hb_index* ix);   // 0, or fails when a stored label was not a multiple of 1 / label_P
int hb_launch_aggregate(const hb_index* ix, const float* qnorm, const int64_t* idx, const float* dist, int64_t nq,
                        int k, int64_t id_base, float beta, float* out, hipStream_t s, const float* norms_all = nullptr, int64_t n_all = 0);
int hb_launch_merge_parts(const float* dist_parts, const int64_t* idx_parts, int parts, int64_t nq, int k, int metric,
                          int64_t dist_stride, int64_t idx_stride, int64_t* out_idx, float* out_dist, hipStream_t s);
int hb_launch_patch_label_hist(const int64_t* y, int64_t B, int H, int W, int ps, int C, int map255, float* out,
                               hipStream_t s);
int hb_launch_normalize_rows(const float* x, int64_t n, int d, float* out, hipStream_t s);
int hb_launch_gather_rows(const float* src, int64_t src_rows, int width, const int64_t* ids, int64_t n, float* out,
                          hipStream_t s);
int hb_launch_upsample_argmax(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out,
                              hipStream_t s);
int hb_launch_upsample_argmax_confusion(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out, const int64_t* gt,
                                        int num_gt, int num_pred, int64_t ignore, int has_ignore, unsigned long long* conf, hipStream_t s);
int hb_launch_upsample_accumulate(const float* label_hat, int64_t B, int S, int C, int win_h, int win_w, float* acc, int H,
                                  int W, int y0, int x0, hipStream_t s);
int hb_launch_argmax_channels(const float* acc, int64_t n, int C, int64_t* out, hipStream_t s);
int hb_launch_confusion(const int64_t* gt, const int64_t* pred, int64_t n, int num_gt, int num_pred, int64_t ignore,
                        int has_ignore, unsigned long long* conf, hipStream_t s);
