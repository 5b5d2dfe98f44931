// extern "C" surface of libhbird_hip.so (declared in include/hbird_hip.h).
#include "../../include/hbird_hip.h"
#include "hbird_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <set>
#include <utility>

static thread_local std::string g_err;
void hb_set_error(const std::string& msg) { g_err = msg; }
int hb_fail(const std::string& msg) { g_err = msg; return -1; }

namespace {
struct roctx_api {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    roctx_api() {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const roctx_api& roctx() { static const roctx_api api; return api; }
}  // namespace
hb_range::hb_range(const char* name) { if (roctx().push) roctx().push(name); }
hb_range::~hb_range() { if (roctx().pop) roctx().pop(); }

int hb_ensure_dyn_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    HB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return 0;
    HB_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kernel, dev});
    return 0;
}

extern "C" const char* hb_last_error(void) { return g_err.c_str(); }

extern "C" int hb_device_count(int* n) {
    if (!n) return hb_fail("hb_device_count: n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
    *n = c;
    return 0;
}

extern "C" int hb_index_create(int d, int metric, int device, hb_index_t** out) {
    if (!out) return hb_fail("hb_index_create: out is NULL");
    *out = nullptr;
    if (d <= 0) return hb_fail("hb_index_create: d must be positive");
    if (metric != HB_METRIC_IP && metric != HB_METRIC_L2) return hb_fail("hb_index_create: unsupported metric");
    int ndev = 0;
    hb_device_count(&ndev);
    if (ndev < 1) return hb_fail("hb_index_create: no GPUs available");
    if (device < 0 || device >= ndev)
        return hb_fail("hb_index_create: invalid GPU id " + std::to_string(device) + ", available 0-" + std::to_string(ndev - 1));
    HB_HIP(hipSetDevice(device));
    hb_index* ix = new hb_index();
    ix->d = d; ix->dp = (d + HB_KC - 1) / HB_KC * HB_KC; ix->g8 = ix->dp / 8; ix->dp16 = (d + 127) / 128 * 128; ix->metric = metric; ix->device = device;
    hipDeviceProp_t prop;
    HB_HIP(hipGetDeviceProperties(&prop, device));
    ix->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HB_HIP(hipMalloc((void**)&ix->bmax, 4));
    HB_HIP(hipMemset(ix->bmax, 0, 4));
    HB_HIP(hipEventCreate(&ix->ev0));
    HB_HIP(hipEventCreate(&ix->ev1));
    *out = ix;
    return 0;
}

extern "C" int hb_index_free(hb_index_t* ix) {
    if (!ix) return 0;
    (void)hipSetDevice(ix->device);
    (void)hipStreamSynchronize(ix->stream);
    void* ptrs[] = {ix->tiles, ix->binit, ix->bnorm, ix->labels, ix->q_tiles, ix->q_aux, ix->state, ix->sched_dev, ix->tmp,
                    ix->tiles16, ix->q16, ix->cand, ix->bmax, ix->fb, ix->fb1, ix->sched_esc_dev, ix->bigk, ix->mtmp, ix->f16_flag, ix->labels16, ix->lab_flag, ix->rows32};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& c : ix->xcal) { if (c.stamp_host) (void)hipHostFree(c.stamp_host); if (c.stamp_ev) (void)hipEventDestroy(c.stamp_ev); }
    if (ix->ev0) (void)hipEventDestroy(ix->ev0);
    if (ix->ev1) (void)hipEventDestroy(ix->ev1);
    delete ix;
    return 0;
}

extern "C" int hb_index_set_stream(hb_index_t* ix, void* s) {
    if (!ix) return hb_fail("hb_index_set_stream: NULL index handle");
    ix->stream = (hipStream_t)s;
    return 0;
}
// the two counters have no status channel: a NULL handle reads as -1 rows (and sets hb_last_error)
extern "C" int64_t hb_index_ntotal(const hb_index_t* ix) {
    if (!ix) { hb_set_error("hb_index_ntotal: NULL index handle"); return -1; }
    return ix->ntotal;
}
extern "C" int64_t hb_index_nlabels(const hb_index_t* ix) {
    if (!ix) { hb_set_error("hb_index_nlabels: NULL index handle"); return -1; }
    return ix->nlabels;
}
extern "C" int hb_index_set_timing(hb_index_t* ix, int enable) {
    if (!ix) return hb_fail("hb_index_set_timing: NULL index handle");
    ix->time_kernels = enable;
    return 0;
}
extern "C" int hb_index_last_knn_ms(const hb_index_t* ix, double* ms) {
    if (!ix || !ms) return hb_fail("hb_index_last_knn_ms: NULL pointer");
    *ms = ix->last_knn_ms;
    return 0;
}
extern "C" int hb_index_set_tuning(hb_index_t* ix, int workgroups, int panel_tiles) {
    if (!ix) return hb_fail("hb_index_set_tuning: NULL index handle");
    if (workgroups < 0 || panel_tiles < 0) return hb_fail("hb_index_set_tuning: negative value");
    ix->force_G = workgroups; ix->force_panel = panel_tiles; ix->sched = hb_schedule(); return 0;
}
extern "C" int hb_index_set_fp16(hb_index_t* ix, int enable) {
    if (!ix) return hb_fail("hb_index_set_fp16: NULL index handle");
    ix->fp16 = enable == 2 ? 2 : (enable ? 1 : 0);   // 2: only where it pays (hb_launch_knn)
    ix->f16_adapt = hb_f16_adapt();
    return 0;
}

extern "C" int hb_index_set_fp16_escalation(hb_index_t* ix, int mode) {
    if (!ix) return hb_fail("hb_index_set_fp16_escalation: NULL index handle");
    if (mode < 0 || mode > 1) return hb_fail("hb_index_set_fp16_escalation: mode must be 0 (second fp16 pass before the fp32 kernel) or 1 (straight to the fp32 kernel)");
    ix->fp16_escalation = mode;
    return 0;
}
extern "C" int hb_index_last_fp16_escalated(const hb_index_t* ix, int64_t* n) {
    if (!ix || !n) return hb_fail("hb_index_last_fp16_escalated: NULL pointer");
    *n = ix->last_fp16_escalated;
    return 0;
}
extern "C" int hb_index_last_fp16_fallbacks(const hb_index_t* ix, int64_t* n) {
    if (!ix || !n) return hb_fail("hb_index_last_fp16_fallbacks: NULL pointer");
    *n = ix->last_fp16_fallbacks;
    return 0;
}

// (hb_schedule_plan / _phased / _shared: hbird_schedule.cpp)

extern "C" int hb_index_set_cluster_sharing(hb_index_t* ix, int mode) {
    if (!ix) return hb_fail("hb_index_set_cluster_sharing: NULL index handle");
    if (mode < 0 || mode > 2) return hb_fail("hb_index_set_cluster_sharing: mode must be 0 (automatic), 1 (off) or 2 (on)");
    ix->xcd_share = mode; ix->sched = hb_schedule();
    return 0;
}

extern "C" int hb_index_cluster_stats(hb_index_t* ix, int64_t out[4]) {
    if (!ix || !out) return hb_fail("hb_index_cluster_stats: NULL pointer");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!ix->cl_stats_dev) return 0;
    HB_HIP(hipSetDevice(ix->device));
    int h[4] = {0, 0, 0, 0};
    HB_HIP(hipMemcpyAsync(h, ix->cl_stats_dev, sizeof(h), hipMemcpyDeviceToHost, ix->stream));
    HB_HIP(hipStreamSynchronize(ix->stream));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];   // [3]: diagnostic builds only
    return 0;
}

extern "C" int hb_index_set_cluster(hb_index_t* ix, int cluster_q, int cluster_b, int sync_lag) {
    if (!ix) return hb_fail("hb_index_set_cluster: NULL index handle");
    if (cluster_q < 0 || cluster_b < 0 || cluster_q * cluster_b > HB_CLUSTER_MAX)
        return hb_fail("hb_index_set_cluster: cluster shape must be 0 x 0 (automatic), 1 x 1 (off) or q x b with q * b <= " + std::to_string(HB_CLUSTER_MAX));
    ix->force_cq = cluster_q; ix->force_cb = cluster_b; ix->sync_lag = sync_lag; ix->sched = hb_schedule();
    return 0;
}

extern "C" int hb_index_set_variant(hb_index_t* ix, int variant) {
    if (!ix) return hb_fail("hb_index_set_variant: NULL index handle");
    if (variant < 0 || variant > 6) return hb_fail("hb_index_set_variant: unknown kernel variant");
    if (variant == 1 || variant == 2 || variant == 5)
        return hb_fail("hb_index_set_variant: variants 1 (4-wave fp32 kernel), 2 (first fp16 design) and 5 (16x16x32 fp16 kernel) were removed in round 4 (same bits, not faster)");
    ix->variant = variant;
    return 0;
}
extern "C" int hb_index_set_search_options(hb_index_t* ix, int phases, int64_t small_limit_stages) {
    if (!ix) return hb_fail("hb_index_set_search_options: NULL index handle");
    if (small_limit_stages < 0) return hb_fail("hb_index_set_search_options: negative limit");
    ix->phases_on = phases ? 1 : 0; ix->small_limit = small_limit_stages; ix->sched = hb_schedule();
    return 0;
}
extern "C" int hb_index_set_xcd_weights(hb_index_t* ix, int mode, const double* w8) {
    if (!ix) return hb_fail("hb_index_set_xcd_weights: NULL index handle");
    if (mode < 0 || mode > 2) return hb_fail("hb_index_set_xcd_weights: mode must be 0 (calibrated), 1 (equal shares) or 2 (the given shares)");
    if (mode == 2 && !w8) return hb_fail("hb_index_set_xcd_weights: mode 2 needs eight shares");
    for (int x = 0; x < 8; ++x) {
        const double w = mode == 2 ? w8[x] : 1.0;
        if (!(w > 0.25 && w < 4.0)) return hb_fail("hb_index_set_xcd_weights: every share must lie in (0.25, 4)");
        ix->xcal[0].w[x] = w; ix->xcal[1].w[x] = w;
    }
    ix->xcd_balance = mode;
    for (auto& c : ix->xcal) { c.stamp_pending = 0; c.rounds = mode == 0 ? 0 : 1; c.locked = 0; c.cur_n = 0; c.best_span = 0.0; c.perm_moves = 0; }
    // (the clusters' own decision -- xcal[0].cl_* -- is a property of the box: it survives a change of the share mode)
    ix->sched = hb_schedule();
    return 0;
}
extern "C" int hb_index_xcd_weights(const hb_index_t* ix, int fp16_kernel, double* w8, int* rounds) {
    if (!ix || !w8) return hb_fail("hb_index_xcd_weights: NULL pointer");
    const hb_index::xcd_cal& c = ix->xcal[fp16_kernel ? 1 : 0];
    for (int x = 0; x < 8; ++x) w8[x] = c.w[x];
    if (rounds) *rounds = c.rounds;
    return 0;
}
extern "C" int hb_index_wg_stamps(hb_index_t* ix, uint32_t* out, int max_blocks, int* workgroups) {
    if (!ix || !out || !workgroups) return hb_fail("hb_index_wg_stamps: NULL pointer");
    *workgroups = 0;
    if (!ix->wg_stamp_dev) return 0;
    if (ix->wg_stamp_blocks > max_blocks) return hb_fail("hb_index_wg_stamps: the buffer is too small");
    HB_HIP(hipSetDevice(ix->device));
    std::vector<unsigned> h((size_t)ix->wg_stamp_blocks * 8);     // device layout: [block][start | end][4] (wg_stamp, hbird_knn_dev.h)
    HB_HIP(hipMemcpyAsync(h.data(), ix->wg_stamp_dev, h.size() * 4, hipMemcpyDeviceToHost, ix->stream));
    HB_HIP(hipStreamSynchronize(ix->stream));
    for (int b = 0; b < ix->wg_stamp_blocks; ++b) { out[4 * b] = h[8 * b]; out[4 * b + 1] = h[8 * b + 4]; out[4 * b + 2] = h[8 * b + 1]; out[4 * b + 3] = 0; }
    *workgroups = ix->wg_stamp_blocks;
    return 0;
}
extern "C" int hb_index_kernel_clock(hb_index_t* ix, double out[4]) {
    if (!ix || !out) return hb_fail("hb_index_kernel_clock: NULL pointer");
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if (!ix->wg_stamp_dev || ix->wg_stamp_blocks <= 0) return 0;
    HB_HIP(hipSetDevice(ix->device));
    const int G = ix->wg_stamp_blocks;
    std::vector<unsigned> h((size_t)G * 8);
    HB_HIP(hipMemcpyAsync(h.data(), ix->wg_stamp_dev, (size_t)G * 32, hipMemcpyDeviceToHost, ix->stream));
    HB_HIP(hipStreamSynchronize(ix->stream));
    std::vector<double> ghz;
    long long first = 0, last = 0;
    bool any = false;
    for (int b = 0; b < G; ++b) {
        const unsigned* sb = h.data() + 8 * (size_t)b;
        const unsigned t0 = sb[0], t1 = sb[4];
        if (t0 == 0u && t1 == 0u) continue;             // the block did not stamp
        const unsigned d = t1 - t0;
        const long long rs = (long long)(int)(t0 - h[0]), re = rs + (long long)d;
        if (!any) { first = rs; last = re; any = true; }
        first = std::min(first, rs); last = std::max(last, re);
        const unsigned long long c0 = ((unsigned long long)sb[3] << 32) | sb[2], c1 = ((unsigned long long)sb[7] << 32) | sb[6];
        if (c1 > c0 && d >= 100u && d < 0x7FFFFFFFu) ghz.push_back((double)(c1 - c0) / (double)d * 0.1);
    }
    if (ghz.empty()) return 0;
    std::sort(ghz.begin(), ghz.end());
    out[0] = ghz[ghz.size() / 2]; out[1] = ghz.front(); out[2] = ghz.back(); out[3] = (double)(last - first) * 1e-5;   // 10 ns ticks -> ms
    return 0;
}
extern "C" int hb_index_xcd_stats(const hb_index_t* ix, int fp16_kernel, double out[12]) {
    if (!ix || !out) return hb_fail("hb_index_xcd_stats: NULL pointer");
    const hb_index::xcd_cal& c = ix->xcal[fp16_kernel ? 1 : 0];
    out[0] = c.rounds; out[1] = c.locked; out[2] = c.reverts; out[3] = c.samples; out[4] = c.rejected;
    out[5] = c.best_span * 1e-5; out[6] = c.cur_span * 1e-5; out[7] = (double)ix->sched_builds;
    out[8] = c.perm_moves; out[9] = c.perm[0];
    out[10] = c.cl_state == 2 ? c.cl_choice : -1; out[11] = (c.cl_span_on - c.cl_span_off) * 1e-5;
    return 0;
}
extern "C" int hb_index_set_rerank_copy(hb_index_t* ix, int mode) {
    if (!ix) return hb_fail("hb_index_set_rerank_copy: NULL index handle");
    if (mode < 0 || mode > 2) return hb_fail("hb_index_set_rerank_copy: mode must be 0 (automatic), 1 (always) or 2 (never)");
    ix->rerank_copy = mode; ix->rows32_declined_cap = -1;
    if (mode == 2 && ix->rows32) {
        (void)hipSetDevice(ix->device);
        HB_HIP(hipStreamSynchronize(ix->stream));
        HB_HIP(hipFree(ix->rows32));
        ix->rows32 = nullptr; ix->rows32_cap_rows = 0; ix->rows32_rows = 0;
    }
    return 0;
}
extern "C" int hb_index_rerank_copy_bytes(const hb_index_t* ix, int64_t* bytes) {
    if (!ix || !bytes) return hb_fail("hb_index_rerank_copy_bytes: NULL argument");
    *bytes = ix->rows32 ? ix->rows32_cap_rows * (int64_t)ix->rows32_rs * 4 : 0;
    return 0;
}
extern "C" int hb_index_schedule_info(const hb_index_t* ix, int64_t out[8]) {
    if (!ix) return hb_fail("hb_index_schedule_info: NULL index handle");
    const hb_schedule& s = ix->sched;
    out[0] = s.G; out[1] = (int64_t)s.segs.size(); out[2] = s.n_slots; out[3] = s.panel; out[4] = s.max_slots_per_qt;
    out[5] = s.nqt; out[6] = s.nbt; out[7] = s.cq * 16 + s.cb;
    return 0;
}

static int grow(void** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) HB_HIP(hipFree(*p));
    *p = nullptr; *have = 0;
    HB_HIP(hipMalloc(p, need));
    *have = need;
    return 0;
}

extern "C" int hb_index_reserve(hb_index_t* ix, int64_t n_rows) {
    if (!ix) return hb_fail("hb_index_reserve: NULL index handle");
    HB_HIP(hipSetDevice(ix->device));
    int64_t cap = (n_rows + HB_BT - 1) / HB_BT * HB_BT;
    if (cap <= ix->cap_rows) return 0;
    hipStream_t s = ix->stream;
    float *tiles = nullptr, *binit = nullptr, *bnorm = nullptr;
    const size_t tb = (size_t)cap * ix->dp * 4;
    HB_HIP(hipMalloc((void**)&tiles, tb));
    HB_HIP(hipMalloc((void**)&binit, (size_t)cap * 4));
    HB_HIP(hipMalloc((void**)&bnorm, (size_t)cap * 4));
    const size_t old_tb = (size_t)ix->cap_rows * ix->dp * 4;
    if (ix->cap_rows > 0) {
        // fragment tiles are row-tile major, so the old bank is a prefix of the new one
        HB_HIP(hipMemcpyAsync(tiles, ix->tiles, old_tb, hipMemcpyDeviceToDevice, s));
        HB_HIP(hipMemcpyAsync(binit, ix->binit, (size_t)ix->cap_rows * 4, hipMemcpyDeviceToDevice, s));
        HB_HIP(hipMemcpyAsync(bnorm, ix->bnorm, (size_t)ix->cap_rows * 4, hipMemcpyDeviceToDevice, s));
    }
    HB_HIP(hipMemsetAsync((char*)tiles + old_tb, 0, tb - old_tb, s));
    // padding rows start from -inf so they can never enter a top-k list
    HB_HIP(hipMemsetD32Async((hipDeviceptr_t)(binit + ix->cap_rows), 0xFF800000u, (size_t)(cap - ix->cap_rows), s));
    HB_HIP(hipMemsetAsync(bnorm + ix->cap_rows, 0, (size_t)(cap - ix->cap_rows) * 4, s));
    HB_HIP(hipStreamSynchronize(s));
    if (ix->tiles) { HB_HIP(hipFree(ix->tiles)); HB_HIP(hipFree(ix->binit)); HB_HIP(hipFree(ix->bnorm)); }
    ix->tiles = tiles; ix->binit = binit; ix->bnorm = bnorm; ix->cap_rows = cap;
    return 0;
}

extern "C" int hb_index_reset(hb_index_t* ix) {
    if (!ix) return hb_fail("hb_index_reset: NULL index handle");
    HB_HIP(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    if (ix->cap_rows > 0) {
        HB_HIP(hipMemsetAsync(ix->tiles, 0, (size_t)ix->cap_rows * ix->dp * 4, s));
        HB_HIP(hipMemsetD32Async((hipDeviceptr_t)ix->binit, 0xFF800000u, (size_t)ix->cap_rows, s));
        HB_HIP(hipMemsetAsync(ix->bnorm, 0, (size_t)ix->cap_rows * 4, s));
        HB_HIP(hipStreamSynchronize(s));
    }
    HB_HIP(hipMemsetAsync(ix->bmax, 0, 4, s));
    ix->ntotal = 0; ix->nlabels = 0; ix->lab_checked = 0; ix->f16_rows = 0; ix->f16_overflow = 0; ix->rows32_rows = 0;
    if (ix->lab_flag) HB_HIP(hipMemsetAsync(ix->lab_flag, 0, 4, s));
    if (ix->f16_flag) HB_HIP(hipMemsetAsync(ix->f16_flag, 0, 4, s));
    return 0;
}

static int stage_in(hb_index* ix, const void* host, size_t bytes, size_t offset) {
    HB_HIP(hipMemcpyAsync(ix->tmp + offset, host, bytes, hipMemcpyHostToDevice, ix->stream));
    return 0;
}

extern "C" int hb_index_add(hb_index_t* ix, const float* x, int64_t n, int x_on_device, int normalize) {
    if (!ix) return hb_fail("hb_index_add: NULL index handle");
    if (n < 0) return hb_fail("hb_index_add: negative row count");
    if (n == 0) return 0;
    if (!x) return hb_fail("hb_index_add: x is NULL");
    hb_range range("hbird:index_add");
    HB_HIP(hipSetDevice(ix->device));
    if (ix->ntotal + n > ix->cap_rows) {
        int64_t want = std::max<int64_t>(ix->ntotal + n, ix->cap_rows + ix->cap_rows / 2);
        if (hb_index_reserve(ix, want)) return -1;
    }
    const float* src = x;
    if (!x_on_device) {
        // host rows are staged in chunks of <= 256 MiB
        const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)ix->d * 4));
        if (grow((void**)&ix->tmp, &ix->tmp_bytes, (size_t)std::min(chunk, n) * ix->d * 4)) return -1;
        for (int64_t r = 0; r < n; r += chunk) {
            const int64_t m = std::min(chunk, n - r);
            if (stage_in(ix, x + r * (int64_t)ix->d, (size_t)m * ix->d * 4, 0)) return -1;
            if (hb_launch_rows_to_tiles((const float*)ix->tmp, m, ix->d, ix->dp, ix->ntotal + r, ix->tiles, ix->binit, ix->bnorm,
                                        ix->metric, normalize, 1, ix->stream)) return -1;
            HB_HIP(hipStreamSynchronize(ix->stream));
        }
    } else {
        // rows_to_tiles reads 32 source rows per block starting at row 0 of src; destination offset = ntotal
        if (hb_launch_rows_to_tiles(src, n, ix->d, ix->dp, ix->ntotal, ix->tiles, ix->binit, ix->bnorm, ix->metric,
                                    normalize, 1, ix->stream)) return -1;
    }
    if (hb_launch_bnorm_max(ix->bnorm + ix->ntotal, n, ix->bmax, ix->stream)) return -1;
    ix->ntotal += n;
    return 0;
}

extern "C" int hb_index_set_label_denominator(hb_index_t* ix, int P) {
    if (!ix) return hb_fail("hb_index_set_label_denominator: NULL index handle");
    if (P < 0 || P > 65535) return hb_fail("hb_index_set_label_denominator: the denominator must be in [0, 65535]");
    if (ix->nlabels > 0 && P != ix->label_P) return hb_fail("hb_index_set_label_denominator: the index already holds label rows");
    // fp32 values <-> uint16 counts: `lab_cap` describes the buffer of ONE form.  After hb_index_reset (which keeps allocations) a change
    // of form would leave it describing the other form's buffer -- a null or smaller one: drop both and start from no capacity.
    if ((P == 0) != (ix->label_P == 0) && (ix->labels || ix->labels16)) {
        HB_HIP(hipSetDevice(ix->device));
        HB_HIP(hipStreamSynchronize(ix->stream));
        if (ix->labels) HB_HIP(hipFree(ix->labels));
        if (ix->labels16) HB_HIP(hipFree(ix->labels16));
        ix->labels = nullptr; ix->labels16 = nullptr; ix->lab_cap = 0;
    }
    ix->label_P = P;
    return 0;
}
extern "C" int hb_index_labels_to_fp32(hb_index_t* ix) {
    if (!ix) return hb_fail("hb_index_labels_to_fp32: NULL index handle");
    if (ix->label_P == 0) return 0;
    HB_HIP(hipSetDevice(ix->device));
    if (hb_labels_checked(ix)) return -1;
    float* nl = nullptr;
    const int64_t cap = std::max<int64_t>(ix->nlabels, ix->lab_cap);
    if (ix->nlabels > 0) {
        HB_HIP(hipMalloc((void**)&nl, (size_t)cap * ix->c * 4));
        // one pass: count j of the value j / P -> (float)j / (float)P, the fp32 value K2 produced (hbird_eval.py:319-320)
        if (hb_launch_gather_label_counts(ix->labels16, ix->nlabels, ix->c, ix->lab_stride(), ix->label_P, nullptr, ix->nlabels, nl, ix->stream)) { (void)hipFree(nl); return -1; }
    }
    HB_HIP(hipStreamSynchronize(ix->stream));
    if (ix->labels16) HB_HIP(hipFree(ix->labels16));
    if (ix->labels) HB_HIP(hipFree(ix->labels));
    ix->labels16 = nullptr; ix->labels = nl; ix->label_P = 0; ix->lab_cap = nl ? cap : 0; ix->lab_checked = 0;
    return 0;
}
extern "C" int hb_index_label_denominator(const hb_index_t* ix, int* P) {
    if (!ix || !P) return hb_fail("hb_index_label_denominator: NULL pointer");
    *P = ix->label_P;
    return 0;
}

// one read-back (a stream synchronisation) after the label table grew: was every stored value a multiple of 1 / P?
int hb_labels_checked(hb_index* ix) {
    if (ix->label_P == 0 || ix->lab_checked >= ix->nlabels || !ix->lab_flag) return 0;
    int bad = 0;
    HB_HIP(hipMemcpyAsync(&bad, ix->lab_flag, 4, hipMemcpyDeviceToHost, ix->stream));
    HB_HIP(hipStreamSynchronize(ix->stream));
    if (bad) return hb_fail("label rows are not multiples of 1 / " + std::to_string(ix->label_P) +
                            " (hb_index_set_label_denominator): store them as fp32 (denominator 0) instead");
    ix->lab_checked = ix->nlabels;
    return 0;
}

extern "C" int hb_index_add_labels(hb_index_t* ix, const float* labels, int64_t n, int c, int on_device) {
    if (!ix) return hb_fail("hb_index_add_labels: NULL index handle");
    if (n < 0 || c <= 0) return hb_fail("hb_index_add_labels: bad shape");
    if (n == 0) return 0;
    HB_HIP(hipSetDevice(ix->device));
    if (ix->c != 0 && ix->c != c && ix->nlabels > 0) return hb_fail("hb_index_add_labels: class count changed");
    if (ix->c != c && ix->lab_cap > 0) {
        // an emptied index (hb_index_reset keeps allocations) takes rows of another width: `lab_cap` counts rows of the OLD width
        HB_HIP(hipStreamSynchronize(ix->stream));
        if (ix->labels) HB_HIP(hipFree(ix->labels));
        if (ix->labels16) HB_HIP(hipFree(ix->labels16));
        ix->labels = nullptr; ix->labels16 = nullptr; ix->lab_cap = 0;
    }
    ix->c = c;
    const size_t esz = ix->label_P ? 2 : 4;      // uint16 counts or fp32 values
    const size_t ls = (size_t)ix->lab_stride();  // elements per stored row (counts: padded to 16 bytes)
    if (ix->nlabels + n > ix->lab_cap) {
        int64_t cap = std::max<int64_t>(ix->nlabels + n, std::max<int64_t>(ix->cap_rows, ix->lab_cap + ix->lab_cap / 2));
        char* nl = nullptr;
        char* old = ix->label_P ? (char*)ix->labels16 : (char*)ix->labels;
        HB_HIP(hipMalloc((void**)&nl, (size_t)cap * ls * esz));
        if (ix->nlabels > 0) HB_HIP(hipMemcpyAsync(nl, old, (size_t)ix->nlabels * ls * esz, hipMemcpyDeviceToDevice, ix->stream));
        HB_HIP(hipStreamSynchronize(ix->stream));
        if (old) HB_HIP(hipFree(old));
        if (ix->label_P) ix->labels16 = (uint16_t*)nl; else ix->labels = (float*)nl;
        ix->lab_cap = cap;
    }
    if (ix->label_P) {
        // values j / P (what K2 produces, hbird_eval.py:319-320) stored as the uint16 count j: half the table, the same fp32 value back
        if (!ix->lab_flag) { HB_HIP(hipMalloc((void**)&ix->lab_flag, 4)); HB_HIP(hipMemsetAsync(ix->lab_flag, 0, 4, ix->stream)); }
        const float* src = labels;
        if (!on_device) {
            if (grow((void**)&ix->tmp, &ix->tmp_bytes, (size_t)n * c * 4)) return -1;
            if (stage_in(ix, labels, (size_t)n * c * 4, 0)) return -1;
            src = (const float*)ix->tmp;
        }
        if (hb_launch_labels_to_counts(src, n, c, (int)ls, ix->label_P, ix->labels16 + ix->nlabels * (int64_t)ls, ix->lab_flag, ix->stream)) return -1;
        if (!on_device) HB_HIP(hipStreamSynchronize(ix->stream));
        ix->nlabels += n;
        return 0;
    }
    HB_HIP(hipMemcpyAsync(ix->labels + ix->nlabels * (int64_t)c, labels, (size_t)n * c * 4,
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ix->stream));
    if (!on_device) HB_HIP(hipStreamSynchronize(ix->stream));
    ix->nlabels += n;
    return 0;
}

static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

static int search_impl(hb_index* ix, const float* q, int64_t nq, int k, int64_t id_base, float beta, float* out_lab,
                       int64_t* out_idx, float* out_dist, int io_on_device, bool aggregate) {
    if (nq < 0) return hb_fail("hb_index_search: negative query count");
    if (k < 1 || k > HB_MAX_K) return hb_fail("hb_index_search: k must be in [1, " + std::to_string(HB_MAX_K) + "] (faiss-gpu's own limit)");
    if (aggregate && k > HB_MAX_K_AGGREGATE)
        return hb_fail("hb_index_search_aggregate: k must be in [1, " + std::to_string(HB_MAX_K_AGGREGATE) + "] (plain searches take k up to " + std::to_string(HB_MAX_K) + ")");
    if (nq == 0) return 0;
    if (!q) return hb_fail("hb_index_search: q is NULL");
    hb_range range(aggregate ? "hbird:search_aggregate" : "hbird:search");
    HB_HIP(hipSetDevice(ix->device));
    const int64_t nqp = (nq + HB_QT - 1) / HB_QT * HB_QT;
    if (grow((void**)&ix->q_tiles, &ix->q_tiles_bytes, (size_t)nqp * ix->dp * 4)) return -1;
    if (grow((void**)&ix->q_aux, &ix->q_aux_bytes, (size_t)nq * 2 * 4)) return -1;
    // staging area in ix->tmp: [queries (host path)] [idx] [dist] [label_hat (host path)]
    const size_t b_q = io_on_device ? 0 : al256((size_t)nq * ix->d * 4);
    const size_t b_idx = al256((size_t)nq * k * 8), b_dist = al256((size_t)nq * k * 4);
    const size_t b_lab = (aggregate && !io_on_device) ? al256((size_t)nq * ix->c * 4) : 0;
    const bool t_idx = !io_on_device || !out_idx, t_dist = !io_on_device || !out_dist;
    const size_t need = b_q + (t_idx ? b_idx : 0) + (t_dist ? b_dist : 0) + b_lab;
    if (need && grow((void**)&ix->tmp, &ix->tmp_bytes, need)) return -1;
    char* cur = ix->tmp;
    const float* qd = q;
    if (!io_on_device) {
        HB_HIP(hipMemcpyAsync(cur, q, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice, ix->stream));
        qd = (const float*)cur;
        cur += b_q;
    }
    int64_t* d_idx = out_idx;
    float* d_dist = out_dist;
    float* d_lab = out_lab;
    if (t_idx) { d_idx = (int64_t*)cur; cur += b_idx; }
    if (t_dist) { d_dist = (float*)cur; cur += b_dist; }
    if (b_lab) { d_lab = (float*)cur; cur += b_lab; }
    {
        hb_range r("hbird:query_tiles");
        if (hb_launch_rows_to_tiles(qd, nq, ix->d, ix->dp, 0, ix->q_tiles, nullptr, nullptr, ix->metric, 0, 0, ix->stream)) return -1;
        if (hb_launch_query_aux(qd, nq, ix->d, ix->q_aux, ix->q_aux + nq, ix->stream)) return -1;
    }
    {
        hb_range r("hbird:knn");
        if (hb_launch_knn(ix, qd, nq, k, id_base, d_idx, d_dist)) return -1;
    }
    if (aggregate) {
        hb_range r("hbird:aggregate");
        if (hb_launch_aggregate(ix, ix->q_aux + nq, d_idx, d_dist, nq, k, id_base, beta, d_lab, ix->stream)) return -1;
    }
    if (!io_on_device) {
        if (out_idx) HB_HIP(hipMemcpyAsync(out_idx, d_idx, (size_t)nq * k * 8, hipMemcpyDeviceToHost, ix->stream));
        if (out_dist) HB_HIP(hipMemcpyAsync(out_dist, d_dist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ix->stream));
        if (aggregate) HB_HIP(hipMemcpyAsync(out_lab, d_lab, (size_t)nq * ix->c * 4, hipMemcpyDeviceToHost, ix->stream));
        HB_HIP(hipStreamSynchronize(ix->stream));
    }
    return 0;
}

extern "C" int hb_index_search(hb_index_t* ix, const float* q, int64_t nq, int k, int64_t id_base, int64_t* out_idx,
                               float* out_dist, int io_on_device) {
    if (!ix) return hb_fail("hb_index_search: NULL index handle");
    if (nq > 0 && (!out_idx || !out_dist)) return hb_fail("hb_index_search: output pointers are NULL");
    return search_impl(ix, q, nq, k, id_base, 0.f, nullptr, out_idx, out_dist, io_on_device, false);
}

extern "C" int hb_index_search_aggregate(hb_index_t* ix, const float* q, int64_t nq, int k, int64_t id_base, float beta,
                                         float* out_label_hat, int64_t* out_idx_opt, float* out_dist_opt,
                                         int io_on_device) {
    if (!ix) return hb_fail("hb_index_search_aggregate: NULL index handle");
    if (nq > 0 && !out_label_hat) return hb_fail("hb_index_search_aggregate: out_label_hat is NULL");
    if (!(beta > 0.f)) return hb_fail("hb_index_search_aggregate: beta must be positive");
    if (!ix->ext_labels && !ix->ext_labels16 && ((!ix->labels && !ix->labels16) || ix->nlabels < ix->ntotal)) return hb_fail("hb_index_search_aggregate: label rows missing (hb_index_add_labels)");
    if (hb_labels_checked(ix)) return -1;
    return search_impl(ix, q, nq, k, id_base, beta, out_label_hat, out_idx_opt, out_dist_opt, io_on_device, true);
}

extern "C" int hb_index_aggregate(hb_index_t* ix, const float* q, int64_t nq, const int64_t* idx, const float* dist,
                                  int k, int64_t id_base, float beta, float* out_label_hat, int io_on_device) {
    if (!ix) return hb_fail("hb_index_aggregate: NULL index handle");
    if (nq == 0) return 0;
    if (!io_on_device) return hb_fail("hb_index_aggregate: host pointers are not supported, pass device memory");
    if (!(beta > 0.f)) return hb_fail("hb_index_aggregate: beta must be positive");
    hb_range range("hbird:aggregate");
    HB_HIP(hipSetDevice(ix->device));
    if (hb_labels_checked(ix)) return -1;
    if (grow((void**)&ix->q_aux, &ix->q_aux_bytes, (size_t)nq * 2 * 4)) return -1;
    if (hb_launch_query_aux(q, nq, ix->d, ix->q_aux, ix->q_aux + nq, ix->stream)) return -1;
    return hb_launch_aggregate(ix, ix->q_aux + nq, idx, dist, nq, k, id_base, beta, out_label_hat, ix->stream);
}

extern "C" int hb_index_aggregate_partial(hb_index_t* ix, const float* q, int64_t nq, const int64_t* idx, const float* dist, int k,
                                          int64_t id_base, float beta, const float* norms_all, int64_t n_all, float* out_partial) {
    if (!ix) return hb_fail("hb_index_aggregate_partial: NULL index handle");
    if (nq == 0) return 0;
    if (!q || !idx || !dist || !norms_all || !out_partial) return hb_fail("hb_index_aggregate_partial: NULL pointer");
    if (!(beta > 0.f)) return hb_fail("hb_index_aggregate_partial: beta must be positive");
    hb_range range("hbird:aggregate_partial");
    HB_HIP(hipSetDevice(ix->device));
    if (hb_labels_checked(ix)) return -1;
    if (ix->ntotal == 0) {   // an empty shard owns no neighbour: its partial sums are zero
        HB_HIP(hipMemsetAsync(out_partial, 0, (size_t)nq * ix->c * 4, ix->stream));
        return 0;
    }
    if (grow((void**)&ix->q_aux, &ix->q_aux_bytes, (size_t)nq * 2 * 4)) return -1;
    if (hb_launch_query_aux(q, nq, ix->d, ix->q_aux, ix->q_aux + nq, ix->stream)) return -1;
    return hb_launch_aggregate(ix, ix->q_aux + nq, idx, dist, nq, k, id_base, beta, out_partial, ix->stream, norms_all, n_all);
}

static int gather_impl(hb_index* ix, const int64_t* ids, int64_t n, int64_t id_base, float* out, int io_on_device,
                       bool labels) {
    if (n == 0) return 0;
    HB_HIP(hipSetDevice(ix->device));
    const int width = labels ? ix->c : ix->d;
    if (labels && !ix->labels && !ix->labels16) return hb_fail("hb_index_gather_labels: no labels stored");
    if (labels && hb_labels_checked(ix)) return -1;
    const int64_t* d_ids = ids;
    float* d_out = out;
    if (!io_on_device) {
        const size_t b_ids = ((size_t)n * 8 + 255) / 256 * 256;
        if (grow((void**)&ix->tmp, &ix->tmp_bytes, b_ids + (size_t)n * width * 4)) return -1;
        if (stage_in(ix, ids, (size_t)n * 8, 0)) return -1;
        d_ids = (const int64_t*)ix->tmp;
        d_out = (float*)(ix->tmp + b_ids);
    }
    if (labels) {
        // shift global ids to local rows inside the kernel via src offset: ids are global, rows local
        if (id_base != 0) return hb_fail("hb_index_gather_labels: id_base != 0 is not supported yet");
        if (ix->label_P ? hb_launch_gather_label_counts(ix->labels16, ix->nlabels, width, ix->lab_stride(), ix->label_P, d_ids, n, d_out, ix->stream)
                        : hb_launch_gather_rows(ix->labels, ix->nlabels, width, d_ids, n, d_out, ix->stream)) return -1;
    } else {
        if (hb_launch_tiles_to_rows(ix->tiles, ix->g8, ix->d, d_ids, n, id_base, d_out, ix->stream)) return -1;
    }
    if (!io_on_device) {
        HB_HIP(hipMemcpyAsync(out, d_out, (size_t)n * width * 4, hipMemcpyDeviceToHost, ix->stream));
        HB_HIP(hipStreamSynchronize(ix->stream));
    }
    return 0;
}

extern "C" int hb_index_reconstruct(hb_index_t* ix, const int64_t* ids, int64_t n, int64_t id_base, float* out,
                                    int io_on_device) {
    if (!ix) return hb_fail("hb_index_reconstruct: NULL index handle");
    return gather_impl(ix, ids, n, id_base, out, io_on_device, false);
}
extern "C" int hb_index_gather_labels(hb_index_t* ix, const int64_t* ids, int64_t n, int64_t id_base, float* out,
                                      int io_on_device) {
    if (!ix) return hb_fail("hb_index_gather_labels: NULL index handle");
    return gather_impl(ix, ids, n, id_base, out, io_on_device, true);
}

extern "C" int hb_index_set_label_table(hb_index_t* ix, const float* labels, const float* bnorm, int64_t n, int c,
                                        int64_t id_base) {
    if (!ix) return hb_fail("hb_index_set_label_table: NULL index handle");
    if (labels && (!bnorm || n <= 0 || c <= 0)) return hb_fail("hb_index_set_label_table: bad arguments");
    ix->ext_labels = labels; ix->ext_labels16 = nullptr; ix->ext_P = 0;
    ix->ext_bnorm = bnorm; ix->ext_n = labels ? n : 0; ix->ext_base = labels ? id_base : 0;
    if (labels) ix->c = c;
    return 0;
}

extern "C" int hb_index_set_label_count_table(hb_index_t* ix, const uint16_t* counts, const float* bnorm, int64_t n, int c, int P,
                                              int64_t id_base) {
    if (!ix) return hb_fail("hb_index_set_label_count_table: NULL index handle");
    if (counts && (!bnorm || n <= 0 || c <= 0 || P <= 0 || P > 65535)) return hb_fail("hb_index_set_label_count_table: bad arguments");
    ix->ext_labels = nullptr; ix->ext_labels16 = counts; ix->ext_P = counts ? P : 0;
    ix->ext_bnorm = bnorm; ix->ext_n = counts ? n : 0; ix->ext_base = counts ? id_base : 0;
    if (counts) ix->c = c;
    return 0;
}

extern "C" int hb_index_copy_label_counts(hb_index_t* ix, uint16_t* out, int on_device) {
    if (!ix) return hb_fail("hb_index_copy_label_counts: NULL index handle");
    if (ix->label_P == 0) return hb_fail("hb_index_copy_label_counts: the labels are stored as fp32 (hb_index_set_label_denominator)");
    if (ix->nlabels == 0) return 0;
    HB_HIP(hipSetDevice(ix->device));
    if (hb_labels_checked(ix)) return -1;
    // (the stored rows are padded to 16 bytes; the caller gets dense [nlabels, C] rows)
    HB_HIP(hipMemcpy2DAsync(out, (size_t)ix->c * 2, ix->labels16, (size_t)ix->lab_stride() * 2, (size_t)ix->c * 2, (size_t)ix->nlabels,
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ix->stream));
    if (!on_device) HB_HIP(hipStreamSynchronize(ix->stream));
    return 0;
}

extern "C" int hb_index_copy_norms(hb_index_t* ix, float* out, int on_device) {
    if (!ix) return hb_fail("hb_index_copy_norms: NULL index handle");
    if (ix->ntotal == 0) return 0;
    HB_HIP(hipSetDevice(ix->device));
    HB_HIP(hipMemcpyAsync(out, ix->bnorm, (size_t)ix->ntotal * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ix->stream));
    if (!on_device) HB_HIP(hipStreamSynchronize(ix->stream));
    return 0;
}

extern "C" int hb_index_set_score_output(hb_index_t* ix, int enable) {
    if (!ix) return hb_fail("hb_index_set_score_output: NULL index handle");
    ix->score_output = enable ? 1 : 0;
    return 0;
}

extern "C" int hb_index_distances_from_scores(hb_index_t* ix, const float* q, int64_t nq, int k, float* dist_inout) {
    if (!ix) return hb_fail("hb_index_distances_from_scores: NULL index handle");
    if (nq == 0 || ix->metric != 1) return 0;            // inner product: the score is the distance
    if (!q || !dist_inout) return hb_fail("hb_index_distances_from_scores: NULL pointer");
    HB_HIP(hipSetDevice(ix->device));
    if (grow((void**)&ix->q_aux, &ix->q_aux_bytes, (size_t)nq * 2 * 4)) return -1;
    if (hb_launch_query_aux(q, nq, ix->d, ix->q_aux, ix->q_aux + nq, ix->stream)) return -1;
    return hb_launch_scores_to_l2(ix->q_aux, nq, k, dist_inout, ix->stream);
}

extern "C" int hb_merge_topk(const float* dist_parts, const int64_t* idx_parts, int parts, int64_t nq, int k, int metric,
                             int64_t* out_idx, float* out_dist, void* stream) {
    hb_range range("hbird:merge_topk");
    if (parts < 1 || k < 1) return hb_fail("hb_merge_topk: bad shape");
    if (nq > 0 && (!dist_parts || !idx_parts || !out_idx || !out_dist)) return hb_fail("hb_merge_topk: NULL pointer");
    return hb_launch_merge_parts(dist_parts, idx_parts, parts, nq, k, metric, nq * (int64_t)k, nq * (int64_t)k, out_idx, out_dist,
                                 (hipStream_t)stream);
}

extern "C" int64_t hb_packed_list_bytes(int64_t nq, int k) {
    const int64_t n = nq * (int64_t)k;
    return (n * 12 + 15) / 16 * 16;
}

extern "C" int hb_merge_topk_packed(const void* packed_parts, int64_t part_bytes, int parts, int64_t nq, int k, int metric,
                                    int64_t* out_idx, float* out_dist, void* stream) {
    hb_range range("hbird:merge_topk");
    if (parts < 1 || k < 1) return hb_fail("hb_merge_topk_packed: bad shape");
    if (nq == 0) return 0;
    if (!packed_parts || !out_idx || !out_dist) return hb_fail("hb_merge_topk_packed: NULL pointer");
    if (part_bytes < nq * (int64_t)k * 12 || part_bytes % 8 != 0)
        return hb_fail("hb_merge_topk_packed: part_bytes must be a multiple of 8 and hold nq*k ids (int64) + nq*k scores (fp32)");
    const char* base = reinterpret_cast<const char*>(packed_parts);
    return hb_launch_merge_parts(reinterpret_cast<const float*>(base + nq * (int64_t)k * 8), reinterpret_cast<const int64_t*>(base),
                                 parts, nq, k, metric, part_bytes / 4, part_bytes / 8, out_idx, out_dist, (hipStream_t)stream);
}

extern "C" int hb_normalize_rows(const float* x, int64_t n, int d, float* out, void* stream) {
    return hb_launch_normalize_rows(x, n, d, out, (hipStream_t)stream);
}
extern "C" int hb_patch_label_hist(const int64_t* y, int64_t B, int H, int W, int ps, int C, int map255, float* out,
                                   void* stream) {
    hb_range range("hbird:patch_label_hist");
    return hb_launch_patch_label_hist(y, B, H, W, ps, C, map255, out, (hipStream_t)stream);
}
extern "C" int hb_gather_rows(const float* src, int64_t src_rows, int width, const int64_t* ids, int64_t n, float* out,
                              void* stream) {
    return hb_launch_gather_rows(src, src_rows, width, ids, n, out, (hipStream_t)stream);
}
extern "C" int hb_upsample_argmax(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out,
                                  void* stream) {
    hb_range range("hbird:upsample_argmax");
    return hb_launch_upsample_argmax(label_hat, B, S, C, h, w, out, (hipStream_t)stream);
}
extern "C" int hb_upsample_argmax_confusion(const float* label_hat, int64_t B, int S, int C, int h, int w, const int64_t* gt, int num_gt,
                                            int num_pred, int64_t ignore_index, int has_ignore, uint64_t* conf, int64_t* out_map_opt,
                                            void* stream) {
    hb_range range("hbird:upsample_argmax_confusion");
    if (!gt || !conf) return hb_fail("hb_upsample_argmax_confusion: gt / conf is NULL");
    return hb_launch_upsample_argmax_confusion(label_hat, B, S, C, h, w, out_map_opt, gt, num_gt, num_pred, ignore_index, has_ignore,
                                               reinterpret_cast<unsigned long long*>(conf), (hipStream_t)stream);
}
extern "C" int hb_upsample_accumulate(const float* label_hat, int64_t B, int S, int C, int win_h, int win_w, float* acc,
                                      int H, int W, int y0, int x0, void* stream) {
    hb_range range("hbird:upsample_accumulate");
    return hb_launch_upsample_accumulate(label_hat, B, S, C, win_h, win_w, acc, H, W, y0, x0, (hipStream_t)stream);
}
extern "C" int hb_argmax_channels(const float* acc, int64_t n, int C, int64_t* out, void* stream) {
    return hb_launch_argmax_channels(acc, n, C, out, (hipStream_t)stream);
}
extern "C" int hb_confusion_update(const int64_t* gt, const int64_t* pred, int64_t n, int num_gt, int num_pred,
                                   int64_t ignore_index, int has_ignore, uint64_t* conf, void* stream) {
    hb_range range("hbird:confusion_update");
    return hb_launch_confusion(gt, pred, n, num_gt, num_pred, ignore_index, has_ignore, (unsigned long long*)conf,
                               (hipStream_t)stream);
}
