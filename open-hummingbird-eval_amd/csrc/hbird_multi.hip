// hb_multi_*: several GPUs behind ONE handle of the C ABI -- what the reference's single-process call builds with
// faiss.index_cpu_to_gpu_multiple_py: faiss.IndexShards (search_faiss.py:53-63; `shard` = 1: contiguous row ranges with
// successive ids, every GPU searches all queries, k-way merge) or faiss.IndexReplicas (65-74; `shard` = 0: every GPU holds
// all rows, the queries are split), one host thread per GPU (faiss `threaded = True`, 57).
//
// Host code only: a composition of the single-GPU entry points (hb_index_*), so that a host that binds libhbird_hip.so
// without Python gets shards / replicas without redoing them (in Python the same composition is
// hbird_mi.nn.search_hip.HipMultiIndex).  Buffers are host memory, like the numpy arrays of the reference's plugin.
// Shards return their ORDERING scores (hb_index_set_score_output): the lists are merged here by (score descending, id
// ascending) and converted to the metric's distances exactly as the single-index search converts them, so the result is the
// single-index result bit for bit.
#include "../../include/hbird_hip.h"
#include "hbird_internal.h"
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

struct hb_multi {
    int d = 0, metric = 0, shard = 0;
    std::vector<hb_index_t*> ix;
    int64_t quota = 0;   // shard mode: planned rows per shard (hb_multi_reserve); 0 = everything into the first
};

namespace {
// run fn(i) for every index on its own host thread; the first failure's message becomes the caller's hb_last_error
template <class F>
int for_each_gpu(const hb_multi* m, F fn) {
    const int n = (int)m->ix.size();
    std::vector<int> rc(n, 0);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    th.reserve(n);
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            rc[i] = fn(i);
            if (rc[i]) err[i] = hb_last_error();   // thread-local: copy it out before the thread ends
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (rc[i]) return hb_fail("hb_multi (GPU entry " + std::to_string(i) + "): " + err[i]);
    return 0;
}
}  // namespace

extern "C" int hb_multi_create(int d, int metric, const int* gpu_ids, int n_gpus, int shard, hb_multi_t** out) {
    if (!out) return hb_fail("hb_multi_create: out is NULL");
    *out = nullptr;
    if (n_gpus < 1 || !gpu_ids) return hb_fail("hb_multi_create: at least one GPU id is needed");
    hb_multi* m = new hb_multi();
    m->d = d; m->metric = metric; m->shard = shard ? 1 : 0;
    for (int i = 0; i < n_gpus; ++i) {
        hb_index_t* h = nullptr;
        if (hb_index_create(d, metric, gpu_ids[i], &h)) {       // validates d, metric and the GPU id (search_faiss.py:23-25)
            const std::string msg = hb_last_error();
            for (hb_index_t* p : m->ix) hb_index_free(p);
            delete m;
            return hb_fail(msg);
        }
        m->ix.push_back(h);
    }
    *out = m;
    return 0;
}

extern "C" int hb_multi_free(hb_multi_t* m) {
    if (!m) return 0;
    for (hb_index_t* p : m->ix) hb_index_free(p);
    delete m;
    return 0;
}

extern "C" int64_t hb_multi_ntotal(const hb_multi_t* m) {
    if (!m) { hb_set_error("hb_multi_ntotal: NULL handle"); return -1; }
    if (!m->shard) return hb_index_ntotal(m->ix[0]);
    int64_t n = 0;
    for (hb_index_t* p : m->ix) n += hb_index_ntotal(p);
    return n;
}

extern "C" int hb_multi_shard_rows(const hb_multi_t* m, int64_t* rows, int n) {
    if (!m || !rows) return hb_fail("hb_multi_shard_rows: NULL pointer");
    for (int i = 0; i < n && i < (int)m->ix.size(); ++i) rows[i] = hb_index_ntotal(m->ix[i]);
    return 0;
}

extern "C" int hb_multi_set_fp16(hb_multi_t* m, int enable) {
    if (!m) return hb_fail("hb_multi_set_fp16: NULL handle");
    for (hb_index_t* p : m->ix) if (hb_index_set_fp16(p, enable)) return -1;
    return 0;
}

extern "C" int hb_multi_reserve(hb_multi_t* m, int64_t n_rows) {
    if (!m) return hb_fail("hb_multi_reserve: NULL handle");
    const int n = (int)m->ix.size();
    const int64_t per = m->shard ? (std::max<int64_t>(n_rows, 1) + n - 1) / n : std::max<int64_t>(n_rows, 1);
    if (m->shard) m->quota = per;
    return for_each_gpu(m, [&](int i) { return hb_index_reserve(m->ix[i], per); });
}

extern "C" int hb_multi_add(hb_multi_t* m, const float* x, int64_t n, int normalize) {
    if (!m) return hb_fail("hb_multi_add: NULL handle");
    if (n < 0) return hb_fail("hb_multi_add: negative row count");
    if (n == 0) return 0;
    if (!x) return hb_fail("hb_multi_add: x is NULL");
    const int ng = (int)m->ix.size();
    if (!m->shard) return for_each_gpu(m, [&](int i) { return hb_index_add(m->ix[i], x, n, 0, normalize); });
    // shards fill up one after the other (successive ids, faiss.IndexShards' successive_ids); the last one also takes
    // whatever arrives beyond the plan
    std::vector<int64_t> lo(ng, 0), cnt(ng, 0);
    int64_t done = 0;
    for (int i = 0; i < ng && done < n; ++i) {
        int64_t room = n - done;
        if (i < ng - 1) {
            if (m->quota <= 0) room = i == 0 ? room : 0;
            else room = std::min<int64_t>(room, std::max<int64_t>(0, m->quota - hb_index_ntotal(m->ix[i])));
        }
        lo[i] = done; cnt[i] = room; done += room;
    }
    return for_each_gpu(m, [&](int i) { return cnt[i] ? hb_index_add(m->ix[i], x + lo[i] * (int64_t)m->d, cnt[i], 0, normalize) : 0; });
}

extern "C" int hb_multi_search(hb_multi_t* m, const float* q, int64_t nq, int k, int64_t* out_idx, float* out_dist) {
    if (!m) return hb_fail("hb_multi_search: NULL handle");
    if (nq < 0) return hb_fail("hb_multi_search: negative query count");
    if (k < 1 || k > HB_MAX_K_AGGREGATE)
        return hb_fail("hb_multi_search: k must be in [1, " + std::to_string(HB_MAX_K_AGGREGATE) + "] (the merge of the GPUs' lists; a single-GPU index takes k up to " + std::to_string(HB_MAX_K) + ")");
    if (nq == 0) return 0;
    if (!q || !out_idx || !out_dist) return hb_fail("hb_multi_search: NULL pointer");
    const int ng = (int)m->ix.size();
    if (!m->shard) {   // replicas: a slice of the queries each, results land in place
        return for_each_gpu(m, [&](int i) {
            const int64_t a = nq * i / ng, b = nq * (i + 1) / ng;
            return hb_index_search(m->ix[i], q + a * (int64_t)m->d, b - a, k, 0, out_idx + a * k, out_dist + a * k, 0);
        });
    }
    std::vector<int64_t> base(ng, 0);
    for (int i = 1; i < ng; ++i) base[i] = base[i - 1] + hb_index_ntotal(m->ix[i - 1]);
    std::vector<std::vector<int64_t>> pi(ng, std::vector<int64_t>((size_t)nq * k));
    std::vector<std::vector<float>> ps(ng, std::vector<float>((size_t)nq * k));
    if (for_each_gpu(m, [&](int i) {
            if (hb_index_set_score_output(m->ix[i], 1)) return -1;
            const int rc = hb_index_search(m->ix[i], q, nq, k, base[i], pi[i].data(), ps[i].data(), 0);
            hb_index_set_score_output(m->ix[i], 0);
            return rc;
        })) return -1;
    // k-way merge of the shards' sorted lists by (score descending, id ascending); missing entries (id -1) sort last.  Rows are
    // independent: a few host threads share them (21,904 x 30 x 8 shards is 5 M steps per search)
    auto merge_rows = [&](int64_t r0, int64_t r1) {
        std::vector<int> pos(ng);
        for (int64_t r = r0; r < r1; ++r) {
            std::fill(pos.begin(), pos.end(), 0);
            float qn2 = 0.0f;   // chain ||q||^2 of the L2 conversion (hbird_layout.hip: query_aux_kernel)
            if (m->metric == HB_METRIC_L2)
                for (int c = 0; c < m->d; ++c) qn2 = std::fmaf(q[r * (int64_t)m->d + c], q[r * (int64_t)m->d + c], qn2);
            for (int j = 0; j < k; ++j) {
                int best = -1;
                for (int i = 0; i < ng; ++i) {
                    if (pos[i] >= k || pi[i][r * k + pos[i]] < 0) continue;
                    if (best < 0) { best = i; continue; }
                    const float s = ps[i][r * k + pos[i]], sb = ps[best][r * k + pos[best]];
                    if (s > sb || (s == sb && pi[i][r * k + pos[i]] < pi[best][r * k + pos[best]])) best = i;
                }
                const int64_t o = r * k + j;
                if (best < 0) { out_idx[o] = -1; out_dist[o] = m->metric == HB_METRIC_L2 ? INFINITY : -INFINITY; continue; }
                const float s = ps[best][r * k + pos[best]];
                out_idx[o] = pi[best][r * k + pos[best]];
                ++pos[best];
                if (m->metric == HB_METRIC_L2) { const float d2 = std::fmaf(-2.0f, s, qn2); out_dist[o] = d2 > 0.0f ? d2 : 0.0f; }
                else out_dist[o] = s;
            }
        }
    };
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), (int64_t)16, nq / 256}));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(merge_rows, nq * t / nt, nq * (t + 1) / nt);
    merge_rows(0, nq / nt);
    for (auto& t : th) t.join();
    return 0;
}
