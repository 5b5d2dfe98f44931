// K4 of SURVEY.md 2.3: exact brute-force top-k of Q . Bank^T (inner product) or ||q-b||^2 (L2) --
// what the reference asks of faiss.GpuIndexFlatIP / GpuIndexFlatL2 (hbird/nn/search_faiss.py:43-46,
// 84-90) -- as ONE persistent gfx950 kernel: fp32 MFMA (v_mfma_f32_32x32x2_f32) contraction over
// LDS-staged fragment tiles with a fused per-query running top-k; the distance matrix is never
// written to HBM.  A second small kernel merges the per-workgroup partial lists.
//
// Work decomposition (host-built work list, see hb_build_schedule):
//   pair = (query tile of 256 rows, bank tile of 256 rows); a workgroup (512 threads = 8 waves,
//   2 per SIMD, one workgroup per CU) walks segments of consecutive bank tiles for one query tile
//   and keeps that query tile's k-best lists in LDS.  Wave w owns query columns [32w, 32w+32) and
//   all 256 bank rows of the tile: acc[8] x f32x16 = 128 accumulator VGPRs, C[m = bank row][n = query]
//   so a lane's 128 scores all belong to ONE query (lane & 31) and are compared against one
//   register-resident threshold.
//
// Numerics: every score is a single k-ascending fp32 fmaf chain started from the bank row's init
// value (0 for IP, -0.5*||b||^2 for L2, -inf for padding rows) -- bit-exact against
// oracle/hbird_oracle.c:orc_knn_chain_f32.  Ordering key: (score descending, row id ascending).
#include "hbird_internal.h"
#include <algorithm>
#include <map>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

struct knn_args {
    const float* bank_tiles;
    const float* binit;
    const float* q_tiles;
    const hb_seg* segs;
    const int* wg_off;
    float* state_s;
    unsigned* state_i;
    int g8;   // Dp / 8
    int k;
};

__device__ __forceinline__ void glds16(const float* gsrc, char* lds_base) {
    // 64 lanes x 16 B: per-lane global source, LDS destination = wave-uniform base + 16*lane
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)gsrc, (lds_void*)lds_base, 16, 0, 0);
}

// Wave-cooperative insertion of candidate (s, id) into the sorted list of local query ql
// (lanes 0..k-1 each hold one entry; the list stays sorted by (score desc, id asc)).
__device__ __forceinline__ void list_insert(float* lst_s, unsigned* lst_i, int ql, int k, float s, unsigned id,
                                            int lane) {
    const int e = lane & 31;
    const float es = lst_s[ql * HB_KL + e];
    const unsigned ei = lst_i[ql * HB_KL + e];
    const bool better = (es > s) || (es == s && ei < id);
    const unsigned long long kmask = (k >= 32) ? 0xFFFFFFFFull : ((1ull << k) - 1ull);
    const int p = __popcll(__ballot(better) & kmask);   // entries 0..p-1 beat the candidate
    if (p >= k) return;                                  // wave-uniform: not among the k best
    if (lane >= p && lane < k - 1) { lst_s[ql * HB_KL + lane + 1] = es; lst_i[ql * HB_KL + lane + 1] = ei; }
    if (lane == p) { lst_s[ql * HB_KL + p] = s; lst_i[ql * HB_KL + p] = id; }
}

#define HB_DUMP_CASE(T, H)                                                                     \
    case (2 * (T) + (H)):                                                                      \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) sc[r * 64 + lane] = acc[T][8 * (H) + r]; \
        break;

__global__ __launch_bounds__(HB_THREADS, 2) void knn_fused_kernel(knn_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    float* lst_s = reinterpret_cast<float*>(smem + HB_LDS_LISTS);
    unsigned* lst_i = reinterpret_cast<unsigned*>(smem + HB_LDS_LISTS + HB_QT * HB_KL * 4);
    float* sc = reinterpret_cast<float*>(smem + HB_LDS_SCRATCH) + w * 512;
    const int g8 = a.g8, k = a.k;
    const int NS = g8 >> 1;   // stages per bank tile
    const int myq = w * 32 + (lane & 31);

    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_off[blockIdx.x + 1];
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = a.segs[si];
        // ---- load (or start) this query tile's partial lists; each wave owns its 32 queries ----
        {
            float* gs = a.state_s + (size_t)seg.slot * (HB_QT * HB_KL) + w * 1024;
            unsigned* gi = a.state_i + (size_t)seg.slot * (HB_QT * HB_KL) + w * 1024;
            for (int e = lane; e < 1024; e += 64) {
                lst_s[w * 1024 + e] = seg.first ? -INFINITY : gs[e];
                lst_i[w * 1024 + e] = seg.first ? HB_ID_NONE : gi[e];
            }
        }
        float thr = lst_s[myq * HB_KL + (k - 1)];

        const float* qsrc = a.q_tiles + ((size_t)(seg.q_tile * 8 + w) * g8) * HB_BLK + lane * 4;
        const int total = seg.n_tiles * NS;
        f32x16 acc[8];

        // stage issue: wave w stages bank row-tile w and query row-tile w (2 KiB each), wave 0 also
        // the 256 row-init values at the first stage of a bank tile
        auto issue = [&](int bt, int ks, int buf) {
            char* sb = smem + buf * HB_STAGE_BYTES;
            const float* bsrc = a.bank_tiles + ((size_t)(bt * 8 + w) * g8 + ks * 2) * HB_BLK + lane * 4;
            glds16(bsrc, sb + (w * 2) * 1024);
            glds16(bsrc + HB_BLK, sb + (w * 2 + 1) * 1024);
            const float* qs = qsrc + (size_t)(ks * 2) * HB_BLK;
            glds16(qs, sb + 16384 + (w * 2) * 1024);
            glds16(qs + HB_BLK, sb + 16384 + (w * 2 + 1) * 1024);
            if (ks == 0 && w == 0) glds16(a.binit + (size_t)bt * HB_BT + lane * 4, sb + 32768);
        };

        int bt = seg.b_tile0, ks = 0;         // stage being computed
        int nbt_ = seg.b_tile0, nks = 0;      // stage being fetched
        issue(nbt_, nks, 0);
        for (int st = 0; st < total; ++st) {
            __syncthreads();   // stage st landed everywhere; everyone is done with the other buffer
            if (st + 1 < total) {
                if (++nks == NS) { nks = 0; ++nbt_; }
                issue(nbt_, nks, (st + 1) & 1);
            }
            const char* sb = smem + (st & 1) * HB_STAGE_BYTES;
            if (ks == 0) {
                const f32x4* bi = reinterpret_cast<const f32x4*>(sb + 32768);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = bi[8 * t + 2 * g + h];
                        acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1];
                        acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
                    }
            }
            const f32x4* A = reinterpret_cast<const f32x4*>(sb);
            const f32x4* B = reinterpret_cast<const f32x4*>(sb + 16384);
#pragma unroll
            for (int gl = 0; gl < 2; ++gl) {
                const f32x4 b = B[(w * 2 + gl) * 64 + lane];
                f32x4 af[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) af[t] = A[(t * 2 + gl) * 64 + lane];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t][s], b[s], acc[t], 0, 0, 0);
            }
            if (++ks == NS) {
                // ---- epilogue: filter the 256x32 score tile of this wave against the thresholds ----
                unsigned hmask = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    bool any0 = false, any1 = false;
#pragma unroll
                    for (int r = 0; r < 8; ++r) { any0 |= acc[t][r] > thr; any1 |= acc[t][8 + r] > thr; }
                    if (__ballot(any0) != 0ull) hmask |= 1u << (2 * t);
                    if (__ballot(any1) != 0ull) hmask |= 1u << (2 * t + 1);
                }
                while (hmask) {   // wave-uniform slow path: some score beat its query's k-th best
                    const int th = __builtin_ctz(hmask);
                    hmask &= hmask - 1;
                    switch (th) {
                        HB_DUMP_CASE(0, 0) HB_DUMP_CASE(0, 1) HB_DUMP_CASE(1, 0) HB_DUMP_CASE(1, 1)
                        HB_DUMP_CASE(2, 0) HB_DUMP_CASE(2, 1) HB_DUMP_CASE(3, 0) HB_DUMP_CASE(3, 1)
                        HB_DUMP_CASE(4, 0) HB_DUMP_CASE(4, 1) HB_DUMP_CASE(5, 0) HB_DUMP_CASE(5, 1)
                        HB_DUMP_CASE(6, 0) HB_DUMP_CASE(6, 1) HB_DUMP_CASE(7, 0) HB_DUMP_CASE(7, 1)
                    }
                    const unsigned row_base = (unsigned)bt * HB_BT + (th >> 1) * 32 + (th & 1) * 16;
                    // ascending bank-row order: (g, h, j) -> row = 8g + 4h + j within the half tile
                    for (int gg = 0; gg < 2; ++gg)
                        for (int hh = 0; hh < 2; ++hh)
                            for (int j = 0; j < 4; ++j) {
                                const float v = sc[(gg * 4 + j) * 64 + lane];
                                unsigned long long m = __ballot(v > thr);
                                m &= hh ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
                                while (m) {
                                    const int l = __builtin_ctzll(m);
                                    m &= m - 1;
                                    const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
                                    const int n = l & 31;
                                    list_insert(lst_s, lst_i, w * 32 + n, k, s, row_base + gg * 8 + hh * 4 + j, lane);
                                    const float kth = lst_s[(w * 32 + n) * HB_KL + (k - 1)];
                                    if ((lane & 31) == n) thr = kth;
                                }
                            }
                }
                ks = 0;
                ++bt;
            }
        }
        // ---- store the partial lists of this segment ----
        {
            float* gs = a.state_s + (size_t)seg.slot * (HB_QT * HB_KL) + w * 1024;
            unsigned* gi = a.state_i + (size_t)seg.slot * (HB_QT * HB_KL) + w * 1024;
            for (int e = lane; e < 1024; e += 64) { gs[e] = lst_s[w * 1024 + e]; gi[e] = lst_i[w * 1024 + e]; }
        }
        __syncthreads();   // staging buffers are reused by the next segment's first issue
    }
}

// ---- merge of the partial lists of one query: rank by counting over <= slots*k candidates --------
__global__ __launch_bounds__(64) void knn_merge_kernel(const float* __restrict__ state_s,
                                                       const unsigned* __restrict__ state_i,
                                                       const int* __restrict__ qt_off, const int* __restrict__ qt_slots,
                                                       int64_t nq, int k, int64_t id_base, int metric,
                                                       const float* __restrict__ qn2, int64_t* __restrict__ out_idx,
                                                       float* __restrict__ out_dist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t q = blockIdx.x;
    const int qt = (int)(q / HB_QT), ql = (int)(q % HB_QT);
    const int s0 = qt_off[qt], ns = qt_off[qt + 1] - s0;
    const int n = ns * k;
    float* cs = reinterpret_cast<float*>(smem);
    unsigned* ci = reinterpret_cast<unsigned*>(smem) + n;
    const int lane = threadIdx.x;
    for (int c = lane; c < n; c += 64) {
        const int sl = qt_slots[s0 + c / k], e = c % k;
        const size_t off = (size_t)sl * (HB_QT * HB_KL) + (size_t)ql * HB_KL + e;
        cs[c] = state_s[off];
        ci[c] = state_i[off];
    }
    __syncthreads();
    for (int c = lane; c < n; c += 64) {
        const float s = cs[c];
        const unsigned id = ci[c];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = cs[j];
            const unsigned ij = ci[j];
            rank += (sj > s) || (sj == s && (ij < id || (ij == id && j < c)));
        }
        if (rank < k) {
            const int64_t o = q * (int64_t)k + rank;
            if (id == HB_ID_NONE) {
                out_idx[o] = -1;
                out_dist[o] = metric == 1 ? INFINITY : -INFINITY;
            } else {
                out_idx[o] = (int64_t)id + id_base;
                if (metric == 1) { const float d2 = fmaf(-2.0f, s, qn2[q]); out_dist[o] = d2 > 0.0f ? d2 : 0.0f; }
                else out_dist[o] = s;
            }
        }
    }
}

// ---- merge of per-shard results [parts][nq][k] (multi-GPU: after the all-gather) -------------------
// IP: larger is better; L2: smaller squared distance is better.  Ties -> lower global id.
__global__ __launch_bounds__(64) void merge_parts_kernel(const float* __restrict__ dist_parts,
                                                         const int64_t* __restrict__ idx_parts, int parts, int64_t nq,
                                                         int k, int metric, int64_t* __restrict__ out_idx,
                                                         float* __restrict__ out_dist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t q = blockIdx.x;
    const int n = parts * k;
    float* cs = reinterpret_cast<float*>(smem);
    int64_t* ci = reinterpret_cast<int64_t*>(smem + ((n * 4 + 15) / 16) * 16);
    const int lane = threadIdx.x;
    for (int c = lane; c < n; c += 64) {
        const size_t off = ((size_t)(c / k) * nq + q) * k + (c % k);
        float d = dist_parts[off];
        cs[c] = metric == 1 ? -d : d;
        ci[c] = idx_parts[off];
    }
    __syncthreads();
    for (int c = lane; c < n; c += 64) {
        const float s = cs[c];
        const int64_t id = ci[c];
        // missing neighbours (id < 0) sort last
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = cs[j];
            const int64_t ij = ci[j];
            bool better;
            if (ij < 0 || id < 0) better = (ij >= 0 && id < 0) || (ij < 0 && id < 0 && j < c);
            else better = (sj > s) || (sj == s && (ij < id || (ij == id && j < c)));
            rank += better;
        }
        if (rank < k) {
            const int64_t o = q * (int64_t)k + rank;
            out_idx[o] = id < 0 ? -1 : id;
            out_dist[o] = id < 0 ? (metric == 1 ? INFINITY : -INFINITY) : (metric == 1 ? -s : s);
        }
    }
}

int hb_launch_merge_parts(const float* dist_parts, const int64_t* idx_parts, int parts, int64_t nq, int k, int metric,
                          int64_t* out_idx, float* out_dist, hipStream_t s) {
    if (nq == 0) return 0;
    const size_t n = (size_t)parts * k;
    const size_t sh = ((n * 4 + 15) / 16) * 16 + n * 8;
    if (sh > 60000) return hb_fail("hb_merge_topk: parts*k too large for the merge kernel");
    merge_parts_kernel<<<dim3((unsigned)nq), dim3(64), sh, s>>>(dist_parts, idx_parts, parts, nq, k, metric, out_idx, out_dist);
    HB_HIP(hipGetLastError());
    return 0;
}

// ---- host-side work list ------------------------------------------------------------------------------
// Pairs (query tile q, bank tile b) are processed panel by panel (a panel = `panel` consecutive bank
// tiles, sized to stay resident in the 256 MiB Infinity Cache together with the queries): inside a
// panel the q-major pair list is cut into G equal contiguous ranges, one per workgroup, so at any time
// all workgroups read the same panel (each bank byte leaves HBM about once per search) while every
// workgroup keeps working on the same <= 2 query tiles for the whole search.  Bank tiles are visited in
// ascending order for every slot, which the strict `score > threshold` filter relies on for ties.
int hb_default_panel(int nqt, int G, size_t tile_bytes) {
    auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
    int p0 = G / gcd(nqt, G);   // smallest panel for which nqt*panel divides evenly over G workgroups
    size_t budget = (size_t)96 << 20;
    int j = (int)std::max<size_t>(1, budget / (tile_bytes * (size_t)p0));
    return p0 * j;
}

void hb_build_schedule(int nqt, int nbt, int G, int panel, hb_schedule& out) {
    out = hb_schedule();
    out.nqt = nqt; out.nbt = nbt; out.panel = panel;
    const long long total_pairs = (long long)nqt * nbt;
    if (total_pairs < G) G = (int)std::max<long long>(1, total_pairs);
    out.G = G;
    std::vector<std::vector<hb_seg>> per_wg(G);
    std::map<std::pair<int, int>, int> slot_of;   // (wg, q_tile) -> slot
    std::vector<std::vector<int>> slots_of_qt(nqt);
    for (int b0 = 0; b0 < nbt; b0 += panel) {
        const int pp = std::min(panel, nbt - b0);
        const long long W = (long long)nqt * pp;
        for (int w = 0; w < G; ++w) {
            long long e0 = (W * w) / G, e1 = (W * (w + 1)) / G;
            while (e0 < e1) {
                const int q = (int)(e0 / pp), b = (int)(e0 % pp);
                const int cnt = (int)std::min<long long>(pp - b, e1 - e0);
                auto key = std::make_pair(w, q);
                auto it = slot_of.find(key);
                hb_seg sg;
                sg.q_tile = q; sg.b_tile0 = b0 + b; sg.n_tiles = cnt;
                if (it == slot_of.end()) {
                    sg.slot = out.n_slots++; sg.first = 1;
                    slot_of[key] = sg.slot;
                    slots_of_qt[q].push_back(sg.slot);
                } else { sg.slot = it->second; sg.first = 0; }
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
    out.wg_off.assign(G + 1, 0);
    for (int w = 0; w < G; ++w) {
        out.wg_off[w + 1] = out.wg_off[w] + (int)per_wg[w].size();
        out.segs.insert(out.segs.end(), per_wg[w].begin(), per_wg[w].end());
    }
    out.qt_off.assign(nqt + 1, 0);
    for (int q = 0; q < nqt; ++q) {
        out.qt_off[q + 1] = out.qt_off[q] + (int)slots_of_qt[q].size();
        out.qt_slots.insert(out.qt_slots.end(), slots_of_qt[q].begin(), slots_of_qt[q].end());
        out.max_slots_per_qt = std::max(out.max_slots_per_qt, (int)slots_of_qt[q].size());
    }
}

static int ensure_bytes(char** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) HB_HIP(hipFree(*p));
    *p = nullptr; *have = 0;
    size_t sz = need + need / 4;
    HB_HIP(hipMalloc((void**)p, sz));
    *have = sz;
    return 0;
}

// q_tiles / q_aux must already be prepared by the caller (hb_index_search).
int hb_launch_knn(hb_index* ix, int64_t nq, int k, int64_t id_base, int64_t* out_idx, float* out_dist) {
    if (k < 1 || k > HB_KL) return hb_fail("hb_index_search: k must be in [1, 32] on the fused path");
    if (nq == 0) return 0;
    const int nqt = (int)((nq + HB_QT - 1) / HB_QT);
    const int nbt = (int)((ix->ntotal + HB_BT - 1) / HB_BT);
    hipStream_t s = ix->stream;
    if (nbt == 0) {
        // empty index: every neighbour is missing (faiss returns -1 labels)
        std::vector<int64_t> hi((size_t)nq * k, -1);
        std::vector<float> hd((size_t)nq * k, ix->metric == 1 ? INFINITY : -INFINITY);
        HB_HIP(hipMemcpyAsync(out_idx, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(out_dist, hd.data(), hd.size() * 4, hipMemcpyHostToDevice, s));
        HB_HIP(hipStreamSynchronize(s));
        return 0;
    }
    const int G = ix->force_G > 0 ? ix->force_G : ix->num_cu;
    const size_t tile_bytes = (size_t)HB_BT * ix->dp * 4;
    const int panel = ix->force_panel > 0 ? ix->force_panel : hb_default_panel(nqt, std::min<long long>(G, (long long)nqt * nbt), tile_bytes);
    hb_schedule& sc = ix->sched;
    const bool rebuilt = !(sc.nqt == nqt && sc.nbt == nbt && sc.panel == panel && (sc.G == G || (long long)nqt * nbt < G));
    if (rebuilt) hb_build_schedule(nqt, nbt, G, panel, sc);
    // device copy of the work list: [segs][wg_off][qt_off][qt_slots]
    const size_t b_segs = sc.segs.size() * sizeof(hb_seg), b_wg = sc.wg_off.size() * 4, b_qo = sc.qt_off.size() * 4,
                 b_qs = sc.qt_slots.size() * 4;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_wg = al(b_segs), o_qo = o_wg + al(b_wg), o_qs = o_qo + al(b_qo), tot = o_qs + al(b_qs);
    const bool need_upload = rebuilt || ix->sched_bytes < tot;
    if (ensure_bytes(&ix->sched_dev, &ix->sched_bytes, tot)) return -1;
    if (need_upload) {
        HB_HIP(hipMemcpyAsync(ix->sched_dev, sc.segs.data(), b_segs, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(ix->sched_dev + o_wg, sc.wg_off.data(), b_wg, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(ix->sched_dev + o_qo, sc.qt_off.data(), b_qo, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(ix->sched_dev + o_qs, sc.qt_slots.data(), b_qs, hipMemcpyHostToDevice, s));
        HB_HIP(hipStreamSynchronize(s));   // host vectors may be rebuilt by the next call
    }
    const size_t state_half = (size_t)sc.n_slots * HB_QT * HB_KL * 4;
    if (ensure_bytes(&ix->state, &ix->state_bytes, 2 * state_half)) return -1;

    knn_args a;
    a.bank_tiles = ix->tiles; a.binit = ix->binit; a.q_tiles = ix->q_tiles;
    a.segs = reinterpret_cast<const hb_seg*>(ix->sched_dev);
    a.wg_off = reinterpret_cast<const int*>(ix->sched_dev + o_wg);
    a.state_s = reinterpret_cast<float*>(ix->state);
    a.state_i = reinterpret_cast<unsigned*>(ix->state + state_half);
    a.g8 = ix->g8; a.k = k;
    static bool attr_set = false;
    if (!attr_set) {
        HB_HIP(hipFuncSetAttribute((const void*)knn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS_TOTAL));
        attr_set = true;
    }
    if (ix->time_kernels) HB_HIP(hipEventRecord(ix->ev0, s));
    knn_fused_kernel<<<dim3((unsigned)sc.G), dim3(HB_THREADS), HB_LDS_TOTAL, s>>>(a);
    HB_HIP(hipGetLastError());
    if (ix->time_kernels) HB_HIP(hipEventRecord(ix->ev1, s));
    const size_t msh = (size_t)sc.max_slots_per_qt * k * 8;
    if (msh > 60000) return hb_fail("hb_index_search: too many partial lists per query tile for the merge kernel");
    const float* qn2 = ix->q_aux;   // [nq] chain ||q||^2 (valid for L2)
    knn_merge_kernel<<<dim3((unsigned)nq), dim3(64), msh, s>>>(a.state_s, a.state_i,
                                                               reinterpret_cast<const int*>(ix->sched_dev + o_qo),
                                                               reinterpret_cast<const int*>(ix->sched_dev + o_qs), nq, k,
                                                               id_base, ix->metric, qn2, out_idx, out_dist);
    HB_HIP(hipGetLastError());
    if (ix->time_kernels) {
        HB_HIP(hipEventSynchronize(ix->ev1));
        float ms = 0.f;
        HB_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
        ix->last_knn_ms = ms;
    }
    return 0;
}
