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
#include "../../include/hbird_hip.h"
#include <algorithm>
#include <array>
#include <mutex>
#include <cstring>
#include <map>

#include "hbird_knn_dev.h"

// One stage = one k8 fragment group of the pair tile: 32 MFMAs per wave, in two halves of 16 (bank row tiles
// 0-3 = "X", 4-7 = "Y").  Fragments of the next half are fetched from LDS while the current half is in the
// matrix pipe; HBM->LDS copies run three stages ahead (4-slot ring, hand-counted vmcnt).  Every non-MFMA
// instruction of the stage (9 ds_read_b128, 2 LDS-DMA copies, scalar bookkeeping) is pinned into its own gap
// between two MFMAs so that the 64-cycle matrix op ahead of it hides its issue.
#define KN_MFMA(T, FR, B, S) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(FR[(T) & 3][S], B[S], acc[T], 0, 0, 0);

// COLD: the small-search instantiation -- cold start of a slot's first tile, scan epilogue, per-tile exchange of threshold floors
// (used for searches with few tiles per workgroup).
// CL: support for L2-sharing clusters (strided segments, soft sync).  A separate instantiation: its extra scalar state
// spilled SGPRs inside the stage loop of the pool (WIDE) instantiation (k = 90: +12 % kernel time).
// (The timing-only ablation instantiations of rounds 1-3 -- no copies / no fragment reads / no barrier / no epilogue -- and the
// in-kernel counters of the small-search work are gone from the product: their numbers are in profiles/LABBOOK.md and profiles/r01 - r03.)
// CEIL (pools only): a later pass of a search with k > 256 (hb_launch_knn_bigk) -- the tile's epilogue only queues rows strictly behind the
// query's ceiling key (the last neighbour already delivered): tile_epilogue<.., CEIL>
template <bool COLD, bool WIDE, bool CL = false, bool CEIL = false>
__global__ __launch_bounds__(HB_THREADS, 2) void knn_fused_kernel(knn_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    float* lst_s = reinterpret_cast<float*>(smem + KN_LISTS);
    unsigned* lst_i = reinterpret_cast<unsigned*>(smem + KN_LISTS + HB_QT * HB_KL * 4);
    float* sc = reinterpret_cast<float*>(smem + KN_SCRATCH) + w * 256;
    unsigned* qf = reinterpret_cast<unsigned*>(smem + KN_QF) + w * 512;   // small-search instantiation: landing zone of the quota floors
    int* pcnt = reinterpret_cast<int*>(smem + KN_LISTS);
    const int g8 = a.g8, k = a.k;   // g8 = stages per bank tile
    const int myq = w * 32 + (lane & 31);
    cl_sync cs;
    if constexpr (CL) cs = cl_init(a.wg_member, a.prog, a.cl, a.lag, blockIdx.x, w == 0, smem + KN_CLWORDS);

    wg_stamp<knn_args>(0);   // (every kNN kernel stamps: the share calibration reads whatever kernel the launcher picked)
    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_end[blockIdx.x];   // this launch's share of the block's segments (phases: hb_launch_knn)
    // "everything before my first segment is done" (a member without any work: everything)
    if constexpr (CL) { if (w == 0) cl_publish(cs, seg_begin < seg_end ? a.segs[seg_begin].tile0 * g8 : 0x7FFFFFFF, lane); }
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = a.segs[si];
        const int klw = a.klw;   // list row stride (HB_KL on the LDS path)
        const int bstride = CL ? seg.stride : 1, clock0 = CL ? seg.tile0 * g8 : 0;
        float* wl_s = a.state_s + (size_t)seg.slot * HB_QT * klw;      // this slot's lists in global memory
        unsigned* wl_i = a.state_i + (size_t)seg.slot * HB_QT * klw;
        float thr;
        if constexpr (WIDE) {
            // k > HB_KL: candidate pools in global memory, fill counts in the (otherwise unused) LDS list area
            thr = pool_begin(knn_args_pool_view{a.state_cnt, a.state_thr}, seg.slot, seg.first, pcnt, myq, lane);
        } else {
            // load (or start) this query tile's partial lists; each wave owns its 32 queries
            for (int e = lane; e < 1024; e += 64) {
                lst_s[w * 1024 + e] = seg.first ? -INFINITY : wl_s[w * 1024 + e];
                lst_i[w * 1024 + e] = seg.first ? HB_ID_NONE : wl_i[w * 1024 + e];
            }
            thr = lst_s[myq * HB_KL + (k - 1)];
        }
        thr = fmaxf(thr, floor_load(a.gthr, seg.q_tile * HB_QT + myq));
        const float* qsrc = a.q_tiles + ((size_t)(seg.q_tile * 8 + w) * g8) * HB_BLK + lane * 4;
        const int total = seg.n_tiles * g8;
        f32x16 acc[8];
        f32x4 fa[4], fy[4], fb, fbk;   // X-half / Y-half bank fragments, query fragment (current, kept for Y)

        int fpar = 0, cpar = 0;   // row-init double buffer: parity of the tile being fetched / computed
        // wave w stages bank row-tile w and query row-tile w of one k8 group (1 KiB each)
        // Only waves 0-3 issue the LDS-DMA copies (4 per stage each: bank row-tiles w, w+4 and query row-tiles
        // w, w+4).  Waves w and w+4 share a SIMD: when both stalled on a copy's issue at the same point of the
        // stage the matrix pipe idled; with one issuer per SIMD the partner keeps it fed (+3.1 % measured).
        auto issue_a = [&](int bt, int ks, int slot) {
            const float* src = a.bank_tiles + ((size_t)(bt * 8 + w) * g8 + ks) * HB_BLK + lane * 4;
            if (w < 4) {
                glds16(src, smem + slot * KN_SLOT_BYTES + w * 1024);
                glds16(src + (size_t)4 * g8 * HB_BLK, smem + slot * KN_SLOT_BYTES + (w + 4) * 1024);
            }
        };
        auto issue_b = [&](int bt, int ks, int slot) {
            const float* src = qsrc + (size_t)ks * HB_BLK;
            if (w < 4) {
                glds16(src, smem + slot * KN_SLOT_BYTES + 8192 + w * 1024);
                glds16(src + (size_t)4 * g8 * HB_BLK, smem + slot * KN_SLOT_BYTES + 8192 + (w + 4) * 1024);
            }
            if (ks == 0 && w == 0) glds16(a.binit + (size_t)bt * HB_BT + lane * 4, smem + KN_BINIT + (CL ? fpar : (bt & 1)) * 1024);
        };

        int bt = seg.b_tile0, ks = 0;          // stage being computed
        int fbt = seg.b_tile0, fks = 0;        // next stage to fetch
        int slot_c = 0, slot_f = 0;            // ring slots of the computed / fetched stage
        int left = total;                      // stages not yet fetched
        // After the last real stage the fetch position stays put and the (free) slot is re-filled with the same
        // stage: the steady-state loop has no data-dependent branches around its LDS traffic, so the compiler
        // counts its lgkmcnt waits instead of draining.
        auto advance_fetch = [&]() {
            if (--left > 0) { if (++fks == g8) { fks = 0; fbt += bstride; if constexpr (CL) fpar ^= 1; } }
            if (++slot_f == KN_RING) slot_f = 0;
        };
        // vmcnt is counted by hand (the compiler does not wait for LDS-DMA at a barrier): an issuing wave has
        // exactly four copies per stage in flight (wave 0 a fifth, the row-init values, at the first stage of a
        // tile), so "all but the newest 4" covers everything up to and including the stage about to be published.
        for (int p = 0; p < KN_RING - 1; ++p) { issue_a(fbt, fks, slot_f); issue_b(fbt, fks, slot_f); advance_fetch(); }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 landed; stages 1-2 in flight
        __syncthreads();
        {
            const f32x4* A = reinterpret_cast<const f32x4*>(smem);
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = A[t * 64 + lane];
            fb = reinterpret_cast<const f32x4*>(smem + 8192)[w * 64 + lane];
        }
        for (int st = 0; st < total; ++st) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // my copies of stage st+1 have landed (st+2 in flight)
            __syncthreads();   // ... everyone's have; the slot of stage st-1 is free for stage st+3
            int slot_n = slot_c + 1; if (slot_n == KN_RING) slot_n = 0;
            const f32x4* Ac = reinterpret_cast<const f32x4*>(smem + slot_c * KN_SLOT_BYTES) + lane;
            const f32x4* An = reinterpret_cast<const f32x4*>(smem + slot_n * KN_SLOT_BYTES) + lane;
            // small searches: the floors of this tile come in by LDS-DMA during its last four stages (older than the stage's copies:
            // the hand-counted vmcnt still holds; four stages of counted waits cover them) and are read at its end -- requested at
            // the tile's START they were one tile staler: 50,176 x 384 4.62 -> 4.53 ms on the kernel with register-resident fragments
            if constexpr (COLD && !WIDE) { if (ks == (g8 > 4 ? g8 - 4 : 0)) small_floor_request(a.qfl, a.gthr, seg, w, lane, qf, sc); }
            if (ks == 0) {
                const f32x4* bi = reinterpret_cast<const f32x4*>(smem + KN_BINIT + (CL ? cpar : (bt & 1)) * 1024);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = bi[8 * t + 2 * g + h];
                        acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1];
                        acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
                    }
            }
            // ---- X half: tiles 0-3, k-steps 0-3; fillers: the four Y fragments, the two LDS-DMA copies ----
            KN_FENCE KN_MFMA(0, fa, fb, 0) KN_FENCE fy[0] = Ac[4 * 64];
            KN_FENCE KN_MFMA(1, fa, fb, 0) KN_FENCE fy[1] = Ac[5 * 64];
            KN_FENCE KN_MFMA(2, fa, fb, 0) KN_FENCE fy[2] = Ac[6 * 64];
            KN_FENCE KN_MFMA(3, fa, fb, 0) KN_FENCE fy[3] = Ac[7 * 64];
            KN_FENCE KN_MFMA(0, fa, fb, 1) KN_MFMA(1, fa, fb, 1) KN_FENCE
            // cluster soft sync: wave 0 pays the issue of one more vector-memory instruction now and then (its SIMD partner
            // covers it like it covers the copies); issued AHEAD of the stage's copies, so the hand-counted vmcnt still holds
            if constexpr (CL) { if (w == 0) cl_tick(cs, clock0 + st, lane); }
            issue_a(fbt, fks, slot_f);
            KN_FENCE KN_MFMA(2, fa, fb, 1) KN_MFMA(3, fa, fb, 1) KN_MFMA(0, fa, fb, 2) KN_MFMA(1, fa, fb, 2) KN_FENCE
            issue_b(fbt, fks, slot_f);
            KN_FENCE KN_MFMA(2, fa, fb, 2) KN_MFMA(3, fa, fb, 2) KN_FENCE
            advance_fetch();
            KN_FENCE KN_MFMA(0, fa, fb, 3) KN_MFMA(1, fa, fb, 3) KN_MFMA(2, fa, fb, 3) KN_MFMA(3, fa, fb, 3) KN_FENCE
            fbk = fb;   // the Y half still needs this stage's query fragment
            // ---- Y half: tiles 4-7; fillers: the X fragments and the query fragment of stage st+1 ----
            KN_FENCE KN_MFMA(4, fy, fbk, 0) KN_FENCE fa[0] = An[0 * 64];
            KN_FENCE KN_MFMA(5, fy, fbk, 0) KN_FENCE fa[1] = An[1 * 64];
            KN_FENCE KN_MFMA(6, fy, fbk, 0) KN_FENCE fa[2] = An[2 * 64];
            KN_FENCE KN_MFMA(7, fy, fbk, 0) KN_FENCE fa[3] = An[3 * 64];
            KN_FENCE KN_MFMA(4, fy, fbk, 1) KN_FENCE
            fb = reinterpret_cast<const f32x4*>(smem + slot_n * KN_SLOT_BYTES + 8192)[w * 64 + lane];
            KN_FENCE
            KN_MFMA(5, fy, fbk, 1) KN_MFMA(6, fy, fbk, 1) KN_MFMA(7, fy, fbk, 1)
            KN_MFMA(4, fy, fbk, 2) KN_MFMA(5, fy, fbk, 2) KN_MFMA(6, fy, fbk, 2) KN_MFMA(7, fy, fbk, 2)
            KN_MFMA(4, fy, fbk, 3) KN_MFMA(5, fy, fbk, 3) KN_MFMA(6, fy, fbk, 3) KN_MFMA(7, fy, fbk, 3)
            KN_FENCE
            slot_c = slot_n;
            if (++ks == g8) {
                if constexpr (WIDE) {
                    // the slot's pool pointers are derived HERE (from one scalar, laundered so that the compiler cannot
                    // hoist them): kept live through the stage loop they crowd out the copy loop's own pointers, which
                    // then come back from spilled SGPRs in every stage (k = 90: +10 % kernel time)
                    int slot_ = seg.slot;
                    asm volatile("" : "+s"(slot_));
                    float* ps = a.state_s + (size_t)slot_ * HB_QT * klw;
                    unsigned* pi = a.state_i + (size_t)slot_ * HB_QT * klw;
                    if constexpr (CEIL) {
                        // (the ceiling is read HERE, once per tile, not kept in registers through the stage loop)
                        const float c_s = a.ceil_s[seg.q_tile * HB_QT + myq];
                        const unsigned c_i = a.ceil_i[seg.q_tile * HB_QT + myq];
                        tile_epilogue<true, true, HB_POOL_MAX / 64, true>(acc, thr, ps, pi, sc, w * 32, lane, k, (unsigned)bt, klw, pcnt, c_s, c_i);
                    } else tile_epilogue<true, true>(acc, thr, ps, pi, sc, w * 32, lane, k, (unsigned)bt, klw, pcnt);
                } else if constexpr (COLD) {
                    // small searches: cold start of a slot's first tile (separate instantiation, see launcher)
                    if (seg.first && bt == seg.b_tile0 && cold_start_needed(thr)) thr = fmaxf(thr, cold_start_threshold(acc, k));
                    // the floors requested four stages ago (waves 4-7 have nothing else in flight; waves 0-3 have passed
                    // counted waits that cover them -- except in a one-stage tile, D <= 8)
                    if (g8 < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 .Lfl_%=\n\ts_waitcnt vmcnt(0)\n.Lfl_%=:" :: "s"(w) : "memory", "scc");
                    thr = fmaxf(thr, small_floor_read(seg, qf, sc, lane));
                    // (few rows per slot -> many insertions per tile): scan + register queue; the big searches keep the plain
                    // epilogue (insertions are rare there, and the scan's registers would spill)
                    list_epilogue_scan(acc, thr, lst_s, lst_i, w * 32, lane, k, (unsigned)bt);
                    small_floor_publish(a.qfl, a.gthr, seg, lst_s, myq, k, thr, lane);
                } else tile_epilogue<true, false>(acc, thr, lst_s, lst_i, sc, w * 32, lane, k, (unsigned)bt);
                ks = 0;
                bt += bstride; if constexpr (CL) cpar ^= 1;
            }
        }
        if constexpr (CL) { if (w == 0) cl_publish(cs, seg.next_tile0 == 0x7FFFFFFF ? 0x7FFFFFFF : seg.next_tile0 * g8, lane); }   // covers idle units
        if constexpr (!WIDE) {   // store the partial lists of this segment
            for (int e = lane; e < 1024; e += 64) { wl_s[w * 1024 + e] = lst_s[w * 1024 + e]; wl_i[w * 1024 + e] = lst_i[w * 1024 + e]; }
        } else {
            pool_end(knn_args_pool_view{a.state_cnt, a.state_thr}, seg.slot, pcnt, thr, myq, lane);
        }
        if (lane < 32) floor_publish(a.gthr, seg.q_tile * HB_QT + myq, thr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (unused) run-ahead copies
        __syncthreads();   // the ring is reused by the next segment's prologue
    }
    if constexpr (CL) cl_finish(cs, a.cl_stats, w == 0, lane);
    wg_stamp<knn_args>(1);
}

// ---- merge of the partial lists of one query: rank by counting over <= slots*k candidates --------
static int ensure_bytes(char** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) HB_HIP(hipFree(*p));
    *p = nullptr; *have = 0;
    size_t sz = need + need / 4;
    HB_HIP(hipMalloc((void**)p, sz));
    *have = sz;
    return 0;
}

// Slot selection of a merge block: either the work list's slots of the query tile (qt_slots) or, in the second
// level of a two-level merge, the `fixed_ng` group lists written by the first level (slot = qt * fixed_ng + j).
// grp_size > 0 (first level): block (q, grp) merges only slots [grp*grp_size, (grp+1)*grp_size) of its query tile
// and writes a sorted list of k entries (row stride klw) into (tmp_s, tmp_i) slot qt * n_groups + grp.
// A slot holds `per` entries per query: a sorted list (per = k, sentinel-padded) or, with cnts != nullptr, an
// unsorted candidate pool of which the first cnts[slot][query] entries are valid.
__global__ __launch_bounds__(64) void knn_merge_kernel(const float* __restrict__ state_s,
                                                       const unsigned* __restrict__ state_i,
                                                       const int* __restrict__ cnts, const float* __restrict__ pthr, int per,
                                                       const int* __restrict__ qt_off, const int* __restrict__ qt_slots,
                                                       int fixed_ng, int grp_size, int n_groups, float* __restrict__ tmp_s,
                                                       unsigned* __restrict__ tmp_i,
                                                       int64_t nq, int k, int klw, int64_t id_base, int metric,
                                                       const float* __restrict__ qn2, int64_t* __restrict__ out_idx,
                                                       float* __restrict__ out_dist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t q = blockIdx.x;
    const int grp = blockIdx.y;
    const int qt = (int)(q / HB_QT), ql = (int)(q % HB_QT);
    int s0, ns;
    if (fixed_ng > 0) { s0 = qt * fixed_ng; ns = fixed_ng; }
    else { s0 = qt_off[qt]; ns = qt_off[qt + 1] - s0; }
    if (grp_size > 0) {
        const int lo = min(ns, grp * grp_size), hi = min(ns, lo + grp_size);
        s0 += lo; ns = hi - lo;
    }
    float* cs = reinterpret_cast<float*>(smem);
    unsigned* ci = reinterpret_cast<unsigned*>(smem) + ns * per;
    const int lane = threadIdx.x;
    // Pre-filter: a slot that knows k candidates reaching v (a sorted list: its k-th entry; a pool: the threshold of its last
    // compaction, or a foreign floor it ran under -- both lower bounds of the k-th best of the union) rules out everything
    // below the largest such v.  With many slots per query tile (few queries against a big bank: one slot per workgroup) the
    // rank-by-counting below is quadratic in slots x k: 1,369 queries x 200 k rows in use_fp16 mode spent 20 ms here.
    float tstar = -INFINITY;
    for (int j = lane; j < ns; j += 64) {
        const int sl = fixed_ng > 0 ? s0 + j : qt_slots[s0 + j];
        const size_t oq = (size_t)sl * HB_QT + ql;
        if (cnts) { if (pthr && cnts[oq] >= k) tstar = fmaxf(tstar, pthr[oq]); }
        else tstar = fmaxf(tstar, state_s[oq * klw + (k - 1)]);
    }
    for (int o = 32; o > 0; o >>= 1) tstar = fmaxf(tstar, __shfl_xor(tstar, o));
    // gather the surviving candidates of the slots, densely packed
    int n = 0;
    for (int j = 0; j < ns; ++j) {
        const int sl = fixed_ng > 0 ? s0 + j : qt_slots[s0 + j];
        const int valid = cnts ? min(per, cnts[(size_t)sl * HB_QT + ql]) : per;
        const size_t off = ((size_t)sl * HB_QT + ql) * klw;
        // all of a slot's entries (at most HB_POOL_MAX = 8 x 64) are requested before the first is looked at: the loop used to wait for a
        // score, then for its id, 64 entries at a time -- two memory round trips per 64 entries, 1.18 of a 15.6 ms search at k = 90
        float vv[HB_POOL_MAX / 64];
        unsigned vi[HB_POOL_MAX / 64];
#pragma unroll
        for (int u = 0; u < HB_POOL_MAX / 64; ++u) {
            const int e = u * 64 + lane;
            vv[u] = -INFINITY; vi[u] = 0u;
            if (e < valid) { vv[u] = state_s[off + e]; vi[u] = state_i[off + e]; }
        }
#pragma unroll
        for (int u = 0; u < HB_POOL_MAX / 64; ++u) {
            if (u * 64 >= valid) break;
            const int e = u * 64 + lane;
            // (sentinel entries -- the padding of a sorted list with fewer than k rows -- stay behind: a seeded second pass of few queries
            // leaves 22 group lists of 256 entries with a hundred real candidates among them, and the ranking below is quadratic: 18 ms
            // for two queries, profiles/r06/fp16_escalation_merge_before.csv)
            const bool keep = e < valid && vi[u] != HB_ID_NONE && (vv[u] >= tstar || tstar == -INFINITY);
            const unsigned long long m = __ballot(keep);
            if (keep) { const int pos = n + __popcll(m & ((1ull << lane) - 1ull)); cs[pos] = vv[u]; ci[pos] = vi[u]; }
            n += __popcll(m);
        }
    }
    // Still many: keep about k of them, so that the rank-by-counting below sees about k candidates instead of slots x k (it is quadratic:
    // 6,144 candidates of 32 pools took a wave 0.4 ms).  A bisection over the monotone keys finds a cut with k .. k + 32 candidates above
    // it (one pass over the candidates per round, counted 64 at a time by ballot; it stops as soon as the count fits: about eight rounds;
    // until round 4 a radix select of the exact k-th key ran here, always 32 rounds: 1.18 of a 15.6 ms use_fp16 search at 300,000 x 768,
    // k = 90) -- every candidate that can rank below k lies above the cut, ties of the k-th included.  Scores that the bisection cannot
    // separate (24 rounds) go through the radix select.
    if (n > k + 64) {                // (the ranking below is quadratic: from k + 64 candidates on a cut pays)
        unsigned cut = 0;            // keep keys > cut
        {
            unsigned lo = 0, hi = 0xFFFFFFFFu;     // at least k keys above lo (all n: no score has key 0), fewer than k above hi
            int clo = n;
            for (int r = 0; r < 24 && hi - lo > 1u && clo > k + 32; ++r) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
                for (int base = 0; base < n; base += 64) {
                    const int idx = base + lane;
                    c += __popcll(__ballot(idx < n && pool_key(cs[idx]) > mid));
                }
                if (c >= k) { lo = mid; clo = c; } else hi = mid;
            }
            cut = lo;
            if (clo > k + 64 && clo > 2 * k) {   // not separated: the exact k-th key (its ties stay)
                unsigned prefix = 0;
                int kk = k;
                for (int b = 31; b >= 0; --b) {
                    const unsigned himask = ~((1u << b) - 1u), want = prefix | (1u << b);
                    int c = 0;
                    for (int base = 0; base < n; base += 64) {
                        const int idx = base + lane;
                        c += __popcll(__ballot(idx < n && (pool_key(cs[idx]) & himask) == want));
                    }
                    if (c >= kk) prefix = want; else kk -= c;
                }
                cut = prefix - 1u;    // (prefix >= 1: the key of a score)
            }
        }
        int m = 0;
        for (int base = 0; base < n; base += 64) {   // in place: a kept entry moves to a position at or below its own
            const int idx = base + lane;
            const float v = idx < n ? cs[idx] : 0.0f;
            const unsigned id = idx < n ? ci[idx] : 0u;
            const bool keep = idx < n && pool_key(v) > cut;
            const unsigned long long mk = __ballot(keep);
            if (keep) { const int pos = m + __popcll(mk & ((1ull << lane) - 1ull)); cs[pos] = v; ci[pos] = id; }
            m += __popcll(mk);
        }
        n = m;
    }
    const size_t tmp_off = grp_size > 0 ? ((size_t)(qt * n_groups + grp) * HB_QT + ql) * klw : 0;
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
            if (grp_size > 0) { tmp_s[tmp_off + rank] = s; tmp_i[tmp_off + rank] = id; continue; }
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
    for (int r = n + lane; r < k; r += 64) {   // fewer than k candidates: missing neighbours
        if (grp_size > 0) { tmp_s[tmp_off + r] = -INFINITY; tmp_i[tmp_off + r] = HB_ID_NONE; continue; }
        out_idx[q * (int64_t)k + r] = -1;
        out_dist[q * (int64_t)k + r] = metric == 1 ? INFINITY : -INFINITY;
    }
}

// Merge the partial lists / pools of every query: one level when the candidates of a query tile's slots fit the merge
// block's LDS, two levels otherwise (few query tiles against a big bank: up to one slot per workgroup).
// cnts == nullptr: sorted lists of k entries; otherwise pools of capacity klw with fill counts cnts.
static int launch_merge(hb_index* ix, const float* state_s, const unsigned* state_i, const int* cnts, const float* pthr, const int* qt_off,
                        const int* qt_slots, int max_slots, int nqt, int64_t nq, int k, int klw, int64_t id_base, int metric,
                        const float* qn2, int64_t* out_idx, float* out_dist, hipStream_t s) {
    const size_t lim = 48 * 1024;
    const int per = cnts ? klw : k;
    if ((size_t)max_slots * per * 8 <= lim) {
        knn_merge_kernel<<<dim3((unsigned)nq), dim3(64), (size_t)max_slots * per * 8, s>>>(state_s, state_i, cnts, pthr, per, qt_off, qt_slots,
                                                                                            0, 0, 0, nullptr, nullptr, nq, k, klw,
                                                                                            id_base, metric, qn2, out_idx, out_dist);
        HB_HIP(hipGetLastError());
        return 0;
    }
    const int grp = std::max<int>(2, (int)(lim / ((size_t)per * 8)));
    const int ng = (max_slots + grp - 1) / grp;
    if ((size_t)ng * k * 8 > lim) return hb_fail("hb_index_search: too many partial lists per query tile for the merge kernel");
    const size_t half = (size_t)nqt * ng * HB_QT * klw * 4;
    if (ensure_bytes(&ix->mtmp, &ix->mtmp_bytes, 2 * half)) return -1;
    float* ts = reinterpret_cast<float*>(ix->mtmp);
    unsigned* ti = reinterpret_cast<unsigned*>(ix->mtmp + half);
    knn_merge_kernel<<<dim3((unsigned)nq, (unsigned)ng), dim3(64), (size_t)grp * per * 8, s>>>(state_s, state_i, cnts, pthr, per, qt_off, qt_slots,
                                                                                                0, grp, ng, ts, ti, nq, k, klw, 0, 0,
                                                                                                nullptr, nullptr, nullptr);
    HB_HIP(hipGetLastError());
    knn_merge_kernel<<<dim3((unsigned)nq), dim3(64), (size_t)ng * k * 8, s>>>(ts, ti, nullptr, nullptr, k, nullptr, nullptr, ng, 0, 0, nullptr,
                                                                                nullptr, nq, k, klw, id_base, metric, qn2, out_idx,
                                                                                out_dist);
    HB_HIP(hipGetLastError());
    return 0;
}

// ---- merge of per-shard results [parts][nq][k] (multi-GPU: after the all-gather) -------------------
// IP: larger is better; L2: smaller squared distance is better.  Ties -> lower global id.
// A part's lists start at dist_parts + p * dist_stride / idx_parts + p * idx_stride (elements): [parts][nq][k] arrays
// (stride nq*k) or the packed per-rank buffers of one all-gather (hb_merge_topk_packed).
__global__ __launch_bounds__(64) void merge_parts_kernel(const float* __restrict__ dist_parts,
                                                         const int64_t* __restrict__ idx_parts, int parts, int64_t nq,
                                                         int k, int metric, int64_t dist_stride, int64_t idx_stride,
                                                         int64_t* __restrict__ out_idx, float* __restrict__ out_dist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t q = blockIdx.x;
    const int n = parts * k;
    float* cs = reinterpret_cast<float*>(smem);
    int64_t* ci = reinterpret_cast<int64_t*>(smem + ((n * 4 + 15) / 16) * 16);
    const int lane = threadIdx.x;
    for (int c = lane; c < n; c += 64) {
        const size_t off = (size_t)q * k + (c % k);
        float d = dist_parts[(size_t)(c / k) * dist_stride + off];
        cs[c] = metric == 1 ? -d : d;
        ci[c] = idx_parts[(size_t)(c / k) * idx_stride + off];
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
                          int64_t dist_stride, int64_t idx_stride, int64_t* out_idx, float* out_dist, hipStream_t s) {
    if (nq == 0) return 0;
    const size_t n = (size_t)parts * k;
    const size_t sh = ((n * 4 + 15) / 16) * 16 + n * 8;
    if (sh > 60000) return hb_fail("hb_merge_topk: parts*k too large for the merge kernel");
    merge_parts_kernel<<<dim3((unsigned)nq), dim3(64), sh, s>>>(dist_parts, idx_parts, parts, nq, k, metric, dist_stride, idx_stride,
                                                                 out_idx, out_dist);
    HB_HIP(hipGetLastError());
    return 0;
}

// (the host-side work list -- hb_build_schedule and friends -- lives in hbird_schedule.cpp: plain C++, also built host-only under sanitizers)

// Phased searches (pools: k > HB_KL and the fp16 candidate pass).  A pool's threshold is the k-th best of ONE slot's share of the rows
// and only rises when the pool is compacted: with S slots per query tile every slot re-discovers what the others already know, and
// what the pools' epilogue costs is the candidates it appends, not the test (measured with perfect floors -- a repeated search that
// starts from the previous one's final floors: fp16 candidate kernel 29.2 -> 19.4 ms at 2,074,072 x 384, 317 -> 300 ms at 10 M x 768;
// 18.1 ms without any epilogue).  So the search is launched in PHASES of growing size -- every workgroup's segment list cut at the
// same clocks (hb_build_schedule; whole segments per phase left most workgroups idle in the small phases and was slower than no
// phases at all) -- and between two phases the slots' pools are merged (the merge kernel of the final result) and the k-th best of
// ALL rows seen so far becomes every slot's floor: exactly the union's k-th best, so the results keep their bits.  Same box, kernel
// ms without / with phases: fp16 k = 30: 50,176 x 384 4.62 / 2.30, 300 k x 768 12.35 / 8.45, 1 M x 1024 (4,096 queries) 12.9 /
// 8.95, 2,074,072 x 384 29.6 / 25.3, 10 M x 768 321.6 / 317; fp32: k = 90 50,176 x 384 9.75 / 6.2, k = 64 300 k x 768 51.2 / 44.8,
// k = 90 2,074,072 x 384 161.4 / 146.1.  The LDS lists (k <= 32) stay unphased: their slots already exchange floors every tile
// (small_floor_*) and phases only added launches (50,176 x 384: 4.72 -> 4.86-5.19 ms).
__global__ __launch_bounds__(256) void seed_floors_kernel(const int64_t* __restrict__ idx, const float* __restrict__ score, int64_t nq, int kk,
                                                         unsigned* __restrict__ gthr) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    if (idx[q * kk + kk - 1] >= 0) atomicMax(gthr + q, pool_key(score[q * kk + kk - 1]));   // kk rows reach this score: a floor (ties pass)
}

__global__ __launch_bounds__(256) void seed_from_scores_kernel(const float* __restrict__ seed, int64_t nq, unsigned* __restrict__ gthr) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float v = seed[q];
    if (v > -INFINITY && v < INFINITY) atomicMax(gthr + q, pool_key(v));      // rows scoring >= v pass (floor_from_key admits ties); NaN / inf: no seed
}

// The same floor straight from the pools, without the merge: one wave per query gathers the scores of its slots' pools (those that
// can matter: at or above the best threshold a full pool already has) into LDS and bisects for a score that at least kk of them
// exceed -- 14 halvings between the smallest and the largest; any such score is a valid floor (cold_start_threshold,
// hbird_knn_dev.h).  The merge kernel ranks its candidates by counting (quadratic) and writes int64 ids: 85-150 us per phase
// boundary for 12,544 queries against 25-30 here.
__global__ __launch_bounds__(256) void pool_floor_kernel(const float* __restrict__ state_s, const int* __restrict__ cnts,
                                                        const float* __restrict__ pthr, const int* __restrict__ qt_off,
                                                        const int* __restrict__ qt_slots, int64_t nq, int kk, int klw, int per_wave,
                                                        unsigned* __restrict__ gthr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (one query per wave: a launch of its own has all the parallelism it wants -- a two-queries-per-wave form, one per lane half, built in round 5
    // for floors computed inside the kNN kernels, measured 25 us per boundary SLOWER here)
    const int64_t q = (int64_t)blockIdx.x * 4 + w;
    if (q >= nq) return;
    pool_floor_query(state_s, cnts, pthr, qt_off, qt_slots, q, kk, klw, reinterpret_cast<float*>(smem) + (size_t)w * per_wave, gthr, lane);
}

// Per-XCD work shares, calibrated.  The eight XCDs of an MI355X do not run the fp32 kernel at one speed: with equal work the workgroups of
// the odd XCDs finish 1-2 % after those of the even ones (profiles/r05/xcd_speed_stamps_headline.txt), and a launch lasts as long as its
// slowest workgroup.  Big fp32 searches therefore stamp every workgroup's start and end (two stores per block), the stamps travel to pinned
// host memory behind the launch, and the NEXT such search -- if that copy has completed; it never waits for it -- turns each XCD group's
// median duration into its share of the work list: share_x <- share_x x (median of all / median of group x).  One round brings the groups
// within 0.1-0.3 % of each other (10 M x 768: 2275 -> 2258 ms, 2.5 M x 768: 573.7 -> 568.4; profiles/r05/xcd_weights_iterated.txt); later
// rounds only act on a change beyond 0.3 %.  Speed only: any shares give the same results.  The fp16 candidate kernel calibrates shares of its own (its
// XCDs differ by other amounts: it runs on the chip's power budget): 274.0 -> 270.0 ms at 10 M x 768, 18.6 -> 18.4 at 2 M x 384 -- but only
// since a phased list's cuts follow the shares (hb_finish_schedule); with common cuts, where a group's extra share lands in the last phase,
// the same shares made it SLOWER (281 -> 288-297 ms).  Shares are remembered per device and family for indexes created later.
static std::mutex g_xcd_mu;
static std::map<std::pair<int, int>, std::array<double, 8>> g_xcd_known;     // (device, kernel family) -> last calibrated shares
static std::map<int, int> g_cl_known;                                        // device -> decided: fp32 clusters on (1) / off (0)
static void hb_xcd_calibrate(hb_index* ix, int fam) {
    hb_index::xcd_cal& c = ix->xcal[fam];
    if (c.rounds == 0 && !c.stamp_pending) {        // a new index starts from what this device is known to need
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        auto it = g_xcd_known.find({ix->device, fam});
        if (it != g_xcd_known.end()) {
            bool differs = false;
            for (int x = 0; x < 8; ++x) { differs = differs || c.w[x] != it->second[x]; c.w[x] = it->second[x]; }
            if (differs) ix->sched = hb_schedule();
            c.rounds = 1;
        }
        auto cl = g_cl_known.find(ix->device);
        if (fam == 0 && cl != g_cl_known.end() && c.cl_state != 2) { c.cl_state = 2; c.cl_choice = cl->second; ix->sched = hb_schedule(); }
    }
    if (!c.stamp_pending || !c.stamp_ev || !c.stamp_host || hipEventQuery(c.stamp_ev) != hipSuccess) { (void)hipGetLastError(); return; }
    const int G = c.stamp_pending;
    c.stamp_pending = 0;
    // the decisions themselves are plain host code (hbird_calibrate.cpp: also fed with synthetic stamps by tests/test_calibrate_cpu.py)
    hb_stamp_set set;
    set.stamps = c.stamp_host; set.G = G; set.key = c.stamp_key; set.frac = c.stamp_frac; set.auto_cluster = c.stamp_auto_cluster;
    for (int x = 0; x < 8; ++x) set.run_shares[x] = c.stamp_w[x];
    const int flags = hb_xcd_step(c, fam, set);
    if (flags & HB_CAL_REBUILD) ix->sched = hb_schedule();                     // rebuilt with the new shares / cluster form by the caller
    if (flags & (HB_CAL_REMEMBER_SHARES | HB_CAL_REMEMBER_CLUSTERS)) {
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        if (flags & HB_CAL_REMEMBER_SHARES) {
            std::array<double, 8> keep;
            for (int x = 0; x < 8; ++x) keep[x] = c.w[x];
            g_xcd_known[{ix->device, fam}] = keep;
        }
        if (flags & HB_CAL_REMEMBER_CLUSTERS) g_cl_known[ix->device] = c.cl_choice;
    }
}
// behind a calibrating launch: its per-block stamps -> pinned host memory, read by the next big search of the family if the copy has completed by then
static int hb_xcd_collect(hb_index* ix, int fam, const unsigned* stamps_dev, const hb_schedule& sc, const double* shares, int n_phases, int nqt, int nbt,
                          int k, hipStream_t s, bool auto_cluster = false) {
    hb_index::xcd_cal& c = ix->xcal[fam];
    if (!stamps_dev || sc.G > 1024 || c.stamp_pending) return 0;
    if (!c.stamp_ev) HB_HIP(hipEventCreateWithFlags(&c.stamp_ev, hipEventDisableTiming));
    if (!c.stamp_host) HB_HIP(hipHostMalloc((void**)&c.stamp_host, 1024 * 32, hipHostMallocDefault));
    HB_HIP(hipMemcpyAsync(c.stamp_host, stamps_dev, (size_t)sc.G * 32, hipMemcpyDeviceToHost, s));
    HB_HIP(hipEventRecord(c.stamp_ev, s));
    c.stamp_pending = sc.G;
    for (int x = 0; x < 8; ++x) c.stamp_w[x] = shares[x];
    c.stamp_key = {nqt, nbt, sc.G, n_phases, k, sc.cq * 16 + sc.cb};
    c.stamp_auto_cluster = auto_cluster ? 1 : 0;
    // a phased search stamps its last launch.  Cuts that follow the shares (long lists, hb_finish_schedule) make it a fair sample; with
    // common cuts a group's extra share is all in that launch -- its part of the work
    const double per_wg = (double)nqt * (double)nbt / std::max(1, sc.G);
    c.stamp_frac = n_phases > 1 && !sc.cuts_scaled && per_wg > 0.0 ? std::min(1.0, std::max(0.25, 1.0 - (double)sc.phase_clock.back() / per_wg)) : 1.0;
    return 0;
}

// ---- k beyond the pools' 256 (the reference forwards any k to Faiss, search_faiss.py:84-85; faiss-gpu takes up to 2048) ------------------------
// The best k = the best 256, then the best 256 of the rows BEHIND those in the ordering (score descending, id ascending), and so on: passes of
// the pool search, each with a per-query ceiling key = the last neighbour delivered so far.  A pass costs a whole search (k = 1024: four), which
// is what an exact flat search of that width costs here; ids and distance bits are those of one search with a list of k.
__global__ __launch_bounds__(256) void bigk_place_kernel(const int64_t* __restrict__ tmp_idx, const float* __restrict__ tmp_sc, int64_t nq, int kp, int k,
                                                         int col0, int64_t id_base, int out_metric, const float* __restrict__ qn2,
                                                         int64_t* __restrict__ out_idx, float* __restrict__ out_dist, float* __restrict__ ceil_s,
                                                         unsigned* __restrict__ ceil_i) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nq * kp) return;
    const int64_t q = t / kp;
    const int j = (int)(t % kp);
    const int64_t id = tmp_idx[t];
    const float s = tmp_sc[t];
    const int64_t o = q * (int64_t)k + col0 + j;
    if (id < 0) { out_idx[o] = -1; out_dist[o] = out_metric == 1 ? INFINITY : -INFINITY; }
    else {
        out_idx[o] = id + id_base;
        if (out_metric == 1) { const float d2 = fmaf(-2.0f, s, qn2[q]); out_dist[o] = d2 > 0.0f ? d2 : 0.0f; }
        else out_dist[o] = s;
    }
    if (j == kp - 1) {      // the next pass starts behind this entry; a list that ran out of rows closes the search (nothing is behind -inf)
        ceil_s[q] = id < 0 ? -INFINITY : s;
        ceil_i[q] = id < 0 ? 0xFFFFFFFFu : (unsigned)id;
    }
}

int hb_launch_knn(hb_index* ix, const float* q_dev, int64_t nq, int k, int64_t id_base, int64_t* out_idx, float* out_dist);
static int hb_launch_knn_bigk(hb_index* ix, const float* q_dev, int64_t nq, int k, int64_t id_base, int64_t* out_idx, float* out_dist) {
    if (nq == 0) return 0;
    hipStream_t s = ix->stream;
    const int64_t nqp = (nq + HB_QT - 1) / HB_QT * HB_QT;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_sc = al((size_t)nq * 256 * 8), o_cs = o_sc + al((size_t)nq * 256 * 4), o_ci = o_cs + al((size_t)nqp * 4), tot = o_ci + al((size_t)nqp * 4);
    if (ensure_bytes(&ix->bigk, &ix->bigk_bytes, tot)) return -1;
    int64_t* tmp_idx = reinterpret_cast<int64_t*>(ix->bigk);
    float* tmp_sc = reinterpret_cast<float*>(ix->bigk + o_sc);
    float* ceil_s = reinterpret_cast<float*>(ix->bigk + o_cs);
    unsigned* ceil_i = reinterpret_cast<unsigned*>(ix->bigk + o_ci);
    HB_HIP(hipMemsetD32Async((hipDeviceptr_t)ceil_s, 0xFF800000u, (size_t)nqp, s));      // (padding queries: nothing behind -inf)
    HB_HIP(hipMemsetD32Async((hipDeviceptr_t)ceil_i, 0xFFFFFFFFu, (size_t)nqp, s));
    const int out_metric = ix->score_output ? 0 : ix->metric;
    const int saved_so = ix->score_output, saved_t = ix->time_kernels;
    double knn_ms = 0.0;
    int rc = 0;
    for (int col0 = 0; col0 < k && !rc; col0 += 256) {
        const int kp = std::min(256, k - col0);
        ix->score_output = 1;                                // ordering scores: what the next pass's ceiling compares with
        ix->ceil_s_dev = col0 ? ceil_s : nullptr; ix->ceil_i_dev = col0 ? ceil_i : nullptr;
        rc = hb_launch_knn(ix, q_dev, nq, kp, 0, tmp_idx, tmp_sc);
        ix->score_output = saved_so; ix->ceil_s_dev = nullptr; ix->ceil_i_dev = nullptr;
        if (rc) break;
        if (saved_t) knn_ms += ix->last_knn_ms;
        bigk_place_kernel<<<dim3((unsigned)((nq * kp + 255) / 256)), dim3(256), 0, s>>>(tmp_idx, tmp_sc, nq, kp, k, col0, id_base, out_metric, ix->q_aux,
                                                                                         out_idx, out_dist, ceil_s, ceil_i);
        HB_HIP(hipGetLastError());
    }
    if (saved_t) ix->last_knn_ms = knn_ms;
    return rc;
}

// q_tiles / q_aux must already be prepared by the caller (hb_index_search).
int hb_launch_knn(hb_index* ix, const float* q_dev, int64_t nq, int k, int64_t id_base, int64_t* out_idx, float* out_dist) {
    if (k < 1 || k > HB_MAX_K) return hb_fail("hb_index_search: k must be in [1, " + std::to_string(HB_MAX_K) + "] (faiss-gpu's limit, search_faiss.py:84-85)");
    if (k > 256) return hb_launch_knn_bigk(ix, q_dev, nq, k, id_base, out_idx, out_dist);
    const bool ceil = ix->ceil_s_dev != nullptr;      // a later pass of a search with k > 256: pools, the LDS-staged kernel's CEIL instantiation
    // fp16 mode 2 (what the plugin's use_fp16=True selects): the candidate pass only where it pays.  Its fixed costs are per query
    // (fp16 query tiles, re-rank of k' = 2k candidates, a second merge) and per search (the phases' launches and floor kernels), what it
    // saves is proportional to the matrix work rows x queries x D: the pass runs from rows x queries x D >= 1.5e10 x (k' / 64)^2 on, and
    // never below 4,096 rows.  Whole searches, fp32 / use_fp16 ms, same box, after round 4's changes (tools/exp_fp16_crossover.py): 196 x 384
    // queries: 65 k rows 0.28 / 0.34, 131 k 0.47 / 0.45, 1 M 1.93 / 0.95; 784 x 384: 16 k 0.28 / 0.34, 32 k 0.46 / 0.46, 65 k 0.64 / 0.51; 3,136 x
    // 384: 8 k 0.41 / 0.45, 16 k 0.65 / 0.47; 1,369 x 768: 4 k 0.30 / 0.23, 16 k 0.59 / 0.48; 12,544 x 384: 4 k 0.61 / 0.52, 50 k 3.9 / 1.5, 1 M
    // 69.3 / 10.6; 21,904 x 768: 4 k 1.43 / 1.09, 1 M 242.7 / 34.3; k = 90, 12,544 x 384: 16 k 1.74 / 2.04, 32 k 3.00 / 2.49, 1 M 70.4 / 13.8.
    // (Until round 4: at least 16,384 rows and rows x queries >= 2^27.)  Same results either way.
    const int esc = ix->esc_level;      // 0: a caller's search; 1: the second fp16 pass over its uncertified queries; 2: the fp32 search of what is left
    const double kc_rel = std::min(256, std::max(64, (2 * k + 7) / 8 * 8)) / 64.0;
    bool f16 = ix->fp16 != 0 && k <= 128 && !ceil &&      // (a later pass of a search with k > 256 runs behind a ceiling, which only the fp32 pool kernel knows)
               (ix->fp16 == 1 || (ix->ntotal >= 4096 && (double)ix->ntotal * (double)nq * (double)ix->d >= 1.5e10 * kc_rel * kc_rel));
    // ADAPTIVE use (mode 2, round 6): on a bank whose neighbours sit closer together than fp16 can tell apart -- token worlds with little
    // noise: profiles/r06/final/fp16_cliff_*.json -- most certificates fail, and passes that certify nothing are pure overhead.  The index keeps
    // moving averages of the share of queries that failed the first certificate (r1) and of the share that reached the fp32 kernel (r12):
    // r12 > 1/2 -> the fp32 kernel right away; r1 > 1/2 -> the first pass is skipped, ONE pass with k' = 256 serves all queries; every 16th
    // search walks the whole chain again, so a bank (or a query stream) that changes is noticed.  Same bits on every path.
    bool wide_first = false;
    int how16 = HB_F16_CHAIN;
    if (f16 && esc == 0 && ix->fp16 == 2 && ix->fp16_escalation == 0) {       // (the policy itself: hbird_calibrate.cpp, with CPU tests)
        how16 = hb_f16_choose(ix->f16_adapt);
        if (how16 == HB_F16_FP32) { f16 = false; ix->f16_skipped = 1; }
        else if (how16 == HB_F16_WIDE_FIRST) wide_first = true;
    }
    if (!f16 && esc == 0) { ix->last_fp16_fallbacks = ix->f16_skipped ? nq : 0; ix->last_fp16_escalated = 0; ix->f16_skipped = 0; }      // a plain fp32 search: nothing fell back (the counters are not left over from an earlier search)
    if (f16 && nq > 0 && ix->ntotal > 0) {
        // bring the fp16 copy of the bank fragment tiles up to date.  A finite value beyond the fp16 range (|x| > 65504) turns
        // into inf there and the scores into inf / NaN, which the exactness certificate cannot bound: such a bank stays on the
        // fp32 kernel (every query counts as a fallback)
        hipStream_t s0 = ix->stream;
        if (!ix->f16_flag) { HB_HIP(hipMalloc((void**)&ix->f16_flag, 4)); HB_HIP(hipMemsetAsync(ix->f16_flag, 0, 4, s0)); }
        if (ix->f16_cap_rows != ix->cap_rows) {
            if (ix->tiles16) HB_HIP(hipFree(ix->tiles16));
            ix->tiles16 = nullptr; ix->f16_rows = 0;
            HB_HIP(hipMalloc(&ix->tiles16, (size_t)ix->cap_rows * ix->dp16 * 2));
            HB_HIP(hipMemsetAsync(ix->tiles16, 0, (size_t)ix->cap_rows * ix->dp16 * 2, s0));
            ix->f16_cap_rows = ix->cap_rows;
        }
        if (ix->f16_rows < ix->ntotal) {
            const int64_t rt0 = ix->f16_rows / 32, need_rt = (ix->ntotal + 31) / 32;
            if (hb_launch_tiles_to_f16(ix->tiles, ix->g8, (_Float16*)ix->tiles16, ix->dp16 / 16, need_rt - rt0, rt0, ix->f16_flag, s0)) return -1;
            ix->f16_rows = ix->ntotal;
            HB_HIP(hipMemcpyAsync(&ix->f16_overflow, ix->f16_flag, 4, hipMemcpyDeviceToHost, s0));
            HB_HIP(hipStreamSynchronize(s0));
        }
        if (ix->f16_overflow) { f16 = false; if (esc == 0) { ix->last_fp16_fallbacks = nq; ix->last_fp16_escalated = 0; } }
        // ... and the row-major fp32 copy for the re-rank (hbird_knn_f16.hip).  Automatic: by the bank's size (below; a 10 M x 768 bank:
        // 30.7 GB of tiles + 15.4 GB of fp16 tiles + 30.7 GB of rows, of 288)
        if (f16 && ix->rerank_copy != 2) {
            const int rs = (ix->g8 * 8 + 31) / 32 * 32;
            if (ix->rows32 && (ix->rows32_cap_rows != ix->cap_rows || ix->rows32_rs != rs)) {
                HB_HIP(hipFree(ix->rows32));
                ix->rows32 = nullptr; ix->rows32_cap_rows = 0; ix->rows32_rows = 0;
            }
            if (!ix->rows32 && (ix->rerank_copy == 1 || ix->rows32_declined_cap != ix->cap_rows)) {
                const size_t need = (size_t)ix->cap_rows * rs * 4;
                size_t free_b = 0, total_b = 0;
                HB_HIP(hipMemGetInfo(&free_b, &total_b));
                ix->rows32_declined_cap = ix->cap_rows;     // (cleared below when the copy is made)
                // automatic (round 6, by measurement: profiles/r06/final/fp16_residency.json): what the copy saves is a few ms of re-rank per search
                // (about 4 ms for 21,904 queries x 64 candidates), whatever the bank's size, and what it costs is the bank once more.  At 300,000
                // x 768 that is 17 % of a search for 0.9 GB; at 10 M x 768 1.5 % for 30.7 GB, at 20 M x 1024 and 27.7 M x 768 0.4 % for 83-85 GB.
                // So only banks of up to 4e9 values (16 GB of fp32: 5.2 M x 768) get it -- a use_fp16 index of a bigger bank holds 1.5 x the bank
                // (fp32 tiles for the exact re-rank and the fp32 searches, fp16 tiles for the candidate pass), not 2.5 x -- and, as before, only
                // where the three copies stay within 55 % of the device with room to spare: the search's own workspace is allocated after the
                // copy, and a device shared with a model or another rank must not be filled to the brim by an optional copy.  An allocation that
                // fails all the same just means no copy.
                const size_t bank_b = (size_t)ix->cap_rows * ix->dp * 4;
                if (ix->rerank_copy == 1 || (bank_b <= (size_t)16e9 && bank_b + bank_b / 2 + need <= total_b / 100 * 55 &&
                                             free_b > need + std::max<size_t>(total_b / 16, (size_t)2 << 30))) {
                    if (hipMalloc((void**)&ix->rows32, need) == hipSuccess) { ix->rows32_cap_rows = ix->cap_rows; ix->rows32_rs = rs; ix->rows32_rows = 0; ix->rows32_declined_cap = -1; }
                    else { (void)hipGetLastError(); ix->rows32 = nullptr; if (ix->rerank_copy == 1) return hb_fail("hb_index_search: no memory for the re-rank copy of the bank"); }
                }
            }
            if (ix->rows32 && ix->rows32_rows < ix->ntotal) {
                const int64_t rt0 = ix->rows32_rows / 32, need_rt = (ix->ntotal + 31) / 32;
                if (hb_launch_tiles_to_rows(ix->tiles, ix->g8, ix->rows32, rs, need_rt - rt0, rt0, s0)) return -1;
                ix->rows32_rows = ix->ntotal;
            }
        }
    }
    // fp16 mode: the fused kernel collects kc >= 2k candidates, the fp32 chain arithmetic re-ranks them
    // k' = 2k, at least 64 (rounded up to 8, not to 64 as until round 4: the candidate kernel's time is linear in k' -- 300,000 x 768, 21,904
    // queries: k' = 64 / 128 / 192 / 256 -> 12.95 / 15.85 / 20.7 / 25.0 ms -- so k = 33 paid for 128 candidates where it needs 66)
    const int kc = f16 ? (esc == 1 || wide_first ? 256 : std::min(256, std::max(64, (2 * k + 7) / 8 * 8))) : k;     // (the second pass: the widest list the re-rank takes)
    // Small searches (few stages per workgroup) on the kernel with register-resident query fragments run on POOLS even for k <= 32:
    // phased, with the bisection cold start and the scan epilogue (hbird_knn_bd.hip <WIDE, COLD>) a pool takes a tile's survivors in one
    // drain, a sorted LDS list one wave-cooperative insertion each.  Same box, kernel ms, lists / pools, k = 30: 50,176 x 384 x 12,544
    // queries 4.61 / 4.13 (k = 32: 4.50 / 3.84), x 21,904 queries 7.04 / 6.52, 50,176 x 768 12.47 / 12.16, 200 k x 384 14.76 / 13.89,
    // 300 k x 768 40.6 / 39.9, 2,074,072 x 384 140.5 / 137.4, 600 k x 1024 105.8 / 105.3, 1.25 M x 768 286.0 / 286.7, 2.5 M x 768
    // 573.5 / 571.5, 20 k x 384 x 784 queries 0.59 / 0.26, 100 k x 384 x 196 queries 0.61 / 0.26; k = 5 at 50,176 x 384 3.61 / 3.68 and
    // k = 1 at 200 k x 384 13.47 / 13.54 (few insertions anyway) -> from k = 8.  (Round 2 measured pools at 8.1 vs 5.0 ms for the first
    // of these: unphased, radix cold start, LDS walk.)  Variant 6 keeps the lists (A/B, tests).
    const long long small_limit = ix->small_limit > 0 ? ix->small_limit : 400000;   // stages per workgroup (hb_index_set_search_options)
    const int G0 = ix->force_G > 0 ? ix->force_G : ix->num_cu;
    const long long pairs0 = (long long)((nq + HB_QT - 1) / HB_QT) * ((ix->ntotal + HB_BT - 1) / HB_BT);
    const bool small_shape = pairs0 / std::max<long long>(1, std::min<long long>(G0, pairs0)) * ix->g8 < std::min<long long>(small_limit, 120000);   // no gain beyond (1.25 M x 768: 157 k stages)
    const bool bd_shape = ix->g8 % 4 == 0 && ix->variant != 4;
    const bool small_pools = !f16 && !ceil && k >= 8 && k <= HB_KL && small_shape && bd_shape && (ix->variant == 0 || ix->variant == 3) && ix->force_cq <= 1;
    const bool wide = f16 || k > HB_KL || small_pools || ceil;
    // pools (k > HB_KL): capacity >= 2 kc so that a compaction is paid for by >= kc cheap appends
    // (smaller / larger pools measure the same on the fp16 candidate kernel: kc + 64, kc + 192)
    const int klw = wide ? std::min(HB_POOL_MAX, (std::max(2 * kc, kc + 128) + 63) / 64 * 64) : HB_KL;
    if (nq == 0) return 0;
    // score output (sharded searches): the ordering score goes out as it is, whatever the metric
    const int out_metric = ix->score_output ? 0 : ix->metric;
    const int nqt = (int)((nq + HB_QT - 1) / HB_QT);
    const int nbt = (int)((ix->ntotal + HB_BT - 1) / HB_BT);
    hipStream_t s = ix->stream;
    if (nbt == 0) {
        // empty index: every neighbour is missing (faiss returns -1 labels)
        std::vector<int64_t> hi((size_t)nq * k, -1);
        std::vector<float> hd((size_t)nq * k, out_metric == 1 ? INFINITY : -INFINITY);
        HB_HIP(hipMemcpyAsync(out_idx, hi.data(), hi.size() * 8, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(out_dist, hd.data(), hd.size() * 4, hipMemcpyHostToDevice, s));
        HB_HIP(hipStreamSynchronize(s));
        return 0;
    }
    const int G = ix->force_G > 0 ? ix->force_G : ix->num_cu;
    const size_t tile_bytes = (size_t)HB_BT * ix->dp * 4;
    // per-XCD work shares (hb_xcd_calibrate above; read BEFORE the cluster shape is chosen: the calibration also decides whether the fp32 clusters stay): calibrated for fp32 searches from 30,000 stages per workgroup (30-60 ms of kernel; from 150,000 until late in
    // round 5: cfg-2's 2 M x 384 bank went without, 135.3 -> 134.7 ms with; phased searches gain in their last phase only);
    // shares given by the caller (mode 2) apply to searches of any size, both kernel families (tests/fuzz_small.py FUZZ_XCD=1)
    const int fam = f16 ? 1 : 0;
    const bool balance = G % 8 == 0 &&
                         (ix->xcd_balance == 2 || (ix->xcd_balance == 0 && (long long)nqt * nbt / std::max(1, G) * ix->g8 >= 30000));
    if (balance && ix->xcd_balance == 0 && esc == 0) hb_xcd_calibrate(ix, fam);     // (nested searches run on the shares in use and leave the calibration alone)
    // L2-sharing clusters (hb_index_set_cluster; automatic shapes below): q x b workgroups of one XCD walk the same bank /
    // query tiles within `lag` stages of each other, so one L2 fill serves several.  Neither kernel is bound by the fabric
    // (the fp32 one by the matrix pipe, the fp16 candidate kernel by its LDS-DMA copies and the power the chip grants it:
    // profiles/LABBOOK.md, profiles/r02), so what they buy is traffic, and time only for the fp16 kernel (-8 %).  The 4-wave variant
    // does not know strided segments.
    int cq = 1, cb = 1;
    bool auto_cluster = false;      // fp32: the cluster shape of this search is the automatic choice (kept only where it measures faster)
    if (ceil) { cq = 1; cb = 1; }
    else if (ix->force_cq > 0 && ix->force_cb > 0) { cq = ix->force_cq; cb = ix->force_cb; }
    // fp16 candidate kernel: from 70 k stages per workgroup up (round 4: with the lean stage loop and the XCD-level query sharing the
    // clusters pay much earlier than the 400 k of round 3).  Same box, kernel ms (phased), none vs automatic: 10 M x 768 321 / 284,
    // 2.5 M x 768 (157 k stages) 83.0 / 77.5, 5 M x 384 (157 k) 85.3 / 80.3, 1.25 M x 768 (79 k) 43.7 / 41.8, 5 M x 768 x 12,544 queries
    // (179 k; 49 query tiles: 4 x 2) 94.0 / 88.2 -- but 2,074,072 x 384 (37 k) 21.3 / 22.8: more slots, shorter segments
    // (profiles/r04/f16_cluster_threshold.txt)
    else if (f16 && ix->variant == 0 && ix->force_cq == 0) {
        if ((long long)nqt * nbt / std::max(1, G) * (ix->dp16 / 16) >= 70000) hb_default_cluster(nqt, nbt, G, false, &cq, &cb);
    }
    // fp32: only beside the kernel with register-resident query fragments (its sync is free of spills), and only for the
    // biggest searches: 2 x 4 clusters cut the fabric reads by 60 % (10 M x 768: 4.79 -> 1.93 TB per search, L2 hit rate
    // 10 % -> 63 %) but the kernel is bound by the matrix pipe, so all they can do for the time is cost little -- measured
    // (same box, kernel ms, none vs 2 x 4): 10 M x 768 2280 vs 2298 (+0.8 %), 5 M x 1024 1528 vs 1531 (+0.2 %), but
    // 1.25 M x 768 289.3 vs 293.9 (+1.6 %), 2 M x 384 142.3 vs 146.4 (+2.9 %): more slots, shorter segments.  Automatic from
    // one million stages per workgroup up (8 M rows at D = 768); hb_index_set_cluster(ix, 1, 1, 0) turns them off, (ix, 2, 4, -1) forces them.
    // Round 6: ... and only where they MEASURE faster on this box (hb_xcd_calibrate: two calibrated launches with, two without, the faster
    // form stays); without the calibration's stamps (equal or given shares) they stay on.
    else if (!f16 && !wide && ix->variant == 0 && ix->force_cq == 0 && ix->g8 % 4 == 0 &&
             (long long)nqt * nbt / std::max(1, G) * ix->g8 >= 1000000) {
        auto_cluster = true;
        const hb_index::xcd_cal& c0 = ix->xcal[0];
        const bool measured = ix->xcd_balance == 0 && G % 8 == 0;
        // (a decision, once made, holds whatever the share mode; before it: on while measuring with, off while measuring without)
        if (c0.cl_state == 2 ? c0.cl_choice != 0 : (!measured || c0.cl_state == 0)) hb_default_cluster(nqt, nbt, G, true, &cq, &cb);
    }
    if ((long long)nqt * nbt < G || cq * cb > HB_CLUSTER_MAX || G % (8 * cq * cb) != 0) { cq = 1; cb = 1; }
    const int panel = ix->force_panel > 0 ? ix->force_panel
                                          : hb_default_panel(nqt, std::min<long long>(G, (long long)nqt * nbt), tile_bytes, cq, cb);
    static const double equal_shares[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    // (shares divided by their mean: eight equal shares of any size are the equal list, which is cached as such)
    double shares_n[8];
    const double* shares = equal_shares;
    if (balance) {
        double mean = 0.0;
        bool uneven = false;
        // calibrated shares belong to the physical XCDs: group g (blocks equal to g mod 8) gets the share of the XCD it was last seen on
        const hb_index::xcd_cal& xc = ix->xcal[fam];
        for (int x = 0; x < 8; ++x) mean += xc.w[x] / 8.0;
        for (int g = 0; g < 8; ++g) { shares_n[g] = xc.w[ix->xcd_balance == 0 ? xc.perm[g] : g] / mean; uneven = uneven || std::fabs(shares_n[g] - 1.0) > 1e-9; }
        if (uneven) shares = shares_n;
    }
    hb_schedule& sc = esc == 0 ? ix->sched : ix->sched_esc;      // (the nested searches of uncertified queries keep a list of their own: the caller's stays cached)
    char*& sched_dev = esc == 0 ? ix->sched_dev : ix->sched_esc_dev;
    size_t& sched_bytes = esc == 0 ? ix->sched_bytes : ix->sched_esc_bytes;
    // phased searches (pools only: "Phased searches" above hb_launch_knn); hb_index_set_search_options(ix, 0, ...) turns them off (A/B, tests)
    // (a nested search of uncertified queries starts from seeded floors: phases would only add boundaries -- and with one query tile over 256
    // workgroups the floors between them go through the merge kernels: 70 ms for two queries)
    const bool phased = wide && ix->phases_on && esc == 0;
    // XCD-level sharing of the query tiles (hb_build_clustered): automatic for the fp16 candidate kernel -- same box, 10 M x 768, 8 x 1
    // clusters: 302.7 -> 291.5 ms and 0.97 -> 0.52 TB of L2-miss traffic per search (L2 hit rate 0.60 -> 0.78); the fp32 kernel's 2 x 4
    // clusters lose 1.6 % with it (2298 -> 2334 ms: 1600 slots instead of 592, and its 768 KiB query tiles do not stay in L2 beside
    // sixteen bank streams anyway: 1.95 -> 1.62 TB) -> off there (profiles/r04/xs_*.txt)
    const bool xs = cq * cb > 1 && (ix->xcd_share == 2 || (ix->xcd_share == 0 && f16));
    const bool rebuilt = !(sc.nqt == nqt && sc.nbt == nbt && sc.panel == panel && sc.cq == cq && sc.cb == cb && sc.phased == phased &&
                           sc.xcd_share == xs && (sc.G == G || (long long)nqt * nbt < G) &&
                           (sc.xcd_w.empty() ? std::equal(shares, shares + 8, equal_shares) : std::equal(shares, shares + 8, sc.xcd_w.begin())));
    if (rebuilt) { hb_build_schedule(nqt, nbt, G, panel, sc, cq, cb, phased, xs, shares); ++ix->sched_builds; }
    // device copy of the work list: [segs][wg_off][qt_off][qt_slots][wg_member]
    const size_t b_segs = sc.segs.size() * sizeof(hb_seg), b_wg = sc.wg_off.size() * 4, b_qo = sc.qt_off.size() * 4,
                 b_qs = sc.qt_slots.size() * 4, b_wm = sc.wg_member.size() * 4;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t b_pb = sc.phase_bounds.size() * 4;
    const size_t o_wg = al(b_segs), o_qo = o_wg + al(b_wg), o_qs = o_qo + al(b_qo), o_wm = o_qs + al(b_qs), o_pb = o_wm + al(b_wm),
                 tot = o_pb + al(b_pb);
    const bool need_upload = rebuilt || sched_bytes < tot;
    if (ensure_bytes(&sched_dev, &sched_bytes, tot)) return -1;
    if (need_upload) {
        HB_HIP(hipMemcpyAsync(sched_dev, sc.segs.data(), b_segs, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(sched_dev + o_wg, sc.wg_off.data(), b_wg, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(sched_dev + o_qo, sc.qt_off.data(), b_qo, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(sched_dev + o_qs, sc.qt_slots.data(), b_qs, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(sched_dev + o_wm, sc.wg_member.data(), b_wm, hipMemcpyHostToDevice, s));
        HB_HIP(hipMemcpyAsync(sched_dev + o_pb, sc.phase_bounds.data(), b_pb, hipMemcpyHostToDevice, s));
        HB_HIP(hipStreamSynchronize(s));   // host vectors may be rebuilt by the next call
    }
    const size_t state_half = (size_t)sc.n_slots * HB_QT * klw * 4;
    const size_t state_aux = wide ? (size_t)sc.n_slots * HB_QT * 4 : 0;   // pools: fill counts + thresholds
    const size_t floor_bytes = (size_t)nqt * HB_QT * 4 * 17;              // shared threshold floors, one per query, + 16 quota-floor keys per query
    const size_t prog_bytes = ((size_t)std::max(1, sc.n_clusters) * HB_CLUSTER_MAX + 1) * HB_CLUSTER_LINE * 4;   // progress words, a line each, + statistics
    const size_t stamp_bytes = (size_t)sc.G * 32;                       // per-block {start, end, XCC id} stamps of the last kNN launch with its shader-cycle counts (wg_stamp, hbird_knn_dev.h)
    if (ensure_bytes(&ix->state, &ix->state_bytes, 2 * state_half + 2 * state_aux + floor_bytes + prog_bytes + stamp_bytes)) return -1;

    knn_args a;
    a.wg_stamp = (ix->time_kernels || (balance && ix->xcd_balance == 0)) ? reinterpret_cast<unsigned*>(ix->state + 2 * state_half + 2 * state_aux + floor_bytes + prog_bytes) : nullptr;
    ix->wg_stamp_dev = a.wg_stamp; ix->wg_stamp_blocks = sc.G;
    if (a.wg_stamp) HB_HIP(hipMemsetAsync(a.wg_stamp, 0, stamp_bytes, s));   // a block that never stamps reads 0 / 0 (hb_stamps_summarise)
    a.bank_tiles = ix->tiles; a.binit = ix->binit; a.q_tiles = ix->q_tiles;
    a.ceil_s = ix->ceil_s_dev; a.ceil_i = ix->ceil_i_dev;
    a.segs = reinterpret_cast<const hb_seg*>(sched_dev);
    a.wg_off = reinterpret_cast<const int*>(sched_dev + o_wg);
    a.wg_end = a.wg_off + 1;
    // this launch's share of every block's segments: [phase_begin(p)[b], phase_end(p)[b]) of the block's list
    const int n_phases = (int)sc.phase_clock.size() + 1;   // 1: a single launch (lists; pools with too little work per workgroup)
    const int* pb = reinterpret_cast<const int*>(sched_dev + o_pb);
    auto phase_begin = [&](int p) { return p == 0 ? reinterpret_cast<const int*>(sched_dev + o_wg) : pb + (size_t)(p - 1) * sc.G; };
    auto phase_end = [&](int p) { return p == n_phases - 1 ? reinterpret_cast<const int*>(sched_dev + o_wg) + 1 : pb + (size_t)p * sc.G; };
    a.state_s = reinterpret_cast<float*>(ix->state);
    a.state_i = reinterpret_cast<unsigned*>(ix->state + state_half);
    a.g8 = ix->g8; a.k = k; a.klw = klw;
    a.state_cnt = reinterpret_cast<int*>(ix->state + 2 * state_half);
    a.state_thr = reinterpret_cast<float*>(ix->state + 2 * state_half + state_aux);
    const int* pool_cnt = wide ? a.state_cnt : nullptr;
    a.gthr = reinterpret_cast<unsigned*>(ix->state + 2 * state_half + 2 * state_aux);
    a.qfl = a.gthr + (size_t)nqt * HB_QT;
    HB_HIP(hipMemsetD32Async((hipDeviceptr_t)a.gthr, 0x007FFFFF, (size_t)nqt * HB_QT * 17, s));   // key(-inf)
    // the padding queries of the last query tile (zero vectors: every row scores the same) start from key(+inf): nothing ever passes
    // their threshold.  Without it a slot's first tile appended all 256 tied rows for each of them and compacted their pools on the spot
    // (50,176 x 384 x 21,904 queries, 112 padding queries, pools: 943 us for a one-tile phase that takes 136 us with 21,760 queries)
    if (nq < (int64_t)nqt * HB_QT) HB_HIP(hipMemsetD32Async((hipDeviceptr_t)(a.gthr + nq), 0xFF800000u, (size_t)((int64_t)nqt * HB_QT - nq), s));
    // a nested search of uncertified queries starts from what the pass before it found out (hb_rerank_seeds): per query a score that the
    // rows which can still matter reach
    if (esc != 0 && ix->seed_dev) {
        seed_from_scores_kernel<<<dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s>>>(ix->seed_dev, nq, a.gthr);
        HB_HIP(hipGetLastError());
    }
    a.wg_member = reinterpret_cast<const int*>(sched_dev + o_wm);
    a.prog = reinterpret_cast<int*>(ix->state + 2 * state_half + 2 * state_aux + floor_bytes);
    a.cl = sc.cq * sc.cb;
    // soft-sync lag in stages: the members stay inside the L2's reach (4 MiB per XCD: tens of fp32 k8 stages); 0 disables
    // the sync (the members then share only while they happen to run together: 36 % instead of 53 % L2 hits)
    a.lag = a.cl > 1 ? (ix->sync_lag >= 0 ? ix->sync_lag : 16) : 0;
    a.cl_stats = a.prog + (size_t)std::max(1, sc.n_clusters) * HB_CLUSTER_MAX * HB_CLUSTER_LINE;
    ix->cl_stats_dev = a.cl > 1 ? a.cl_stats : nullptr;
    if (a.cl > 1) HB_HIP(hipMemsetAsync(a.prog, 0, prog_bytes, s));
    // between two phases of a pool search: the kk-th best of all rows seen so far becomes every slot's floor -- straight from the pools
    // where a query tile's pools fit the floor kernel's LDS, else through the merge (few queries against a big bank: many slots)
    auto seed_floors = [&](int kk, int64_t* scratch_idx, float* scratch_dist) -> int {
        const int* qo = reinterpret_cast<const int*>(sched_dev + o_qo);
        const int* qs = reinterpret_cast<const int*>(sched_dev + o_qs);
        const size_t per_wave = (size_t)sc.max_slots_per_qt * klw;
        // (up to 144 KiB of the CU's 160: 32 slots per query tile x pools of 256, 16 x 512 -- the merge path below costs a phase boundary
        // ten times as much: 10 M x 768, k = 90 with 16 slots per query tile: 2.9 -> 44.7 ms per search, profiles/r04/cluster_tail_rows_ab.txt)
        if (per_wave * 16 <= 144 * 1024) {
            if (per_wave * 16 > 48 * 1024 && hb_ensure_dyn_lds((const void*)pool_floor_kernel, 144 * 1024)) return -1;
            pool_floor_kernel<<<dim3((unsigned)((nq + 3) / 4)), dim3(256), per_wave * 16, s>>>(a.state_s, a.state_cnt, a.state_thr, qo, qs, nq, kk,
                                                                                               klw, (int)per_wave, a.gthr);
            HB_HIP(hipGetLastError());
            return 0;
        }
        if (launch_merge(ix, a.state_s, a.state_i, a.state_cnt, a.state_thr, qo, qs, sc.max_slots_per_qt, nqt, nq, kk, klw, 0, 0, nullptr,
                         scratch_idx, scratch_dist, s)) return -1;
        seed_floors_kernel<<<dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s>>>(scratch_idx, scratch_dist, nq, kk, a.gthr);
        HB_HIP(hipGetLastError());
        return 0;
    };
    if (f16) {
        // fp16 copy of the query fragment tiles (the bank's is up to date: top of this function)
        const int64_t nqp = (int64_t)nqt * HB_QT;
        if (ensure_bytes((char**)&ix->q16, &ix->q16_bytes, (size_t)nqp * ix->dp16 * 2)) return -1;
        if (hb_launch_tiles_to_f16(ix->q_tiles, ix->g8, (_Float16*)ix->q16, ix->dp16 / 16, nqp / 32, 0, nullptr, s)) return -1;
        if (ensure_bytes(&ix->cand, &ix->cand_bytes, (size_t)nq * kc * 12)) return -1;
        int64_t* cand_idx = reinterpret_cast<int64_t*>(ix->cand);
        float* cand_dist = reinterpret_cast<float*>(ix->cand + (size_t)nq * kc * 8);
        knn16_args h;
        h.wg_stamp = a.wg_stamp;
        h.bank16 = reinterpret_cast<const _Float16*>(ix->tiles16); h.binit = ix->binit; h.q16 = reinterpret_cast<const _Float16*>(ix->q16); h.segs = a.segs; h.wg_off = a.wg_off; h.wg_end = a.wg_end;
        h.state_s = a.state_s; h.state_i = a.state_i; h.g16 = ix->dp16 / 16; h.k = kc; h.klw = klw;
        h.state_cnt = a.state_cnt; h.state_thr = a.state_thr; h.gthr = a.gthr;
        h.wg_member = a.wg_member; h.prog = a.prog; h.cl = a.cl; h.lag = a.lag; h.cl_stats = a.cl_stats;
        if (ix->time_kernels) HB_HIP(hipEventRecord(ix->ev0, s));
        if (n_phases > 1) {   // slots that start in a later phase must read as empty pools in the merges between the phases
            HB_HIP(hipMemsetAsync(a.state_cnt, 0, state_aux, s));
            HB_HIP(hipMemsetD32Async((hipDeviceptr_t)a.state_thr, 0xFF800000u, state_aux / 4, s));
        }
        for (int ph = 0; ph < n_phases; ++ph) {
            h.wg_off = phase_begin(ph); h.wg_end = phase_end(ph);
            if (hb_knn_f16_launch(h, sc.G, s)) return -1;
            if (ph + 1 < n_phases) {   // the k'-th best of all rows seen so far -> every slot's floor
                if (seed_floors(kc, cand_idx, cand_dist)) return -1;
                if (a.cl > 1) HB_HIP(hipMemsetAsync(a.prog, 0, prog_bytes - HB_CLUSTER_LINE * 4, s));   // progress words (not the statistics)
            }
        }
        if (ix->time_kernels) HB_HIP(hipEventRecord(ix->ev1, s));
        if (balance && ix->xcd_balance == 0 && esc == 0 && hb_xcd_collect(ix, 1, a.wg_stamp, sc, shares, n_phases, nqt, nbt, kc, s)) return -1;
        if (launch_merge(ix, a.state_s, a.state_i, pool_cnt, pool_cnt ? a.state_thr : nullptr, reinterpret_cast<const int*>(sched_dev + o_qo),
                         reinterpret_cast<const int*>(sched_dev + o_qs), sc.max_slots_per_qt, nqt, nq, kc, klw, 0, 0, nullptr,
                         cand_idx, cand_dist, s)) return -1;
        // workspace of this level: [certificates nq + 64][exact k-th scores nq][floors for a second pass nq] and, once the failures are
        // known, [their rows][queries][aux][floors][ids][distances]
        char*& fbuf = esc == 0 ? ix->fb : ix->fb1;
        size_t& fbuf_bytes = esc == 0 ? ix->fb_bytes : ix->fb1_bytes;
        auto al2 = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t o_kth = al2((size_t)nq + 64), o_flo = o_kth + al2((size_t)nq * 4), o_rows = o_flo + al2((size_t)nq * 4);
        if (ensure_bytes(&fbuf, &fbuf_bytes, o_rows)) return -1;
        unsigned char* cert = reinterpret_cast<unsigned char*>(fbuf);
        float* kth = reinterpret_cast<float*>(fbuf + o_kth);
        float* flo = reinterpret_cast<float*>(fbuf + o_flo);
        HB_HIP(hipMemsetD32Async((hipDeviceptr_t)kth, 0xFF800000u, (o_rows - o_kth) / 4, s));    // -inf: no seed (a query with fewer than k candidates)
        const float* seed_in = esc == 1 ? ix->seed_dev : nullptr;
        if (ix->rows32 && ix->rerank_copy != 2 && ix->rows32_rows >= ix->ntotal) {
            if (hb_launch_rerank_rows(ix->rows32, ix->rows32_rs, ix->binit, ix->d, q_dev, ix->q_aux, cand_idx, cand_dist, ix->q_aux + nq, ix->bmax,
                                      cert, kc, nq, k, id_base, ix->metric, out_metric, ix->ntotal, out_idx, out_dist, s, seed_in, kth, flo)) return -1;
        } else if (hb_launch_rerank(ix->tiles, ix->binit, ix->g8, ix->d, q_dev, ix->q_aux, cand_idx, cand_dist, ix->q_aux + nq, ix->bmax,
                                    cert, kc, nq, k, id_base, ix->metric, out_metric, ix->ntotal, out_idx, out_dist, s, seed_in, kth, flo)) return -1;
        if (ix->time_kernels) {
            HB_HIP(hipEventSynchronize(ix->ev1));
            float ms = 0.f;
            HB_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
            ix->last_knn_ms = ms;
        }
        // Queries whose certificate failed are searched again.  ESCALATION (round 6): first by a second fp16 pass with k' = 256 candidates
        // (the certificate compares the exact k-th best with the fp16 score of rank k': four times the ranks apart) whose pools start from
        // the floor `kth - 1.001 E` -- every row that can still matter scores above it in fp16, few others do, so the pass appends little and
        // a list that does not fill up is complete by construction; only what fails again goes to the exact fp32 kernel, which starts from
        // the exact k-th best found so far as its floor.  A failing query used to cost a share of a whole-bank fp32 search (10 M x 768: 27 ms
        // per started tile of 256 queries; 5 % failing queries = +45 % on the step): the second pass costs a twentieth of that per query.
        std::vector<unsigned char> hc((size_t)nq);
        HB_HIP(hipMemcpyAsync(hc.data(), cert, (size_t)nq, hipMemcpyDeviceToHost, s));
        HB_HIP(hipStreamSynchronize(s));
        std::vector<int64_t> bad;
        for (int64_t i = 0; i < nq; ++i) if (!hc[i]) bad.push_back(i);
        const int64_t nf = (int64_t)bad.size();
        if (esc == 0) { ix->last_fp16_escalated = 0; ix->last_fp16_fallbacks = 0; }
        if (esc == 0 && nf == 0) hb_f16_observe(ix->f16_adapt, how16, nq, 0, 0);
        if (nf > 0) {
            // the second pass needs k' = 256 > the first one's, a bank worth a candidate pass, and is not repeated
            const bool again16 = esc == 0 && ix->fp16_escalation == 0 && kc < 256 && ix->ntotal >= 4096;
            const size_t o_q = o_rows + al2((size_t)nf * 8), o_aux = o_q + al2((size_t)nf * ix->d * 4), o_seed = o_aux + al2((size_t)nf * 8),
                         o_idx = o_seed + al2((size_t)nf * 4), o_dist = o_idx + al2((size_t)nf * k * 8), tot2 = o_dist + al2((size_t)nf * k * 4);
            if (fbuf_bytes < tot2) {
                char* nb = nullptr;
                HB_HIP(hipMalloc((void**)&nb, tot2 + tot2 / 4));
                HB_HIP(hipMemcpyAsync(nb, fbuf, o_rows, hipMemcpyDeviceToDevice, s));      // (the seeds of this level)
                HB_HIP(hipStreamSynchronize(s));
                HB_HIP(hipFree(fbuf));
                fbuf = nb; fbuf_bytes = tot2 + tot2 / 4;
                kth = reinterpret_cast<float*>(fbuf + o_kth); flo = reinterpret_cast<float*>(fbuf + o_flo);
            }
            int64_t* d_rows = reinterpret_cast<int64_t*>(fbuf + o_rows);
            float* d_q = reinterpret_cast<float*>(fbuf + o_q);
            float* d_aux = reinterpret_cast<float*>(fbuf + o_aux);
            float* d_seed = reinterpret_cast<float*>(fbuf + o_seed);
            int64_t* d_fi = reinterpret_cast<int64_t*>(fbuf + o_idx);
            float* d_fd = reinterpret_cast<float*>(fbuf + o_dist);
            HB_HIP(hipMemcpyAsync(d_rows, bad.data(), (size_t)nf * 8, hipMemcpyHostToDevice, s));
            if (hb_launch_gather_rows(q_dev, nq, ix->d, d_rows, nf, d_q, s)) return -1;
            if (hb_launch_gather_rows(again16 ? flo : kth, nq, 1, d_rows, nf, d_seed, s)) return -1;
            if (hb_launch_rows_to_tiles(d_q, nf, ix->d, ix->dp, 0, ix->q_tiles, nullptr, nullptr, ix->metric, 0, 0, s)) return -1;
            float* saved_aux = ix->q_aux;
            ix->q_aux = d_aux;                       // chain ||q||^2 of the re-searched queries (L2 distances)
            int rc = hb_launch_query_aux(d_q, nf, ix->d, d_aux, d_aux + nf, s);
            const int saved = ix->fp16, saved_t = ix->time_kernels, saved_esc = ix->esc_level;
            const float* saved_seed = ix->seed_dev;
            ix->fp16 = again16 ? 1 : 0; ix->time_kernels = 0; ix->esc_level = again16 ? 1 : 2; ix->seed_dev = d_seed;
            if (!again16) ix->last_fp16_fallbacks = nf;
            if (!rc) rc = hb_launch_knn(ix, d_q, nf, k, id_base, d_fi, d_fd);
            ix->fp16 = saved; ix->time_kernels = saved_t; ix->q_aux = saved_aux; ix->esc_level = saved_esc; ix->seed_dev = saved_seed;
            if (esc == 0) { ix->last_fp16_escalated = nf; hb_f16_observe(ix->f16_adapt, how16, nq, nf, ix->last_fp16_fallbacks); }
            if (rc) return -1;
            if (hb_launch_scatter_rows(d_rows, nf, k, d_fi, d_fd, out_idx, out_dist, s)) return -1;
            HB_HIP(hipStreamSynchronize(s));         // `bad` and the workspace are reused by the next call
        }
        return 0;
    }
    typedef void (*knn_fn)(knn_args);
    knn_fn fn = wide ? (knn_fn)knn_fused_kernel<false, true> : (knn_fn)knn_fused_kernel<false, false>;
    if (a.cl > 1) fn = wide ? (knn_fn)knn_fused_kernel<false, true, true> : (knn_fn)knn_fused_kernel<false, false, true>;
    // Few stages per workgroup: a slot sees few rows, so its cold start (the first tile inserts all 256 rows of every
    // query) and its insertions (k ln(rows / k) per query) are a visible share of the search -> the instantiations with the
    // cold start, the scan epilogue (register queue + immediate inserts) and the per-tile exchange of
    // threshold floors (hbird_knn_dev.h: small_floor_*).  Same box, kernel ms, LDS-staged small / B-direct plain / B-direct small:
    // 50,176 x 384: 4.86 / 6.76 / 4.62 (round 1: 6.3; 0.49 -> 0.665 of the fp32 MFMA peak); 200 k x 384: 15.9 / 17.7 / 15.1;
    // 300 k x 768: 74.7 / 72.8 / 70.7; 2 M x 384: 149.3 / 143.1 / 142.3; 1.25 M x 768: 305.9 / 291.2 / 289.8; 2.5 M x 768 (315 k
    // stages per workgroup): 612.7 / 579.1 / 579.6 -> small below 400 k stages.  The big searches keep the plain
    // instantiations: at 10 M x 768 the extra code costs 0.3 % (same-box A/B).
    static const knn_fn cold_fn = knn_fused_kernel<true, false>;
    // lists: cold_fn / <false, false, COLD>; pools: <WIDE, false, COLD> -- for k > 32 only below 50 k stages (k = 90: 50,176 x 384 4.78 -> 4.45 ms,
    // k = 64 at 300 k x 768 41.9 -> 40.3, but 2,074,072 x 384 (74 k stages) 142.0 -> 142.7)
    const long long stages_per_wg = (long long)nqt * nbt / std::max(1, sc.G) * ix->g8;
    const bool small = !f16 && a.cl == 1 && stages_per_wg < (k > HB_KL ? std::min<long long>(small_limit, 50000) : small_limit);
    if (small && !wide) fn = cold_fn;
    const int threads = HB_THREADS;
    int lds_bytes = fn == cold_fn ? KN_LDS_TOTAL_COLD : KN_LDS_TOTAL;
    // The query fragments straight into registers (hbird_knn_bd.hip): -3.8 % kernel time at 10 M x 768 (0.895 -> 0.93 of the
    // fp32 MFMA peak), same bits.  Default for the big LDS-list searches whose stage count per tile is a multiple of four
    // (D = 384, 768, 1024, ...); variant 3 forces it wherever it applies (tests), variant 4 keeps the LDS-staged kernel.
    if (ceil) { fn = (knn_fn)knn_fused_kernel<false, true, false, true>; lds_bytes = KN_LDS_TOTAL; }
    else if (bd_shape && (ix->variant == 0 || ix->variant == 3 || ix->variant == 6)) {
        fn = hb_knn_bd_kernel(wide, a.cl > 1, small);
        lds_bytes = hb_knn_bd_lds_bytes(small && !wide);
    }
    if (hb_ensure_dyn_lds((const void*)fn, lds_bytes)) return -1;   // per (kernel, device)
    if (ix->time_kernels) HB_HIP(hipEventRecord(ix->ev0, s));
    if (n_phases > 1) {   // (pools only) slots that start in a later phase must read as empty pools in the merges between the phases
        HB_HIP(hipMemsetAsync(a.state_cnt, 0, state_aux, s));
        HB_HIP(hipMemsetD32Async((hipDeviceptr_t)a.state_thr, 0xFF800000u, state_aux / 4, s));
    }
    for (int ph = 0; ph < n_phases; ++ph) {
        a.wg_off = phase_begin(ph); a.wg_end = phase_end(ph);
        fn<<<dim3((unsigned)sc.G), dim3(threads), lds_bytes, s>>>(a);
        HB_HIP(hipGetLastError());
        if (ph + 1 < n_phases) {   // the k-th best ORDERING score of all rows seen so far -> every slot's floor (the outputs serve as scratch)
            if (seed_floors(k, out_idx, out_dist)) return -1;   // (the outputs serve as scratch)
            if (a.cl > 1) HB_HIP(hipMemsetAsync(a.prog, 0, prog_bytes - HB_CLUSTER_LINE * 4, s));
        }
    }
    if (ix->time_kernels) HB_HIP(hipEventRecord(ix->ev1, s));
    if (balance && ix->xcd_balance == 0 && esc == 0 && hb_xcd_collect(ix, 0, a.wg_stamp, sc, shares, n_phases, nqt, nbt, k, s, auto_cluster)) return -1;
    const float* qn2 = ix->q_aux;   // [nq] chain ||q||^2 (valid for L2)
    if (launch_merge(ix, a.state_s, a.state_i, pool_cnt, pool_cnt ? a.state_thr : nullptr, reinterpret_cast<const int*>(sched_dev + o_qo),
                     reinterpret_cast<const int*>(sched_dev + o_qs), sc.max_slots_per_qt, nqt, nq, k, klw, id_base, out_metric,
                     qn2, out_idx, out_dist, s)) return -1;
    if (ix->time_kernels) {
        HB_HIP(hipEventSynchronize(ix->ev1));
        float ms = 0.f;
        HB_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
        ix->last_knn_ms = ms;
    }
    return 0;
}
