// Device-side helpers shared by the kNN kernels (hbird_knn.hip, hbird_knn_bd.hip, hbird_knn_f16.hip).
#pragma once
#include "hbird_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

struct knn_args_pool_view { int* cnt; float* thr; };

struct knn_args {
    const float* bank_tiles;
    const float* binit;
    const float* q_tiles;
    const hb_seg* segs;
    const int* wg_off;    // per block: first segment of this launch ...
    const int* wg_end;    // ... and one past its last (one launch: the block's whole list; phased searches: a part of it)
    float* state_s;
    unsigned* state_i;
    int g8;   // Dp / 8
    int k;
    int klw;  // row stride in the state buffer: HB_KL (sorted lists), or the pool capacity when k > HB_KL
    int* state_cnt;     // pools only: fill count and threshold per (slot, query), kept between segments
    float* state_thr;
    unsigned* gthr;     // [query]: shared threshold floor (monotone key of a score that k rows are known to reach)
    const int* wg_member;   // per block: progress word of the block in `prog` (cluster * HB_CLUSTER_LINE + member)
    int* prog;              // cluster progress words (stage clocks), zeroed before the launch
    int cl;                 // workgroups per cluster (1: no clusters)
    int lag;                // soft sync: a member waits while another one is more than `lag` stages behind (0: never)
    int* cl_stats;          // {checks, spins, timeouts} of the launch
    unsigned* qfl;          // [query][16]: quota floors of small searches (monotone keys; columns 0-6 / 8-14 per slot of the query tile, 7 the plain floor)
    unsigned* wg_stamp;     // diagnostics (hb_index_set_timing) and share calibration: [block][2][4], see wg_stamp(); nullptr: off
    const float* ceil_s;    // k > 256 (hb_launch_knn_bigk): per query the ordering key (score, row) of the last neighbour the passes before
    const unsigned* ceil_i; // have delivered -- only rows strictly behind it take part in this pass (knn_fused_kernel<.., CEIL>); nullptr: off
};

// A kernel argument read again from the kernarg segment at the point of use (through a laundered pointer, so that the
// compiler cannot keep it in an SGPR from the kernel's start): for arguments used only at segment / tile boundaries, which
// would otherwise crowd the stage loop's own scalars into spills that are reloaded in every stage.
#define HB_KARG(ARGS_T, FIELD) hb_karg<decltype(ARGS_T::FIELD)>(offsetof(ARGS_T, FIELD))
template <class T>
__device__ __forceinline__ T hb_karg(size_t offset) {
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<const T __attribute__((address_space(4)))*>(p + offset);
}

// arguments of the fp16 candidate kernel (hbird_knn_f16.hip); the fields shared with knn_args mean the same
struct knn16_args {
    const _Float16* bank16;   // fp16 copies of the bank fragment tiles
    const float* binit;
    const _Float16* q16;      // fp16 copies of the query fragment tiles
    const hb_seg* segs;
    const int* wg_off;
    const int* wg_end;
    float* state_s;
    unsigned* state_i;
    int g16;   // Dp16 / 16
    int k;     // k' (candidates per query)
    int klw;   // pool capacity
    int* state_cnt;
    float* state_thr;
    unsigned* gthr;
    const int* wg_member;
    int* prog;
    int cl;
    int lag;
    int* cl_stats;
    unsigned* wg_stamp;
};

// Diagnostics: when and where a workgroup ran (per-XCD speed differences show up as the last blocks of every launch belonging to one XCD),
// and at which clock: s_memtime counts shader cycles, s_memrealtime 10 ns ticks, so (d cycles / d ticks) x 100 MHz is the clock the
// workgroup's CU held over the launch -- read without any profiler attached (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Two stamps per
// workgroup and launch, outside every loop: [block][which] = {s_memrealtime (low word), XCC id, s_memtime lo, hi}, 16 bytes in ONE store.
// The store is inline asm WITHOUT a memory clobber (nothing in the kernel reads the stamps), and there is one per stamp: the variants with
// several stores, or with a table offset read from the arguments, made hipcc turn the segment loop's work-list loads into vector loads,
// whose values the stage loops feed to "s" operands ("illegal VGPR to SGPR copy").
template <class ARGS>
__device__ __forceinline__ void wg_stamp(int which) {
    unsigned* st = HB_KARG(ARGS, wg_stamp);
    if (st == nullptr || threadIdx.x != 0) return;
    st += 8 * blockIdx.x + 4 * which;
    unsigned long long cyc, rt;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc), "=s"(rt));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    v[0] = (unsigned)rt; v[1] = xcc & 0xFu; v[2] = (unsigned)cyc; v[3] = (unsigned)(cyc >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(st), "v"(v));
}

// ---- soft sync of an L2-sharing cluster (hb_build_clustered) ---------------------------------------------------------
// The members of a cluster run the same stage sequence; one L2 fill serves several of them only while they stay within
// a few stages of each other (an XCD's 4 MiB L2 turns over in tens of stages).  They mostly do by themselves (same start,
// same work); what drifts them apart is slow (epilogue insertions, memory jitter).  So every member publishes its stage
// clock every 4 stages (one fire-and-forget store to a 128-B line of its own), looks at the others once per HB_CL_PERIOD
// stages (one LDS-DMA load of the members' words into a few words of LDS, read a few stages later so that its latency is
// never waited for), and holds back only while the slowest member is more than `lag` stages behind.  No data is handed
// over, so nothing depends on it: the spin is bounded, and a member that times out stops syncing for the rest of the
// launch.
// What it costs was measured on the fp32 kernel (10 M x 768, one box, kernel ms; plain list 2360-2380):
//  * loads / stores the compiler tracks: it waits for them at the next control-flow merge -- a write-through round trip
//    per exchange, +3..5 %.  Hence an inline-asm store and an LDS-DMA load (both invisible to the wait-count pass).
//  * issued by wave 4..7 (the waves without copies): +3.4 % at one store per 4 stages, wherever it sits in the stage --
//    a vector-memory instruction queues behind the CU's LDS-DMA copies, the wave stalls on its issue, and a wave and
//    its SIMD partner stalled together idle the matrix pipe.  Issued by wave 0, which already pays that price for its
//    copies and whose partner covers it, AHEAD of the stage's copies (so that the hand-counted vmcnt still holds):
//    +1.9 % without any sharing, +0.3 % net with it.
#define HB_CL_PERIOD 32
#define HB_CL_SPINS 8192    // re-polls before a member gives up waiting (each about 0.5-1 us); 1024 let 20 of 256 members of the
                            // fp16 kernel give up during the slots' cold starts (10 M x 768: 330.8 -> 328.2 ms with 8192)
struct cl_sync {
    int* line;      // progress words of the cluster, one 128-B line per member
    int* lds;       // HB_CLUSTER_MAX words of LDS: landing zone of the poll
    int me, cl, lag;
    int n_checks, n_spins, n_timeouts;   // statistics of this workgroup (hb_index_cluster_stats)
    bool on;
};
__device__ __forceinline__ void cl_store(int* p, int v) {
    asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void cl_poll(const cl_sync& cs, int lane) {
    // lanes < cl: member `lane`'s progress word -> lds[lane] (LDS-DMA: wave-uniform LDS base + 4 * lane; sc1 = past the L1)
    if (lane < cs.cl) __builtin_amdgcn_global_load_lds((gbl_cvoid*)(cs.line + lane * HB_CLUSTER_LINE), (lds_void*)cs.lds, 4, 0, 16);
}
__device__ __forceinline__ int cl_min_landed(const cl_sync& cs, int lane, bool wait) {
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int m = lane < cs.cl ? cs.lds[lane] : 0x7FFFFFFF;
#pragma unroll
    for (int o = 1; o < HB_CLUSTER_MAX; o <<= 1) m = min(m, __shfl_xor(m, o));
    return __builtin_amdgcn_readfirstlane(m);
}
// Called by wave 0, once per stage (acts every 4th), ahead of the stage's copies.
template <int PERIOD = HB_CL_PERIOD>   // stages between two looks at the other members (a power of two >= 16)
__device__ __forceinline__ void cl_tick(cl_sync& cs, int now, int lane) {
    if (!cs.on || (now & 3) != 0) return;
    if (lane == 0) cl_store(cs.line + cs.me * HB_CLUSTER_LINE, now);
    const int ph = now & (PERIOD - 1);
    if (ph == 4) cl_poll(cs, lane);
    else if (ph == 12) {
        // the poll was issued two exchanges ago, ahead of that stage's copies: the per-stage vmcnt wait has covered it
        // (keep this loop wave-uniform: with a divergent statement in it -- `if (lane == 0) <count in LDS>` -- hipcc structurised it into
        // nested exec-masked loops whose inner spin no longer re-polled, and every 4-member cluster timed out on stale words)
        bool wait = false;
        int spins = 0;
        ++cs.n_checks;
        while (cl_min_landed(cs, lane, wait) < now - cs.lag) {
            wait = true;
            ++cs.n_spins;
            if (++spins > HB_CL_SPINS) { cs.on = false; ++cs.n_timeouts; break; }       // a member is not running (or not visible): never wait again
            __builtin_amdgcn_s_sleep(8);
            cl_poll(cs, lane);
        }
    }
}
__device__ __forceinline__ void cl_publish(const cl_sync& cs, int clock, int lane) {
    if (cs.cl > 1 && lane == 0) cl_store(cs.line + cs.me * HB_CLUSTER_LINE, clock);
}
__device__ __forceinline__ cl_sync cl_init(const int* wg_member, int* prog, int cl, int lag, int block, bool poller, char* lds_words) {
    cl_sync cs;
    const int mem = cl > 1 ? wg_member[block] : -1;
    cs.line = prog + (mem < 0 ? 0 : (mem & ~(HB_CLUSTER_LINE - 1)) * HB_CLUSTER_MAX);   // a 128-B line per MEMBER
    cs.lds = reinterpret_cast<int*>(lds_words);
    cs.me = mem < 0 ? 0 : mem & (HB_CLUSTER_LINE - 1);
    cs.cl = cl; cs.lag = lag;
    cs.n_checks = cs.n_spins = cs.n_timeouts = 0;
    cs.on = mem >= 0 && lag > 0 && poller;
    return cs;
}
// end of the kernel: add this workgroup's statistics to the launch's (three words behind the progress lines)
__device__ __forceinline__ void cl_finish(const cl_sync& cs, int* stats, bool poller, int lane) {
    if (cs.cl > 1 && poller && lane == 0 && cs.n_checks) {
        atomicAdd(stats, cs.n_checks); atomicAdd(stats + 1, cs.n_spins); atomicAdd(stats + 2, cs.n_timeouts);
    }
}

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

// ---- candidate pools (k > HB_KL, and the fp16 candidate pass) ------------------------------------------------
// A query's running best-k does not fit the LDS lists, so it lives in global memory as an UNSORTED pool of
// `cap` entries (cap >= 2k, multiple of 64): a score above the query's threshold is simply appended (one store per
// lane, all 32 queries of a wave in parallel, the fill count in LDS).  When a pool is full the wave compacts it to
// about its best k -- every entry above a cut that at least k entries exceed, found by bisection over the wave (pool_compact) -- and
// raises the threshold to that cut.  The threshold is therefore only as fresh as the last compaction: a few more scores pass than with
// sorted lists, but an append costs one store instead of a serialised read-modify-write of the list.
// Only the owning wave ever touches a pool; pool loads bypass the L1 and follow an s_waitcnt vmcnt(0).

// monotone key: larger float <=> larger unsigned
__device__ __forceinline__ unsigned pool_key(float s) {
    const unsigned u = __builtin_bit_cast(unsigned, s + 0.0f);   // -0 and +0 share a key
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// kk-th largest (1-based) of the keys flagged active, E entries per lane; wave-uniform result
template <int EMAX>
__device__ __forceinline__ unsigned pool_kth(const unsigned (&key)[EMAX], bool (&act)[EMAX], int E, int kk) {
    unsigned prefix = 0;
    for (int b = 31; b >= 0; --b) {
        const unsigned bit = 1u << b;
        int c1 = 0;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) c1 += __popcll(__ballot(act[e] && (key[e] & bit)));
        const bool take = c1 >= kk;
        if (take) prefix |= bit; else kk -= c1;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) act[e] = act[e] && (((key[e] & bit) != 0) == take);
    }
    return prefix;
}

// Compact the full pool (cap entries, cap <= 64 EMAX) at (gs, gi): afterwards entries [0, n) hold every entry above a new threshold t,
// k <= n <= k + (cap - k) / 4 wherever the scores allow it; returns (n << 32) | bits of t.  `thr` = the query's current threshold (every
// entry that matters exceeds it; entries at or below it -- a floor raised since they were appended -- are dropped, and if fewer than k
// remain, that is the whole compaction).  A pool is a SUPERSET of its query's best k of the rows its slot has seen: it need not hold
// exactly k after a compaction, only never lose one of them, and its threshold need only be a lower bound on its k-th best.  So t comes
// from a bisection over the monotone keys between the threshold and the largest one (a compare, a ballot and a count per entry register
// and round; it stops as soon as the count above the midpoint is within the slack: five to eight rounds) instead of the exact radix
// select (32 rounds, and 32 more among tied scores) that this function ran until round 4 -- that one is still the way out when the
// bisection does not get there in 24 rounds (a pool of equal scores).  300,000 x 768, k = 90, use_fp16: compactions were 5 % of the wave
// cycles (profiles/r04/f16_epilogue_counts.txt).
// (A variant for partly filled pools -- per-lane validity masks -- cost the fp32 pool kernel its scalar registers:
// reloads of spilled SGPRs in every stage, +10 % kernel time at k = 90.)
// NOT inlined (round 4): a real function call on the rare compaction path.  Inlined at every call site its 4 x EMAX value registers and
// the select's masks set the register budget of the whole kernel -- spilled SGPRs reloaded in the stage loops (fp16 candidate
// kernel, pools of 384: 373 -> 328 ms at 10 M x 768, k = 90; every pool instantiation lost half of its spills or more).
__device__ __forceinline__ float pool_unkey(unsigned key) {   // the float whose pool_key is `key`
    return __builtin_bit_cast(float, (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}
template <int EMAX = HB_POOL_MAX / 64>
__device__ __attribute__((noinline)) unsigned long long pool_compact(float* gs, unsigned* gi, int cap, int k, int lane, float thr) {
    const int E = cap >> 6;
    float es[EMAX];
    unsigned ei[EMAX], key[EMAX];
    bool act[EMAX];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's appends have landed
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        es[e] = 0.f; ei[e] = 0; key[e] = 0; act[e] = false;
        if (e < E) {
            es[e] = __hip_atomic_load(gs + e * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ei[e] = __hip_atomic_load(gi + e * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            key[e] = pool_key(es[e]);
            act[e] = true;
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    // keys above `cut` survive: count them ...
    auto above_cut = [&](unsigned cut) {
        int c = 0;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) c += __popcll(__ballot(key[e] > cut));
        return c;
    };
    // ... and move them to the front
    auto keep_above = [&](unsigned cut) {
        int base = 0;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const bool keep = key[e] > cut;
                const unsigned long long m = __ballot(keep);
                if (keep) { const int p = base + __popcll(m & lt); gs[p] = es[e]; gi[p] = ei[e]; }
                base += __popcll(m);
            }
        return base;
    };
    const int slack = (cap - k) >> 2;
    unsigned lo = pool_key(thr);       // -inf for a query without a threshold yet: below every score a pool can hold
    int clo = above_cut(lo);
    if (clo > k + slack) {
        unsigned hi = 0xFFFFFFFFu;     // invariant: at least k keys above lo, fewer than k above hi
        for (int r = 0; r < 24 && hi - lo > 1u; ++r) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            const int c = above_cut(mid);
            if (c >= k) { lo = mid; clo = c; if (c <= k + slack) break; }
            else hi = mid;
        }
    }
    if (clo <= k + slack) {
        const int n = keep_above(lo);
        return ((unsigned long long)(unsigned)n << 32) | (unsigned long long)__builtin_bit_cast(unsigned, fmaxf(thr, pool_unkey(lo)));
    }
    // the exact way: the best k by (score desc, id asc)
    const unsigned kt = pool_kth<EMAX>(key, act, E, k);   // key of the k-th best score
    int above = 0, tied = 0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) { above += __popcll(__ballot(key[e] > kt)); tied += __popcll(__ballot(key[e] == kt)); }
    unsigned id_cut = 0xFFFFFFFFu;   // among the tied scores the lower ids stay
    if (above + tied > k) {
        unsigned ik[EMAX];
#pragma unroll
        for (int e = 0; e < EMAX; ++e) { ik[e] = ~ei[e]; act[e] = (e < E) && key[e] == kt; }
        id_cut = ~pool_kth<EMAX>(ik, act, E, k - above);
    }
    float kth = 0.f;
    int base = 0;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) {
            const bool keep = key[e] > kt || (key[e] == kt && ei[e] <= id_cut);
            const unsigned long long m = __ballot(keep);
            if (keep) { const int p = base + __popcll(m & lt); gs[p] = es[e]; gi[p] = ei[e]; }
            base += __popcll(m);
            const unsigned long long mk = __ballot(key[e] == kt);
            if (mk) kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, es[e]), __builtin_ctzll(mk)));
        }
    return ((unsigned long long)(unsigned)k << 32) | (unsigned long long)__builtin_bit_cast(unsigned, kth);
}

// Append the register queues of a wave (at most four (score, row code) entries per lane; np = how many this lane holds) to the
// pools of its 32 queries: queue entry i of every lane, one lane half at a time (the two halves hold different rows of the SAME
// query and would race on its fill count; an LDS-atomic variant that drained both at once measured the same and needed a compaction
// of partly filled pools).  A pool that fills up is compacted on the spot (pool_compact: to k .. k + slack entries), which raises the query's threshold.
template <int EMAX>
__device__ __forceinline__ void pool_drain(int np, float q0v, float q1v, float q2v, float q3v, int q0c, int q1c, int q2c, int q3c, float& thr,
                                           float* pool_s, unsigned* pool_i, int qb, int lane, int k, unsigned row0, int klw, int* cnt) {
    const int myq = qb + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool has = np > i;
        if (__ballot(has) == 0ull) break;
        const float v = i == 0 ? q0v : i == 1 ? q1v : i == 2 ? q2v : q3v;
        const int code = i == 0 ? q0c : i == 1 ? q1c : i == 2 ? q2c : q3c;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const bool pass = has && (lane >> 5) == hh;
            if (__ballot(pass) == 0ull) continue;
            int c = 0;
            if (pass) {
                c = cnt[myq];
                pool_s[(size_t)myq * klw + c] = v;
                pool_i[(size_t)myq * klw + c] = row0 + (unsigned)code;
                cnt[myq] = c + 1;
            }
            unsigned long long full = __ballot(pass && c + 1 == klw);
            while (full) {
                const int n = __builtin_ctzll(full) & 31;
                full &= full - 1;
                const size_t off = (size_t)(qb + n) * klw;
                const float tq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thr), n));   // query n's threshold (both lane halves hold it)
                const unsigned long long r = pool_compact<EMAX>(pool_s + off, pool_i + off, klw, k, lane, tq);
                if (lane == 0) cnt[qb + n] = (int)(r >> 32);
                if ((lane & 31) == n) thr = fmaxf(thr, __builtin_bit_cast(float, (unsigned)r));
            }
        }
    }
}

// dump the 4 registers of quarter Q (rows 8Q .. 8Q+7 of the 32-row tile) of accumulator tile T
#define HB_DUMP_CASE(T, Q)                                                                  \
    case (4 * (T) + (Q)):                                                                   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) sc[r * 64 + lane] = acc[T][4 * (Q) + r]; \
        break;
#define HB_DUMP_TILE(T) HB_DUMP_CASE(T, 0) HB_DUMP_CASE(T, 1) HB_DUMP_CASE(T, 2) HB_DUMP_CASE(T, 3)
// ... or pick them into four scalars-per-lane (pools)
#define HB_PICK_CASE(T, Q)                                                                  \
    case (4 * (T) + (Q)):                                                                   \
        r0 = acc[T][4 * (Q)]; r1 = acc[T][4 * (Q) + 1]; r2 = acc[T][4 * (Q) + 2]; r3 = acc[T][4 * (Q) + 3]; \
        break;
#define HB_PICK_TILE(T) HB_PICK_CASE(T, 0) HB_PICK_CASE(T, 1) HB_PICK_CASE(T, 2) HB_PICK_CASE(T, 3)

// Epilogue of one (query tile, bank tile) pair for one wave: filter the wave's 256 x 32 scores against the
// per-query thresholds (phase 1, always) and hand the rare survivors to the lists / pools (phase 2).
// qb = first of the 32 queries (of the workgroup's 256) that `acc` holds.
// WIDE = false: lst_s / lst_i are the sorted LDS lists.  WIDE = true: they are the slot's pools in global memory
// (row stride klw = capacity) and `cnt` holds the fill counts of the workgroup's 256 queries in LDS.
// CEIL (pools only; a later pass of a search with k > 256, hb_launch_knn_bigk): only rows strictly BEHIND the query's ceiling key (c_s, c_i) in
// the ordering (score descending, row ascending) take part -- tested where a survivor is queued, four compares per flagged quarter.  (The
// rows ahead of the ceiling flag their quarters for nothing: at most k quarters per query over the whole search.  Masking the 128
// accumulators ahead of the epilogue instead cost 56 spilled registers inside the stage loop: a pass 2.8 x as long.)
template <bool SLOW = true, bool WIDE = false, int EMAX = HB_POOL_MAX / 64, bool CEIL = false>
__device__ __forceinline__ int tile_epilogue(f32x16 (&acc)[8], float& thr, float* lst_s, unsigned* lst_i, float* sc,
                                              int qb, int lane, int k, unsigned bt, int klw = HB_KL, int* cnt = nullptr,
                                              float c_s = INFINITY, unsigned c_i = 0u) {
    unsigned qmask = 0;   // bit 4t+q: quarter q (8 bank rows) of row tile t holds a score above its query's threshold
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool any = (acc[t][4 * q] > thr) | (acc[t][4 * q + 1] > thr) | (acc[t][4 * q + 2] > thr) | (acc[t][4 * q + 3] > thr);
            if (__ballot(any) != 0ull) qmask |= 1u << (4 * t + q);
        }
    if (!SLOW) { asm volatile("" :: "s"(qmask)); return 0; }
    const int flagged = __builtin_popcount(qmask);
    if constexpr (WIDE) {
        // pools: a flagged quarter's four registers (8 bank rows x 32 queries) are tested lane by lane, the survivors queued in registers
        // and appended right away (pool_drain) -- so the queue cannot overflow and later quarters already see a threshold that a
        // compaction raised.  (Until round 3 the quarter went through LDS and was walked row by row like the lists below: 256 serial
        // steps for a slot's first tiles, where every quarter is flagged; fp16 candidate kernel, 50,176 x 384: 2.30 -> 2.04 ms.)
        while (qmask) {
            const int bit = __builtin_ctz(qmask);
            qmask &= qmask - 1;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            switch (bit) {
                HB_PICK_TILE(0) HB_PICK_TILE(1) HB_PICK_TILE(2) HB_PICK_TILE(3)
                HB_PICK_TILE(4) HB_PICK_TILE(5) HB_PICK_TILE(6) HB_PICK_TILE(7)
            }
            float q0v = 0.f, q1v = 0.f, q2v = 0.f, q3v = 0.f;
            int np = 0;
            const unsigned row0 = bt * HB_BT + (bit >> 2) * 32 + (bit & 3) * 8 + 4u * (unsigned)(lane >> 5);
            bool p0 = r0 > thr, p1 = r1 > thr, p2 = r2 > thr, p3 = r3 > thr;
            if constexpr (CEIL) {
                p0 = p0 && (r0 < c_s || (r0 == c_s && row0 > c_i));
                p1 = p1 && (r1 < c_s || (r1 == c_s && row0 + 1u > c_i));
                p2 = p2 && (r2 < c_s || (r2 == c_s && row0 + 2u > c_i));
                p3 = p3 && (r3 < c_s || (r3 == c_s && row0 + 3u > c_i));
            }
            if (p0) { q0v = r0; np = 1; }
            if (p1) { q3v = q2v; q2v = q1v; q1v = q0v; q0v = r1; ++np; }
            if (p2) { q3v = q2v; q2v = q1v; q1v = q0v; q0v = r2; ++np; }
            if (p3) { q3v = q2v; q2v = q1v; q1v = q0v; q0v = r3; ++np; }
            // the row codes follow from which registers passed: entry i (newest first) is the i-th highest set bit
            const int pm = (p0 ? 1 : 0) | (p1 ? 2 : 0) | (p2 ? 4 : 0) | (p3 ? 8 : 0);
            int m = pm;
            const int q0c = m ? 31 - __builtin_clz(m) : 0; m &= ~(1 << q0c);
            const int q1c = m ? 31 - __builtin_clz(m) : 0; m &= ~(1 << q1c);
            const int q2c = m ? 31 - __builtin_clz(m) : 0; m &= ~(1 << q2c);
            const int q3c = m ? 31 - __builtin_clz(m) : 0;
            pool_drain<EMAX>(np, q0v, q1v, q2v, q3v, q0c, q1c, q2c, q3c, thr, lst_s, lst_i, qb, lane, k, row0, klw, cnt);
        }
        return flagged;
    }
    while (qmask) {   // wave-uniform slow path; ascending bit order = ascending bank row
        const int bit = __builtin_ctz(qmask);
        qmask &= qmask - 1;
        switch (bit) {
            HB_DUMP_TILE(0) HB_DUMP_TILE(1) HB_DUMP_TILE(2) HB_DUMP_TILE(3)
            HB_DUMP_TILE(4) HB_DUMP_TILE(5) HB_DUMP_TILE(6) HB_DUMP_TILE(7)
        }
        const unsigned row_base = bt * HB_BT + (bit >> 2) * 32 + (bit & 3) * 8;
        // rows inside the quarter: lane half hh holds rows 4*hh + j in register j
        for (int hh = 0; hh < 2; ++hh)
            for (int j = 0; j < 4; ++j) {
                const float v = sc[j * 64 + lane];
                unsigned long long m = __ballot(v > thr);
                m &= hh ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1;
                    const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
                    const int n = l & 31;
                    list_insert(lst_s, lst_i, qb + n, k, s, row_base + hh * 4 + j, lane);
                    const float kth = lst_s[(qb + n) * HB_KL + (k - 1)];
                    if ((lane & 31) == n) thr = fmaxf(thr, kth);
                }
            }
    }
    return flagged;
}

// ---- pool epilogue, second design: scan into a per-lane register queue, then drain ---------------------------------------
// With candidate pools a 32-query x 256-row tile has SEVERAL survivors (k' = 64 and thresholds as stale as the last
// compaction: about four per tile at 10 M rows), so the "rare slow path" of tile_epilogue is the common path there: per
// flagged quarter it dumps registers to LDS and walks them back one LDS round trip at a time (measured on the fp16
// candidate kernel: 72 of 344 ms).  Here every accumulator register is tested once (v_cmp into VCC + a branch that is
// almost never taken), a passing lane pushes (score, row code) onto a four-deep queue in its own registers, and the
// queues are drained afterwards, one lane half at a time (the two halves hold different rows of the SAME query).  A
// lane with more than four survivors in one tile (the first tiles of a slot) sends the wave through tile_epilogue
// instead -- nothing has been appended by then, so nothing is appended twice.
#define HB_SCAN_REG(T, R)                                                                                    \
    {                                                                                                        \
        if (__builtin_expect(__ballot(acc[T][R] > thr) != 0ull, 0)) {                                        \
            float a_ = acc[T][R];                                                                            \
            asm volatile("" : "+v"(a_));   /* compare again in here: the mask of the test need not be kept for this path */ \
            if (a_ > thr) {                                                                                  \
                q3v = q2v; q3c = q2c; q2v = q1v; q2c = q1c; q1v = q0v; q1c = q0c;                            \
                q0v = a_; q0c = (T) * 32 + 8 * ((R) >> 2) + ((R) & 3);                                       \
                ++np;                                                                                        \
            }                                                                                                \
            asm volatile("" : "+v"(np), "+v"(q0v), "+v"(q0c));   /* the push happens HERE */                  \
        }                                                                                                    \
    }
// four registers (8 bank rows x 32 queries) share one test: a VALU compare feeding a scalar branch costs about 20 cycles,
// two max instructions 8
#define HB_SCAN_QUAD(T, Q)                                                                                   \
    {                                                                                                        \
        const float m_ = fmaxf(fmaxf(acc[T][4 * (Q)], acc[T][4 * (Q) + 1]), fmaxf(acc[T][4 * (Q) + 2], acc[T][4 * (Q) + 3])); \
        if (__builtin_expect(__ballot(m_ > thr) != 0ull, 0)) {                                               \
            HB_SCAN_REG(T, 4 * (Q)) HB_SCAN_REG(T, 4 * (Q) + 1) HB_SCAN_REG(T, 4 * (Q) + 2) HB_SCAN_REG(T, 4 * (Q) + 3) \
        }                                                                                                    \
    }
// ... and the 16 registers of a row tile (32 bank rows x 32 queries) share one test in front of that: with phased floors a wave's
// 256 x 32 tile has well under one survivor on average (k' ln(N / n0) candidates per query over the whole search), so nearly every
// row tile is dismissed by 8 max instructions + 1 compare instead of its four quads' 4 x (4 + 1 + 1).  v_max3_f32 from inline asm:
// fmaxf() makes hipcc quiet a possible signalling NaN first (a v_max_f32 x, x per operand); MFMA results are never signalling,
// a quiet NaN loses every maximum (IEEE maxNum) and the final `>` is false for it either way.
__device__ __forceinline__ float hb_max3(float a, float b, float c) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}
#define HB_SCAN_TILE(T)                                                                                      \
    {                                                                                                        \
        const float a_ = hb_max3(acc[T][0], acc[T][1], acc[T][2]), b_ = hb_max3(acc[T][3], acc[T][4], acc[T][5]);      \
        const float c_ = hb_max3(acc[T][6], acc[T][7], acc[T][8]), d_ = hb_max3(acc[T][9], acc[T][10], acc[T][11]);    \
        const float e_ = hb_max3(acc[T][12], acc[T][13], acc[T][14]);                                          \
        const float t_ = hb_max3(hb_max3(a_, b_, c_), hb_max3(d_, e_, acc[T][15]), -INFINITY);                  \
        if (__builtin_expect(__ballot(t_ > thr) != 0ull, 0)) { HB_SCAN_QUAD(T, 0) HB_SCAN_QUAD(T, 1) HB_SCAN_QUAD(T, 2) HB_SCAN_QUAD(T, 3) } \
    }

// `bulk` (wave-uniform, kept by the caller across the tiles of a segment): the tile goes straight to tile_epilogue's quarter loop.
// A slot's first tiles -- and the first tiles of every phase while the floors are loose -- have survivors in every quarter: the
// unrolled scan would take its 128 out-of-line branches only to find a queue overflowing.  The mode ends when a tile flags at most
// HB_BULK_QUADS of its 32 quarters and comes back when a queue overflows.  (The quarter loop for EVERY tile measured 1-5 % slower
// at 50 k - 2 M rows and 0.5 % faster at 10 M x 768: its test is four compares per quarter against the scan's max tree.)
#define HB_BULK_QUADS 12   // (18 / 24 / 30 measure the same from 50 k to 10 M rows, k = 30 and 90: profiles/r04/bulk_threshold_ab.txt)
template <int EMAX>
__device__ __forceinline__ void pool_epilogue_scan(f32x16 (&acc)[8], float& thr, float* pool_s, unsigned* pool_i, float* sc, int qb,
                                                   int lane, int k, unsigned bt, int klw, int* cnt, bool& bulk) {
    if (!bulk) {
        float q0v = 0.f, q1v = 0.f, q2v = 0.f, q3v = 0.f;
        int q0c = 0, q1c = 0, q2c = 0, q3c = 0, np = 0;
        HB_SCAN_TILE(0) HB_SCAN_TILE(1) HB_SCAN_TILE(2) HB_SCAN_TILE(3) HB_SCAN_TILE(4) HB_SCAN_TILE(5) HB_SCAN_TILE(6) HB_SCAN_TILE(7)
        if (__ballot(np != 0) == 0ull) return;
        if (__ballot(np > 4) == 0ull) {
            pool_drain<EMAX>(np, q0v, q1v, q2v, q3v, q0c, q1c, q2c, q3c, thr, pool_s, pool_i, qb, lane, k,
                             bt * HB_BT + 4u * (unsigned)(lane >> 5), klw, cnt);
            return;
        }
        // a queue overflowed: the quarter loop rescans the tile (nothing was appended yet)
    }
    // (the threshold goes through an opaque copy: otherwise the compiler keeps the scan's 128 compare masks alive, in
    // spilled SGPRs, to reuse them in the quarter loop's own compares)
    float t2 = thr;
    asm volatile("" : "+v"(t2));
    const int flagged = tile_epilogue<true, true, EMAX>(acc, t2, pool_s, pool_i, sc, qb, lane, k, bt, klw, cnt);
    thr = t2;
    bulk = flagged > HB_BULK_QUADS;
}

// The same scan for the sorted LDS lists (k <= HB_KL): a flagged quad's survivors (at most four per lane) go into the
// register queue and are inserted right away, one by one, with the wave-cooperative list_insert -- so the queue cannot
// overflow and later quads already see the raised thresholds.  Insertion order inside a tile is free: every queued score
// passed a threshold that is strict against rows of earlier tiles (lower ids), and list_insert compares the full key.
// Measured at 50,176 x 384 (few rows per slot: 170 insertions per query and slot): the dump-and-walk path of
// tile_epilogue spent 45 % of the kernel there.
#define HB_LIST_QUAD(T, Q)                                                                                   \
    {                                                                                                        \
        const float m_ = fmaxf(fmaxf(acc[T][4 * (Q)], acc[T][4 * (Q) + 1]), fmaxf(acc[T][4 * (Q) + 2], acc[T][4 * (Q) + 3])); \
        if (__builtin_expect(__ballot(m_ > thr) != 0ull, 0)) {                                               \
            float q0v = 0.f, q1v = 0.f, q2v = 0.f, q3v = 0.f;                                                \
            int q0c = 0, q1c = 0, q2c = 0, q3c = 0, np = 0;                                                  \
            HB_SCAN_REG(T, 4 * (Q)) HB_SCAN_REG(T, 4 * (Q) + 1) HB_SCAN_REG(T, 4 * (Q) + 2) HB_SCAN_REG(T, 4 * (Q) + 3) \
            list_drain(lst_s, lst_i, qb, lane, k, row0, thr, np, q0v, q1v, q2v, q3v, q0c, q1c, q2c, q3c);    \
        }                                                                                                    \
    }
#define HB_LIST_TILE(T) HB_LIST_QUAD(T, 0) HB_LIST_QUAD(T, 1) HB_LIST_QUAD(T, 2) HB_LIST_QUAD(T, 3)
__device__ __forceinline__ void list_drain(float* lst_s, unsigned* lst_i, int qb, int lane, int k, unsigned row0, float& thr, int np,
                                           float q0v, float q1v, float q2v, float q3v, int q0c, int q1c, int q2c, int q3c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long m = __ballot(np > i);
        if (m == 0ull) break;
        const float v = i == 0 ? q0v : i == 1 ? q1v : i == 2 ? q2v : q3v;
        const int code = i == 0 ? q0c : i == 1 ? q1c : i == 2 ? q2c : q3c;
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
            const unsigned row = row0 + 4u * (unsigned)(l >> 5) + (unsigned)__builtin_amdgcn_readlane(code, l);
            const int n = l & 31;
            list_insert(lst_s, lst_i, qb + n, k, s, row, lane);
            const float kth = lst_s[(qb + n) * HB_KL + (k - 1)];
            if ((lane & 31) == n) thr = fmaxf(thr, kth);
        }
    }
}
__device__ __forceinline__ void list_epilogue_scan(f32x16 (&acc)[8], float& thr, float* lst_s, unsigned* lst_i, int qb, int lane,
                                                   int k, unsigned bt) {
    const unsigned row0 = bt * HB_BT;
    HB_LIST_TILE(0) HB_LIST_TILE(1) HB_LIST_TILE(2) HB_LIST_TILE(3) HB_LIST_TILE(4) HB_LIST_TILE(5) HB_LIST_TILE(6) HB_LIST_TILE(7)
}

// ---- shared threshold floor ----------------------------------------------------------------------------------
// A query tile's bank rows are spread over several slots (workgroups / panels), each with its own running best-k.
// The k-th best score of ANY slot is a lower bound of the final k-th best, so slots publish theirs (atomic max of the
// monotone key) when a segment ends and start their next segment from the best bound published so far: a slot no
// longer has to rediscover a threshold that another one already knows (the cold start of every new slot, and the
// k ln(rows/k) insertions of a slot that only ever sees a slice of the bank).  A foreign bound g admits ties
// (score == g may still win on the id), hence the floor is the float just below g; own k-th scores keep the strict
// rule.  Which scores get filtered early depends on timing, the merged result does not.
__device__ __forceinline__ float floor_from_key(unsigned g) {
    if (g <= 0x007FFFFFu) return -INFINITY;                 // key(-inf): nothing published yet
    g -= 1u;
    if (g == 0x7FFFFFFFu) g = 0x7FFFFFFEu;                  // skip -0.0 (equal to +0.0): largest float below zero
    return __builtin_bit_cast(float, (g & 0x80000000u) ? (g ^ 0x80000000u) : ~g);
}
__device__ __forceinline__ float floor_load(const unsigned* gthr, int q) {
    return floor_from_key(__hip_atomic_load(gthr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void floor_publish(unsigned* gthr, int q, float thr) {
    if (thr > -INFINITY) atomicMax(gthr + q, pool_key(thr));
}

// ---- per-tile floors of small searches (fewer than ~16 k stages per workgroup) ----------------------------------------------
// A slot sees few rows there (37 tiles at 50,176 x 384) and would insert several times as much on its own bound, so the slots of
// a query tile exchange floors every TILE, not only at the end of a segment (hbird_knn_bd.hip: every tile while floors are loose, every
// 16th afterwards -- an exchange costs a tile about 1 %).
//  * Plain floor: the maximum of the slots' k-th bests (atomic max of a monotone key) -- the k-th best of ONE slot's share.
//  * QUOTA floors (query tiles with up to seven slots): the slots see disjoint rows, so if each of the S slots knows
//    c = ceil(k / S) rows that reach v_i, S * c >= k rows reach min v_i: every slot publishes its c-th best (a column of its own,
//    plain agent-scope stores) and filters below the minimum of the S columns -- about the k-th best of ALL rows seen.  A
//    query tile's first slot starts late (its workgroup finishes another query tile first), so every slot also publishes its
//    ceil(k / (S - 1))-th best (columns 8..14): the SECOND smallest of those bounds the union whenever S - 1 slots have published.
//    16 keys per query (column 7: the plain floor), 2 KiB per wave and tile.  50,176 x 384: 105 -> 55-60 candidates per wave and
//    tile.
// The keys come in by LDS-DMA at the tile's start (older than the stage's copies, so the kernels' hand-counted vmcnt holds) and
// are read at its end; a foreign bound admits ties (floor_from_key: the float just below).
#define HB_QUOTA_MAX 7
__device__ __forceinline__ void small_floor_request(const unsigned* qfl, const unsigned* gthr, const hb_seg& seg, int w, int lane,
                                                    unsigned* qf /* 512 words of LDS */, float* sc /* 256 words */) {
    if (seg.nsl <= HB_QUOTA_MAX) {
        const unsigned* src = qfl + (size_t)(seg.q_tile * HB_QT + w * 32) * 16 + lane * 4;
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)qf, 16, 0, 16);
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + 256), (lds_void*)(qf + 256), 16, 0, 16);
    } else if (lane < 32)
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(gthr + seg.q_tile * HB_QT + w * 32 + lane), (lds_void*)sc, 4, 0, 16);
}
__device__ __forceinline__ float small_floor_read(const hb_seg& seg, const unsigned* qf, const float* sc, int lane) {
    if (seg.nsl > HB_QUOTA_MAX) return floor_from_key(reinterpret_cast<const unsigned*>(sc)[lane & 31]);
    const unsigned* kk = qf + (lane & 31) * 16;   // my query's columns
    unsigned m1 = 0xFFFFFFFFu, lo = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;   // min of level 1; smallest, second smallest of level 2
#pragma unroll
    for (int i = 0; i < HB_QUOTA_MAX; ++i) {
        const unsigned v1 = kk[i], v2 = kk[8 + i];
        if (i < seg.nsl) {
            m1 = min(m1, v1);
            m2 = min(m2, max(lo, v2));
            lo = min(lo, v2);
        }
    }
    if (seg.nsl < 2) m2 = 0u;   // a single slot: no "all but one"
    return floor_from_key(max(max(m1, m2), kk[7]));
}
__device__ __forceinline__ void small_floor_publish(unsigned* qfl, unsigned* gthr, const hb_seg& seg, const float* lst_s, int myq, int k,
                                                    float thr, int lane) {
    if (lane >= 32) return;
    if (seg.nsl <= HB_QUOTA_MAX) {
        unsigned* col = qfl + (size_t)(seg.q_tile * HB_QT + myq) * 16;
        const int c1 = (k + seg.nsl - 1) / seg.nsl, c2 = seg.nsl > 1 ? (k + seg.nsl - 2) / (seg.nsl - 1) : k;
        const float v1 = lst_s[myq * HB_KL + (c1 - 1)], v2 = lst_s[myq * HB_KL + (c2 - 1)];
        if (v1 > -INFINITY) __hip_atomic_store(col + seg.ord, pool_key(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v2 > -INFINITY) __hip_atomic_store(col + 8 + seg.ord, pool_key(v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        floor_publish(col, 7, thr);
    } else floor_publish(gthr, seg.q_tile * HB_QT + myq, thr);
}

// pools: fill counts / thresholds of the wave's 32 queries at the start and the end of a segment
__device__ __forceinline__ float pool_begin(const knn_args_pool_view& pv, int slot, bool first, int* cnt, int myq, int lane) {
    float thr = -INFINITY;
    int c = 0;
    if (!first) {
        c = __hip_atomic_load(pv.cnt + (size_t)slot * HB_QT + myq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        thr = __hip_atomic_load(pv.thr + (size_t)slot * HB_QT + myq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane < 32) cnt[myq] = c;
    return thr;
}
__device__ __forceinline__ void pool_end(const knn_args_pool_view& pv, int slot, const int* cnt, float thr, int myq, int lane) {
    if (lane < 32) { pv.cnt[(size_t)slot * HB_QT + myq] = cnt[myq]; pv.thr[(size_t)slot * HB_QT + myq] = thr; }
}

// Cold start of a slot: with no threshold yet, the first bank tile would insert all of its 256 rows into every query's list (or
// pool) one by one.  Instead a score that at least k of the tile's 256 scores of each query EXCEED -- they sit in two lanes
// (lane, lane ^ 32) x 128 accumulator registers -- is found by bisection between the smallest and the largest of them, and
// everything at or below it is filtered like any other score: about k insertions per query remain.  Any such score is a valid
// threshold; twelve halvings leave 256 / 4096 rows too many on average.  (Until round 3 this was an exact 32-round radix select
// on monotone keys: 28 k vector instructions per wave, 108 us per cold start measured in the fp16 candidate kernel -- nine bank
// tiles' worth -- against 4 k here.)  Returns -inf when the tile has fewer than k real rows (padding rows score -inf).
// NaN scores (a bank row of NaNs: a zero token through the reference's eps-free normalisation, hbird_eval.py:324; a query
// with inf / NaN components) never enter a list -- every filter is a strict `score > threshold`, false for NaN, as in a
// comparison-based k-select.  The tile's NaNs are replaced by -inf first (in place: for the filters after this call -inf and NaN
// are the same thing), so that they count as "no row" here as well.
// ... and only where a query of the wave has no floor yet: a slot that starts in a later phase (or after another slot of its query tile
// has published its k-th best) begins from the floor of ALL rows seen so far, which a single tile's k-th best cannot beat -- the 17 us
// of the bisection were pure skew between the workgroups of that phase.
__device__ __forceinline__ bool cold_start_needed(float thr) { return __ballot(thr == -INFINITY) != 0ull; }
__device__ __forceinline__ float cold_start_threshold(f32x16 (&acc)[8], int k) {
    float hi = -INFINITY, mn = INFINITY;
    int real = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (acc[t][r] == acc[t][r]) ? acc[t][r] : -INFINITY;
            acc[t][r] = v;
            hi = fmaxf(hi, v);
            mn = fminf(mn, v > -INFINITY ? v : INFINITY);
            real += v > -INFINITY ? 1 : 0;
        }
    hi = fmaxf(hi, __shfl_xor(hi, 32));
    mn = fminf(mn, __shfl_xor(mn, 32));
    real += __shfl_xor(real, 32);
    hi = fminf(hi, 3.4028234664e38f);   // +inf scores are ordinary scores; the bisection needs finite ends
    // lo: a float below the smallest real score, so that every real row exceeds it (by a NORMAL amount: nothing here may depend on
    // how compares and adds treat denormals -- a zero query scores 0 against every row)
    float lo = mn - fmaxf(fabsf(mn) * 1e-6f, 1.2e-38f);
    if (real < k) { lo = -INFINITY; hi = -INFINITY; }   // fewer than k real rows: the answer stays -inf
    // An astronomically large score (a +inf or 1e30-sized outlier: an inf component in the query, unnormalised rows of huge norm under
    // IP) stretches the interval so far that twelve LINEAR halvings never reach the bulk of the scores: lo would stay just below the
    // minimum and the whole tile would pass (correct, but the slow path this function exists to avoid).  Such a query halves the
    // interval of the monotone integer KEYS instead (the number of representable values between the ends, whatever their
    // distribution; coarser for ordinary data, which keeps the linear midpoint).  Same loop, another midpoint per lane.
    const bool stretched = !(hi - mn <= 1e30f);
#pragma unroll 1
    for (int it = 0; it < 12; ++it) {
        float mid = 0.5f * lo + 0.5f * hi;
        if (stretched) {
            const unsigned klo = pool_key(lo), khi = pool_key(hi);
            unsigned km = klo + ((khi - klo) >> 1);
            if (km == 0x7FFFFFFFu) km = 0x7FFFFFFEu;      // the key of -0.0, which compares equal to the +0.0 above it
            mid = __builtin_bit_cast(float, (km & 0x80000000u) ? (km ^ 0x80000000u) : ~km);
        }
        int c = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) c += acc[t][r] > mid ? 1 : 0;
        c += __shfl_xor(c, 32);
        if (c >= k) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- floors between two phases of a pool search ---------------------------------------------------------------------------
// (Round 5 also ran all phases in ONE launch -- resident workgroups, a grid barrier and the floor computation inside the kNN kernels; it
// measured 2-10 % slower than a launch per phase, whose cost is arrival skew and the floor computation, not the launches, and was removed in
// round 6: profiles/r05/one_launch_*.txt, profiles/LABBOOK.md, git history.)
typedef __attribute__((address_space(1))) unsigned hb_gu32;
__device__ __forceinline__ unsigned gb_load(const unsigned* p) {
    return __hip_atomic_load((const hb_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The floor of query q from its slots' pools: the scores that can matter (at or above the best threshold a full pool already has) are
// gathered into `cs` (LDS, per_wave floats of this wave) and a score that at least kk of them EXCEED is found by 14 halvings between the
// smallest and the largest (any such score is a valid floor: cold_start_threshold).  One wave; called by pool_floor_kernel (a launch
// between two phases).  Loads bypass the L1 (sc1).  All of a slot's entries are requested before the first is looked at, and the
// next slot's while one is compacted into LDS.
__device__ __forceinline__ float pf_load(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load((const hb_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __attribute__((noinline, unused)) void pool_floor_query(const float* __restrict__ state_s, const int* __restrict__ cnts,
                                                            const float* __restrict__ pthr, const int* __restrict__ qt_off,
                                                            const int* __restrict__ qt_slots, int64_t q, int kk, int klw, float* cs,
                                                            unsigned* __restrict__ gthr, int lane) {
    const int qt = (int)(q / HB_QT), ql = (int)(q % HB_QT);
    const int s0 = qt_off[qt], ns = qt_off[qt + 1] - s0;
    // lane j: slot j's fill count and threshold (slots beyond 64: a second round)
    float tstar = -INFINITY;
    int my_cnt = 0, my_slot = 0;
    for (int j = lane; j < ns; j += 64) {
        const int slot = qt_slots[s0 + j];
        const size_t oq = (size_t)slot * HB_QT + ql;
        const int c = (int)gb_load(reinterpret_cast<const unsigned*>(cnts) + oq);
        if (c >= kk) tstar = fmaxf(tstar, pf_load(pthr + oq));
        if (j < 64) { my_cnt = c; my_slot = slot; }
    }
    for (int o = 32; o > 0; o >>= 1) tstar = fmaxf(tstar, __shfl_xor(tstar, o));
    int n = 0;
    float hi = -INFINITY, mn = INFINITY;
    constexpr int EM = HB_POOL_MAX / 64;
    float va[EM], vb[EM];          // two slots in flight (static register sets: the slot loop is unrolled by two)
    int na = 0, nb = 0;
    auto request = [&](int j, float (&v)[EM], int& nv) {
        int c, slot;
        if (j < 64) { c = __builtin_amdgcn_readlane(my_cnt, j); slot = __builtin_amdgcn_readlane(my_slot, j); }
        else { slot = qt_slots[s0 + j]; c = (int)gb_load(reinterpret_cast<const unsigned*>(cnts) + (size_t)slot * HB_QT + ql); }
        const float* src = state_s + ((size_t)slot * HB_QT + ql) * klw;
        nv = min(klw, c);
#pragma unroll
        for (int u = 0; u < EM; ++u) {
            const int e = u * 64 + lane;
            v[u] = -INFINITY;
            if (u * 64 < nv) { if (e < nv) v[u] = pf_load(src + e); }
        }
    };
    auto consume = [&](const float (&v)[EM], int nv) {
#pragma unroll
        for (int u = 0; u < EM; ++u) {
            if (u * 64 >= nv) break;
            const int e = u * 64 + lane;
            const bool keep = e < nv && (v[u] >= tstar || tstar == -INFINITY);
            const unsigned long long m = __ballot(keep);
            if (keep) { cs[n + __popcll(m & ((1ull << lane) - 1ull))] = v[u]; hi = fmaxf(hi, v[u]); mn = fminf(mn, v[u]); }
            n += __popcll(m);
        }
    };
    if (ns > 0) request(0, va, na);
    for (int j = 0; j < ns; j += 2) {
        if (j + 1 < ns) request(j + 1, vb, nb);
        consume(va, na);
        if (j + 2 < ns) request(j + 2, va, na);
        if (j + 1 < ns) consume(vb, nb);
    }
    if (n < kk) return;   // fewer than kk rows seen so far: no floor yet
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to its lanes
    __builtin_amdgcn_wave_barrier();
    for (int o = 32; o > 0; o >>= 1) { hi = fmaxf(hi, __shfl_xor(hi, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    hi = fminf(hi, 3.4028234664e38f);
    float lo = mn - fmaxf(fabsf(mn) * 1e-6f, 1.2e-38f);   // all n exceed it (cold_start_threshold)
    // the first 256 gathered scores in registers (usually all of them): a round is then compares and ballots only
    const float r0 = lane < n ? cs[lane] : -INFINITY, r1 = 64 + lane < n ? cs[64 + lane] : -INFINITY;
    const float r2 = 128 + lane < n ? cs[128 + lane] : -INFINITY, r3 = 192 + lane < n ? cs[192 + lane] : -INFINITY;
    const bool stretched = !(hi - mn <= 1e30f);   // an astronomically large score: halve the interval of the monotone keys (cold_start_threshold)
    for (int it = 0; it < 14; ++it) {
        float mid = 0.5f * lo + 0.5f * hi;
        if (stretched) {
            const unsigned klo = pool_key(lo), khi = pool_key(hi);
            unsigned km = klo + ((khi - klo) >> 1);
            if (km == 0x7FFFFFFFu) km = 0x7FFFFFFEu;
            mid = __builtin_bit_cast(float, (km & 0x80000000u) ? (km ^ 0x80000000u) : ~km);
        }
        int c = __popcll(__ballot(r0 > mid)) + __popcll(__ballot(r1 > mid)) + __popcll(__ballot(r2 > mid)) + __popcll(__ballot(r3 > mid));
        for (int base = 256; base < n; base += 64) c += __popcll(__ballot(base + lane < n && cs[base + lane] > mid));
        if (c >= kk) lo = mid; else hi = mid;
    }
    __builtin_amdgcn_wave_barrier();      // (the next query of this wave reuses `cs`)
    if (lane == 0 && lo > -INFINITY) atomicMax(gthr + q, pool_key(lo));   // kk rows exceed lo
}

// LDS map shared by the variants (bytes): a 4-slot ring of k8 stages, two row-init buffers, the lists, the scratch
#define KN_SLOT_BYTES 16384                 // 8 KiB bank fragments + 8 KiB query fragments (32 rows x 8 k blocks)
#define KN_RING 4
#define KN_BINIT (KN_RING * KN_SLOT_BYTES)  // 2 x 1 KiB
#define KN_LISTS (KN_BINIT + 2048)
#define KN_SCRATCH (KN_LISTS + 2 * HB_QT * HB_KL * 4)
#define KN_CLWORDS (KN_SCRATCH + 8192)        // scratch: 1 KiB per wave (8 waves) / 2 KiB per wave (4 waves)
#define KN_LDS_TOTAL (KN_CLWORDS + 64)        // landing zone of the cluster progress poll
#define KN_QF KN_LDS_TOTAL                    // small-search instantiation only: quota floors, 2 KiB per wave
#define KN_LDS_TOTAL_COLD (KN_QF + 8 * 2048)
#define KN_FENCE __builtin_amdgcn_sched_barrier(0);

typedef void (*hb_knn_fn)(knn_args);
hb_knn_fn hb_knn_bd_kernel(bool wide, bool clustered, bool small);   // query fragments straight into registers, hbird_knn_bd.hip
int hb_knn_bd_lds_bytes(bool small);
