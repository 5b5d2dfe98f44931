// Third design of the fp16 candidate kernel (use_fp16): the stage loop of knn_f16v2_kernel (hbird_knn_f16.hip:
// bank fragments through an 8-slot LDS ring by LDS-DMA, query fragments straight from global memory into registers, hand-counted
// vmcnt, raw s_barrier) on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.
//
// Why: under the power cap the small shape holds a higher clock at the same FLOP per cycle -- 1925 against 1630 TFLOP/s on a bare
// pipe with random operands (tools/ubench/mfma_f16_shapes.hip), and a timing-only build of the second design's stage loop with its
// MFMAs swapped ran 4.8 % faster (profiles/r03/f16_mfma_shape_ab.md).
//
// STATUS: correct (every fp16 parity test passes on it: same candidates, same final bits) and NOT faster -- 320-325 ms against
// 316-324 ms for the second design at 10 M x 768 on the same boxes (the timing-only build fed both MFMAs of a pair the same
// operands; with real operands the loop is bound by its vector-memory instructions, and a lane holding two queries makes the
// epilogue dearer: four quarter passes per queue entry instead of two halves).  Selectable for A/B (hb_index_set_variant(ix, 5));
// the second design stays the default.
//
// What changes with the shape:
//  * fragment blocks: a 1 KiB block is 16 rows x 32 k (lane l = 16 kb + i: row i, k = 8 kb .. 8 kb + 7) instead of 32 rows x 16 k.
//    The fp16 copies keep their addressing -- [32-row tile][k32 stage][2][1 KiB] -- the two blocks of a (tile, stage) are now the two
//    16-row halves instead of the two k16 groups (tiles_to_f16s_kernel), so copies, ring and query-fragment loads are those of the
//    second design.
//  * accumulators: f32x4 acc[32], tile 2 rt + c = bank rows 16 rt .. 16 rt + 15 x queries 16 c .. 16 c + 15 of the wave's 32; lane
//    l = 16 g + j holds rows 16 rt + 4 g + i (i = 0..3) of query 16 c + j.  A lane thus holds TWO queries (j and 16 + j, two
//    thresholds) and four lanes share one -- cold start, scan, drain and overflow path below are written for that layout.
//  * a stage is 16 fragment pairs: MFMA(2 p, fa, b0), MFMA(2 p + 1, fa, b1), then the fragment register is reloaded for its next use
//    eight pairs later.  Both query fragments of a stage are live during the whole stage, so the pair of the stage three ahead is
//    requested during the first half of the stage (into the buffers of the stage just finished); the bank pieces of the stage four
//    ahead during the second half, as before.  Per wave and stage: B0 B1 | C0 C1 -- "all but the newest 8" at the barrier.
#include "hbird_knn_dev.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define F3_RING 8
#define F3_SLOT 16384                                  // bank fragments of one k32 stage: [16-row tile 0..15][1 KiB]
#define F3_BINIT (F3_RING * F3_SLOT)
#define F3_SCRATCH (F3_BINIT + 2048)
#define F3_PCNT (F3_SCRATCH + 8192)
#define F3_CLWORDS (F3_PCNT + 1024)                    // landing zone of the cluster progress poll
#define F3_LDS_TOTAL (F3_CLWORDS + 64)
static_assert(F3_LDS_TOTAL <= 160 * 1024, "LDS budget");

// fp32 fragment tiles (block(rt32, g8): element (i, kk) at ((kk & 1) * 32 + i) * 4 + (kk >> 1)) -> fp16 blocks of 16 rows x 32 k:
// block ((rt32 * g32 + g) * 2 + sub) holds rows 32 rt32 + 16 sub + 0..15, k = 32 g + 0..31; lane l = 16 kb + i reads the 8 halves
// k = 32 g + 8 kb + 0..7 of row i at byte 16 l.  One thread per 8 output halves.  *overflow as in tiles_to_f16_kernel.
__global__ __launch_bounds__(256) void tiles_to_f16s_kernel(const float* __restrict__ t32, int g8, _Float16* __restrict__ t16, int g32,
                                                            int64_t n_row_tiles, int64_t rt0, int* __restrict__ overflow) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one per (row tile, g32, sub, lane)
    const int64_t total = n_row_tiles * g32 * 128;
    if (gid >= total) return;
    const int l = (int)(gid & 63), sub = (int)((gid >> 6) & 1);
    const int64_t blk = gid >> 7;
    const int g = (int)(blk % g32);
    const int64_t rt = rt0 + blk / g32;
    const int i = 16 * sub + (l & 15), gg = 4 * g + (l >> 4);          // row within the 32-row tile, source k8 group
    f16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.0f;
        if (gg < g8) v = t32[((rt * g8) + gg) * HB_BLK + ((j & 1) * 32 + i) * 4 + (j >> 1)];
        out[j] = (_Float16)v;
        if (overflow && fabsf(v) > 65504.0f && fabsf(v) < INFINITY) *overflow = 1;
    }
    reinterpret_cast<f16x8*>(t16)[((rt * g32 + g) * 2 + sub) * 64 + l] = out;
}

int hb_launch_tiles_to_f16s(const float* t32, int g8, _Float16* t16, int g32, int64_t n_row_tiles, int64_t rt0, int* overflow, hipStream_t s) {
    const int64_t total = n_row_tiles * g32 * 128;
    if (total == 0) return 0;
    tiles_to_f16s_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(t32, g8, t16, g32, n_row_tiles, rt0, overflow);
    HB_HIP(hipGetLastError());
    return 0;
}

// ---- epilogue for the 16x16 accumulator layout ------------------------------------------------------------------------------------
// k-th largest of a query's 256 scores of the tile (four lanes x 64 registers), for both queries of the lane at once: the radix cold
// start of hbird_knn_dev.h (NaNs -> -inf first; returns the floats just below, or -inf with fewer than k real rows).
template <int C>   // query 16 C + (lane & 15) of the wave: accumulator tiles 2 rt + C
__device__ __forceinline__ float cold_start16_one(const f32x4 (&acc)[32], int k) {
    unsigned prefix = 0;
    int kk = k;
    for (int b = 31; b >= 0; --b) {
        const unsigned himask = ~((1u << b) - 1u), cand = prefix | (1u << b);
        int c = 0;
#pragma unroll
        for (int rt = 0; rt < 16; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[2 * rt + C][i];
                asm volatile("" : "+v"(v));   // keeps the 64 keys from being hoisted out of the bit loop (registers)
                c += ((pool_key(v) & himask) == cand) ? 1 : 0;
                asm volatile("" : "+v"(c));   // ... and one compare mask alive at a time (volatile asms keep their order)
            }
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 32);
        if (c >= kk) prefix = cand; else kk -= c;
    }
    return floor_from_key(prefix);   // one key below the k-th largest: its ties still pass; -inf with fewer than k real rows
}
__device__ __forceinline__ void cold_start16(f32x4 (&acc)[32], int k, float& t0, float& t1) {
#pragma unroll
    for (int t = 0; t < 32; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = (acc[t][i] == acc[t][i]) ? acc[t][i] : -INFINITY;
    t0 = fmaxf(t0, cold_start16_one<0>(acc, k));   // one query after the other: half the live compare masks
    t1 = fmaxf(t1, cold_start16_one<1>(acc, k));
}

// Append the lanes' candidates (v, row) to the pools of their queries qq (0..31 of the wave), one 16-lane quarter at a time: the lanes
// of a quarter hold distinct queries, the four quarters rows of the SAME queries (they would race on the fill counts).  A pool that
// fills up is compacted by the whole wave and raises its query's threshold (lanes j = n & 15, t0 / t1 by n >> 4).
template <int EMAX>
__device__ __forceinline__ void pool_append16(bool has, float v, unsigned row, int qq, float& t0, float& t1, float* pool_s, unsigned* pool_i,
                                              int qb, int lane, int k, int klw, int* cnt) {
#pragma nounroll   // one copy of the compaction below (it is inlined, and big)
    for (int g = 0; g < 4; ++g) {
        const bool pass = has && (lane >> 4) == g;
        if (__ballot(pass) == 0ull) continue;
        int c = 0;
        if (pass) {
            c = cnt[qb + qq];
            pool_s[(size_t)(qb + qq) * klw + c] = v;
            pool_i[(size_t)(qb + qq) * klw + c] = row;
            cnt[qb + qq] = c + 1;
        }
        unsigned long long full = __ballot(pass && c + 1 == klw);
        while (full) {
            const int l = __builtin_ctzll(full);
            full &= full - 1;
            const int n = __builtin_amdgcn_readlane(qq, l);
            const size_t off = (size_t)(qb + n) * klw;
            const float kth = pool_compact<EMAX>(pool_s + off, pool_i + off, klw, k, lane);
            if (lane == 0) cnt[qb + n] = k;
            if ((lane & 15) == (n & 15)) { if (n < 16) t0 = fmaxf(t0, kth); else t1 = fmaxf(t1, kth); }
        }
    }
}

// one accumulator tile = 16 bank rows x 16 queries (a lane: 4 rows of one query): one test per tile, its survivors pushed one by one
#define F3_SCAN_REG(T, I)                                                                                    \
    {                                                                                                        \
        float a_ = acc[T][I];                                                                                \
        asm volatile("" : "+v"(a_));   /* compare again in here: the mask of the test need not be kept */    \
        if (a_ > (((T) & 1) ? t1 : t0)) {                                                                    \
            q3v = q2v; q3c = q2c; q2v = q1v; q2c = q1c; q1v = q0v; q1c = q0c;                                \
            q0v = a_; q0c = ((T) >> 1) * 16 + (I) + (((T) & 1) << 8);                                        \
            ++np;                                                                                            \
        }                                                                                                    \
        asm volatile("" : "+v"(np), "+v"(q0v), "+v"(q0c));   /* the push happens HERE */                      \
    }
#define F3_SCAN_TILE(T)                                                                                      \
    {                                                                                                        \
        const float m_ = fmaxf(fmaxf(acc[T][0], acc[T][1]), fmaxf(acc[T][2], acc[T][3]));                    \
        if (__builtin_expect(__ballot(m_ > (((T) & 1) ? t1 : t0)) != 0ull, 0)) {                             \
            F3_SCAN_REG(T, 0) F3_SCAN_REG(T, 1) F3_SCAN_REG(T, 2) F3_SCAN_REG(T, 3)                          \
        }                                                                                                    \
    }
#define F3_SCAN_4(T) F3_SCAN_TILE(T) F3_SCAN_TILE(T + 1) F3_SCAN_TILE(T + 2) F3_SCAN_TILE(T + 3)
#define F3_DUMP_CASE(T) case (T): _Pragma("unroll") for (int i = 0; i < 4; ++i) sc[i * 64 + lane] = acc[T][i]; break;
#define F3_DUMP_4(T) F3_DUMP_CASE(T) F3_DUMP_CASE(T + 1) F3_DUMP_CASE(T + 2) F3_DUMP_CASE(T + 3)

// Scan the 128 accumulators against the lane's two thresholds into a four-deep register queue, then drain it (pool_epilogue_scan of
// hbird_knn_dev.h for this layout).  A lane with more than four survivors (the first tiles of a slot) sends the wave through the
// general path: every tile with a survivor is dumped to the wave's LDS scratch and walked -- nothing was appended before, so nothing
// is appended twice.
template <int EMAX>
__device__ __forceinline__ void pool_epilogue16(f32x4 (&acc)[32], float& t0, float& t1, float* pool_s, unsigned* pool_i, float* sc, int qb,
                                                int lane, int k, unsigned bt, int klw, int* cnt) {
    float q0v = 0.f, q1v = 0.f, q2v = 0.f, q3v = 0.f;
    int q0c = 0, q1c = 0, q2c = 0, q3c = 0, np = 0;
    F3_SCAN_4(0) F3_SCAN_4(4) F3_SCAN_4(8) F3_SCAN_4(12) F3_SCAN_4(16) F3_SCAN_4(20) F3_SCAN_4(24) F3_SCAN_4(28)
    if (__ballot(np != 0) == 0ull) return;
    const unsigned row0 = bt * HB_BT + 4u * (unsigned)(lane >> 4);
    // ONE loop and one copy of the append code for both cases: the (up to four) queued entries, or -- after an overflow -- all 128
    // accumulators again, a tile at a time through the wave's LDS scratch, tested against the thresholds as they rise
    const bool ovf = __ballot(np > 4) != 0ull;
    const int items = ovf ? 128 : 4;
#pragma nounroll
    for (int it = 0; it < items; ++it) {
        bool has;
        float v;
        int code;
        if (ovf) {
            const int T = it >> 2, i = it & 3;
            if (i == 0) {
                switch (T) {
                    F3_DUMP_4(0) F3_DUMP_4(4) F3_DUMP_4(8) F3_DUMP_4(12) F3_DUMP_4(16) F3_DUMP_4(20) F3_DUMP_4(24) F3_DUMP_4(28)
                }
            }
            v = sc[i * 64 + lane];
            has = v > ((T & 1) ? t1 : t0);
            code = (T >> 1) * 16 + i + ((T & 1) << 8);
        } else {
            has = np > it;   // the queue is rotated: entry `it`, newest first (the order is free)
            v = q0v; code = q0c;
            q0v = q1v; q0c = q1c; q1v = q2v; q1c = q2c; q2v = q3v; q2c = q3c;
        }
        if (__ballot(has) == 0ull) { if (ovf) continue; else break; }
        pool_append16<EMAX>(has, v, row0 + (unsigned)(code & 255), (code >> 8) * 16 + (lane & 15), t0, t1, pool_s, pool_i, qb, lane, k, klw, cnt);
    }
}

template <int EMAX>   // pool capacity / 64 that the instantiation can compact
__global__ __launch_bounds__(HB_THREADS, 2) void knn_f16s_kernel(knn16_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sc = reinterpret_cast<float*>(smem + F3_SCRATCH) + w * 256;
    int* pcnt = reinterpret_cast<int*>(smem + F3_PCNT);
    const int g16 = a.g16, k = a.k, klw = a.klw;
    const int NS = g16 / 2;   // k32 stages per bank tile, a multiple of 4 (dp16 is a multiple of 128)
    const int q0 = w * 32 + (lane & 15), q1 = q0 + 16;   // my two queries in the tile
    const unsigned lane_off = (unsigned)lane * 16u;
    cl_sync cs = cl_init(a.wg_member, a.prog, a.cl, a.lag, blockIdx.x, w == 0, smem + F3_CLWORDS);

    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_end[blockIdx.x];   // this launch's share of the block's segments (phases: hb_launch_knn)
    if (w == 0) cl_publish(cs, seg_begin < seg_end ? a.segs[seg_begin].tile0 * NS : 0x7FFFFFFF, lane);
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = a.segs[si];
        const int bstride = seg.stride;
        float* wl_s = a.state_s + (size_t)seg.slot * HB_QT * klw;
        unsigned* wl_i = a.state_i + (size_t)seg.slot * HB_QT * klw;
        // fill counts (lanes 0-31: query 32 w + lane) and thresholds (every lane: its two queries) of the slot
        float t0 = -INFINITY, t1 = -INFINITY;
        {
            int c = 0;
            if (!seg.first) {
                c = __hip_atomic_load(a.state_cnt + (size_t)seg.slot * HB_QT + w * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t0 = __hip_atomic_load(a.state_thr + (size_t)seg.slot * HB_QT + q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t1 = __hip_atomic_load(a.state_thr + (size_t)seg.slot * HB_QT + q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane < 32) pcnt[w * 32 + lane] = c;
        }
        t0 = fmaxf(t0, floor_load(a.gthr, seg.q_tile * HB_QT + q0));
        t1 = fmaxf(t1, floor_load(a.gthr, seg.q_tile * HB_QT + q1));
        const int total = seg.n_tiles * NS, clock0 = seg.tile0 * NS;
        f32x4 acc[32];
        f16x8 fa[8];        // bank fragments: a ring of eight, fragment r lives in fa[r & 7] and is loaded eight pairs (16 MFMAs) ahead
        f16x8 bq[4][2];     // query fragments of four stages, queries 0-15 / 16-31 of the wave

        const char* bank_w = reinterpret_cast<const char*>(a.bank16) + (size_t)w * g16 * 1024;
        const char* query_w = reinterpret_cast<const char*>(a.q16) + (size_t)(seg.q_tile * 8 + w) * g16 * 1024;
        int fbt = seg.b_tile0, fks = 0, slot_f = 0, left = total, fpar = 0;
        const char* qsrc_b = query_w;   // query fragments of the newest stage whose bank pieces have been requested
#define F3_QSRC() (query_w + (size_t)fks * 2048)
#define F3_BL(REG, SRC, OFF, CLS)                                                                                            \
        asm volatile("s_cmp_lt_u32 %3, 4\n\ts_cbranch_scc" #CLS " .Lf3s_%=\n\tglobal_load_dwordx4 %0, %1, %2 offset:" #OFF "\n.Lf3s_%=:" \
                     : "+v"(REG) : "v"(lane_off), "s"(SRC), "s"(w) : "memory", "scc");
#define F3_BL_ALL(REG, SRC, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(REG) : "v"(lane_off), "s"(SRC) : "memory");
#define F3_COPY(I)   /* bank piece I of this wave: rows 32 w + 16 I .. + 15 of the tile */                                    \
        {                                                                                                                    \
            const char* bsrc = bank_w + ((size_t)fbt * 8 * g16 + (size_t)fks * 2 + (I)) * 1024 + lane_off;                   \
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)bsrc, (lds_void*)(smem + slot_f * F3_SLOT + (w * 2 + (I)) * 1024), 16, 0, 0); \
        }
#define F3_ADVANCE()                                                                                                         \
        {                                                                                                                    \
            if (fks == 0 && w == 0) {   /* (the lane offset is laundered: hoisted out of the loop, the per-lane address is spilled) */ \
                int l2 = lane;                                                                                               \
                asm volatile("" : "+v"(l2));                                                                                 \
                glds16(a.binit + (size_t)fbt * HB_BT + l2 * 4, smem + F3_BINIT + fpar * 1024);                               \
            }                                                                                                                \
            qsrc_b = F3_QSRC();                                                                                              \
            if (--left > 0) { if (++fks == NS) { fks = 0; fbt += bstride; fpar ^= 1; } }                                     \
            slot_f = (slot_f + 1) & (F3_RING - 1);                                                                           \
        }
        // vmcnt by hand.  Per stage s a wave requests B0 B1 of stage s+3 before the stage's barrier and C0 C1 of stage s+4 after it
        // (wave 0 now and then one more request, which only makes a wait stricter).  At the barrier of stage s everything of stage
        // s+1 must have landed; its youngest request is B1(s+1) (issued in stage s-2), behind it come C0 C1 (s+2), B0 B1 (s+2),
        // C0 C1 (s+3), B0 B1 (s+3) -> "all but the newest 8".
        // Prologue: C(0) B(0) C(1) B(1) C(2) B(2) C(3); stage 0 needs C(0) B(0): ten younger requests.
        F3_COPY(0) F3_COPY(1) F3_BL_ALL(bq[0][0], F3_QSRC(), 0) F3_BL_ALL(bq[0][1], F3_QSRC(), 1024) F3_ADVANCE()
        F3_COPY(0) F3_COPY(1) F3_BL_ALL(bq[1][0], F3_QSRC(), 0) F3_BL_ALL(bq[1][1], F3_QSRC(), 1024) F3_ADVANCE()
        F3_COPY(0) F3_COPY(1) F3_BL_ALL(bq[2][0], F3_QSRC(), 0) F3_BL_ALL(bq[2][1], F3_QSRC(), 1024) F3_ADVANCE()
        F3_COPY(0) F3_COPY(1) F3_ADVANCE()
        asm volatile("s_waitcnt vmcnt(10)" : "+v"(bq[0][0]), "+v"(bq[0][1]) :: "memory");
        __syncthreads();
        int slot_c = 0, ks = 0, bt = seg.b_tile0, cpar = 0;
#define F3_INIT_TILE()                                                                                                       \
        {                                                                                                                    \
            const f32x4* bi = reinterpret_cast<const f32x4*>(smem + F3_BINIT + cpar * 1024) + (lane >> 4);                   \
            _Pragma("unroll") for (int rt = 0; rt < 16; ++rt) { acc[2 * rt] = bi[4 * rt]; acc[2 * rt + 1] = acc[2 * rt]; }   \
        }
#define F3_FIRST_FRAGMENTS()                                                                                                 \
        {                                                                                                                    \
            const f16x8* A = reinterpret_cast<const f16x8*>(smem + slot_c * F3_SLOT) + lane;                                 \
            _Pragma("unroll") for (int t = 0; t < 8; ++t) fa[t] = A[t * 64];                                                 \
        }
        F3_FIRST_FRAGMENTS()
        F3_INIT_TILE()
// one fragment at a time against both query fragments (two fragments x two query fragments with consecutive MFMAs sharing an
// operand -- f0 b0, f1 b0, f0 b1, f1 b1 -- measured 1.3 % slower)
#define F3_MM2(P, U, RD0, RD1, REQ0, REQ1)                                                                                   \
        KN_FENCE acc[2 * (P)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(P) & 7], bq[U][0], acc[2 * (P)], 0, 0, 0);        \
        acc[2 * (P) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(P) & 7], bq[U][1], acc[2 * (P) + 1], 0, 0, 0);         \
        KN_FENCE fa[(P) & 7] = RD0; REQ0                                                                                     \
        KN_FENCE acc[2 * (P) + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[((P) + 1) & 7], bq[U][0], acc[2 * (P) + 2], 0, 0, 0); \
        acc[2 * (P) + 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[((P) + 1) & 7], bq[U][1], acc[2 * (P) + 3], 0, 0, 0);   \
        KN_FENCE fa[((P) + 1) & 7] = RD1; REQ1
#define F3_NONE
#define F3_STAGE(U)                                                                                                          \
        {                                                                                                                    \
            const f16x8* Ac = reinterpret_cast<const f16x8*>(smem + slot_c * F3_SLOT) + lane;                                \
            const int slot_n = (slot_c + 1) & (F3_RING - 1);                                                                 \
            const f16x8* An = reinterpret_cast<const f16x8*>(smem + slot_n * F3_SLOT) + lane;                                \
            /* fragments 0-7; fillers: fragments 8-15 of this stage; the query fragments of the stage three ahead */         \
            F3_MM2(0, U, Ac[8 * 64], Ac[9 * 64], F3_BL(bq[((U) + 3) & 3][0], qsrc_b, 0, 0), F3_BL(bq[((U) + 3) & 3][0], qsrc_b, 0, 1)) \
            F3_MM2(2, U, Ac[10 * 64], Ac[11 * 64], F3_NONE, F3_NONE)                                                         \
            F3_MM2(4, U, Ac[12 * 64], Ac[13 * 64], F3_BL(bq[((U) + 3) & 3][1], qsrc_b, 1024, 0), F3_BL(bq[((U) + 3) & 3][1], qsrc_b, 1024, 1)) \
            F3_MM2(6, U, Ac[14 * 64], Ac[15 * 64], F3_NONE, F3_NONE)                                                         \
            KN_FENCE                                                                                                         \
            /* the next stage has landed: my requests (and my query fragments of it), then everyone's */                    \
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(bq[((U) + 1) & 3][0]), "+v"(bq[((U) + 1) & 3][1]) :: "memory");         \
            __builtin_amdgcn_s_barrier();   /* raw: __syncthreads() would drain the LDS-DMA copies (vmcnt(0)) */              \
            if (w == 0) cl_tick(cs, clock0 + st + (U), lane);   /* cluster soft sync, ahead of the stage's copies */         \
            /* fragments 8-15; fillers: fragments 0-7 of the next stage; the bank pieces of the stage four ahead */          \
            F3_MM2(8, U, An[0 * 64], An[1 * 64], F3_NONE, F3_NONE)                                                           \
            F3_MM2(10, U, An[2 * 64], An[3 * 64], if (w < 4) F3_COPY(0), if (w >= 4) F3_COPY(0))                             \
            F3_MM2(12, U, An[4 * 64], An[5 * 64], if (w < 4) F3_COPY(1), if (w >= 4) F3_COPY(1))                             \
            F3_MM2(14, U, An[6 * 64], An[7 * 64], F3_NONE, F3_NONE)                                                          \
            KN_FENCE                                                                                                         \
            F3_ADVANCE()                                                                                                     \
            KN_FENCE                                                                                                         \
            slot_c = slot_n;                                                                                                 \
        }
        for (int st = 0; st < total; st += 4) {
            F3_STAGE(0) F3_STAGE(1) F3_STAGE(2) F3_STAGE(3)
            ks += 4;
            if (ks == NS) {
#if defined(F16_ABL) && (F16_ABL & 1)
#pragma unroll
                for (int t = 0; t < 32; ++t) asm volatile("" :: "v"(acc[t]));   // timing only: no epilogue
#else
#ifndef F3_NO_COLD
                if (seg.first && bt == seg.b_tile0) cold_start16(acc, k, t0, t1);
#endif
                pool_epilogue16<EMAX>(acc, t0, t1, wl_s, wl_i, sc, w * 32, lane, k, (unsigned)bt, klw, pcnt);
#endif
                ks = 0; bt += bstride; cpar ^= 1;
                F3_INIT_TILE()
                F3_FIRST_FRAGMENTS()   // the next tile's first fragments again (not kept live across the epilogue)
            }
        }
        // the run-ahead requests still target the query-fragment registers: drain them while those registers are live
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]),
                     "+v"(bq[3][0]), "+v"(bq[3][1]) :: "memory");
#undef F3_STAGE
#undef F3_MM2
        if (w == 0) cl_publish(cs, seg.next_tile0 == 0x7FFFFFFF ? 0x7FFFFFFF : seg.next_tile0 * NS, lane);   // covers idle units
        if (lane < 32) a.state_cnt[(size_t)seg.slot * HB_QT + w * 32 + lane] = pcnt[w * 32 + lane];
        if (lane < 16) {
            a.state_thr[(size_t)seg.slot * HB_QT + q0] = t0;
            a.state_thr[(size_t)seg.slot * HB_QT + q1] = t1;
            floor_publish(a.gthr, seg.q_tile * HB_QT + q0, t0);
            floor_publish(a.gthr, seg.q_tile * HB_QT + q1, t1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    cl_finish(cs, a.cl_stats, w == 0, lane);
}

int hb_knn_f16s_launch(const knn16_args& args, int grid, hipStream_t s) {
    void (*fn)(knn16_args) = knn_f16s_kernel<4>;
    if (hb_ensure_dyn_lds((const void*)fn, F3_LDS_TOTAL)) return -1;
    fn<<<dim3((unsigned)grid), dim3(HB_THREADS), F3_LDS_TOTAL, s>>>(args);
    HB_HIP(hipGetLastError());
    return 0;
}
