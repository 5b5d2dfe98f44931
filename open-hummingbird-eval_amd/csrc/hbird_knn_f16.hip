// Row f5 of SURVEY.md 8: `use_fp16` (reference hbird/nn/search_faiss.py:7, 40: GpuIndexFlatConfig.useFloat16 --
// fp16 storage and fp16 GEMM with fp32 accumulation).
//
// Here: a CANDIDATE pass on fp16 copies of the bank and query fragment tiles with v_mfma_f32_32x32x16_f16 (16x the
// fp32 MFMA rate), fused running top-k' (k' = 64 .. 256 >= 2k) exactly like the fp32 kernel, followed by an EXACT
// re-rank of the k' candidates in the fp32 chain arithmetic of the fp32 kernel (every score one k-ascending fmaf
// chain on the fp32 tiles).  Indices and distances are therefore those of the fp32 definition whenever the true
// top-k lie within the fp16 top-k' -- their fp16 scores would have to be off by more than the gap between rank k
// and rank k' for that to fail.
//
// The per-query candidate pools live in HBM (the WIDE path of the fp32 kernels), only the thresholds stay in registers.
#include "hbird_knn_dev.h"
#include <algorithm>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


// fp32 fragment tiles -> fp16 fragment tiles.  fp16 block(rt, g16) = 32 rows x 16 k = 1 KiB, element (i, kk) at half
// index ((kk >> 3) * 32 + i) * 8 + (kk & 7): lane l = h*32 + i reads 8 halves = k = 16 g16 + 8h + 0..7, the A/B
// operand of one 32x32x16 MFMA.  One thread per 8 output halves.
// *overflow (optional) is raised when a FINITE fp32 value has no finite fp16 image (|x| > 65504): the candidate pass'
// certificate assumes finite fp16 operands, so such a bank / query is served by the fp32 kernel.  (A NaN stays a NaN in
// both precisions and is excluded by both kernels alike.)
__global__ __launch_bounds__(256) void tiles_to_f16_kernel(const float* __restrict__ t32, int g8, _Float16* __restrict__ t16,
                                                           int g16, int64_t n_row_tiles, int64_t rt0, int* __restrict__ overflow) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one per (row tile, g16, h, i)
    const int64_t total = n_row_tiles * g16 * 64;
    if (gid >= total) return;
    const int i = (int)(gid & 31), h = (int)((gid >> 5) & 1);
    const int64_t blk = gid >> 6;
    const int g = (int)(blk % g16);
    const int64_t rt = rt0 + blk / g16;
    f16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * g + 8 * h + j;
        const int gg = k >> 3, kk = k & 7;
        float v = 0.0f;
        if (gg < g8) v = t32[((rt * g8) + gg) * HB_BLK + ((kk & 1) * 32 + i) * 4 + (kk >> 1)];
        out[j] = (_Float16)v;
        if (overflow && fabsf(v) > 65504.0f && fabsf(v) < INFINITY) *overflow = 1;
    }
    reinterpret_cast<f16x8*>(t16)[(rt * g16 + g) * 64 + h * 32 + i] = out;
}

int hb_launch_tiles_to_f16(const float* t32, int g8, _Float16* t16, int g16, int64_t n_row_tiles, int64_t rt0, int* overflow,
                           hipStream_t s) {
    const int64_t total = n_row_tiles * g16 * 64;
    if (total == 0) return 0;
    tiles_to_f16_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(t32, g8, t16, g16, n_row_tiles, rt0, overflow);
    HB_HIP(hipGetLastError());
    return 0;
}

// ---- the candidate kernel ---------------------------------------------------------------------------------------------
// (Its first design staged BOTH operands through a 4-slot LDS ring of k32 stages and ran at 0.34 of the nominal fp16 peak; it served
// pools beyond 256 entries until round 4 and is gone now -- this kernel's <8> instantiation compacts pools of up to 512.)
// What bounded that first design was measured in round 2 (profiles/r02): NOT the fabric -- the L2's average read latency
// at the memory side is ~610 cycles under this kernel's load (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ), and 41 % fewer
// fabric reads (L2-sharing clusters) bought 1.7 % -- but the LDS: per k16 group the workgroup reads 72 KiB of fragments
// (>= 288 cycles) and takes in 16 KiB of LDS-DMA copies (281 cycles: an LDS-DMA byte costs the LDS about four times a
// read byte) against 512 cycles of matrix work per SIMD, and the two streams serialise.
// A wave's query fragment (32 queries x k16 = 1 KiB per group) is read by that wave ONLY: staging it through LDS buys no
// reuse.  Here it goes straight from global memory into registers (inline-asm global_load_dwordx4, four k32 stages ahead,
// counted by hand with the copies), which halves the LDS-DMA traffic and drops the ninth fragment read of every group;
// the ring then holds bank fragments only (8 slots x 16 KiB) and the bank fragments need ONE register set: fragment t is
// re-loaded right after MFMA t has issued, 8 MFMAs before its next use.  Per group: 64 KiB of reads + 8 KiB of copies.
// Stages are unrolled by four so that the register buffers of the query fragments have static indices.
//
// Round 4: the stage loop's SCALAR side.  A model of this loop with the same traffic on random operands
// (tools/ubench/f16_stage_model.hip: 16 MFMAs + 16 fragment reads + 2 query-fragment loads + 2 copies per wave and k32 stage, one
// barrier) runs at 0.55 of the nominal fp16 peak where this kernel's loop, without its epilogue, ran at 0.46: the difference was
// not memory but instructions -- per copy a 64-bit multiply chain for the source address, a per-lane 64-bit VALU add (the
// builtin's flat-address form), v_readlane reloads of 78 spilled SGPRs, compare + branch pairs for the two wave classes and the
// cluster tick's tests in every stage.  Now: the fetch position is three running 64-bit scalars (query fragments, bank row
// tile, row-init values) that advance by constants, the copies use the saddr form (wave-uniform base + 32-bit lane offset, M0
// = wave-uniform LDS address) from inline asm, every wave issues at the same MFMA gaps, the cluster tick sits in the one
// stage of four that can act, and arguments used only at segment boundaries are re-read from the kernarg segment (HB_KARG).
#define F2_RING 8
#define F2_SLOT 16384                                  // bank fragments of one k32 stage: [row tile 0..7][group 0..1][1 KiB]
#define F2_BINIT (F2_RING * F2_SLOT)
#define F2_SCRATCH (F2_BINIT + 2048)
#define F2_PCNT (F2_SCRATCH + 8192)
#define F2_CLWORDS (F2_PCNT + 1024)                    // landing zone of the cluster progress poll
#define F2_LDS_TOTAL (F2_CLWORDS + 64)
static_assert(F2_LDS_TOTAL <= 160 * 1024, "LDS budget");

// one 1 KiB LDS-DMA piece: global (wave-uniform 64-bit base + 32-bit lane offset) -> LDS (wave-uniform address in M0 + 16 * lane).
// M0 is on the clobber list so that the compiler never assumes a value of its own survives the statement (hipcc notes that it will
// not SAVE a reserved register around it -- nothing to save: -Wno-inline-asm in the Makefile).
#define F2_DMA(SRC, LDS_ADDR, VOFF)                                                                                          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(VOFF), "s"(SRC), "s"(LDS_ADDR) : "memory", "m0");
// a query-fragment load into registers (untracked by the compiler: the hand-counted vmcnt below names the registers it releases)
#define F2_BL(REG, SRC, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "+v"(REG) : "v"(lane_off), "s"(SRC) : "memory");

// EMAX: pool capacity / 64 that the instantiation can compact (registers of the rare compaction path)
template <int EMAX>
__global__ __launch_bounds__(HB_THREADS, 2) void knn_f16v2_kernel(knn16_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    float* sc = reinterpret_cast<float*>(smem + F2_SCRATCH) + w * 256;
    int* pcnt = reinterpret_cast<int*>(smem + F2_PCNT);
    const int g16 = a.g16;
    const int NS = g16 / 2;   // k32 stages per bank tile, a multiple of 4 (dp16 is a multiple of 128)
    const int myq = w * 32 + (lane & 31);
    const unsigned lane_off = (unsigned)lane * 16u;
    const unsigned lds_0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lds_w = lds_0 + (unsigned)w * 2048u;   // this wave's two pieces of a ring slot
    cl_sync cs = cl_init(a.wg_member, a.prog, a.cl, a.lag, blockIdx.x, w == 0, smem + F2_CLWORDS);

    wg_stamp<knn16_args>(0);
    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_end[blockIdx.x];   // this launch's share of the block's segments (phases: hb_launch_knn)
    // "everything before my first segment is done" (a member without any work: everything)
    if (w == 0) cl_publish(cs, seg_begin < seg_end ? a.segs[seg_begin].tile0 * NS : 0x7FFFFFFF, lane);
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = HB_KARG(knn16_args, segs)[si];
        const int k = HB_KARG(knn16_args, k), klw = HB_KARG(knn16_args, klw);
        float* wl_s = HB_KARG(knn16_args, state_s) + (size_t)seg.slot * HB_QT * klw;
        unsigned* wl_i = HB_KARG(knn16_args, state_i) + (size_t)seg.slot * HB_QT * klw;
        float thr;
        {
            const knn_args_pool_view pv{HB_KARG(knn16_args, state_cnt), HB_KARG(knn16_args, state_thr)};
            thr = pool_begin(pv, seg.slot, seg.first, pcnt, myq, lane);
            thr = fmaxf(thr, floor_load(HB_KARG(knn16_args, gthr), seg.q_tile * HB_QT + myq));
        }
        const int total = seg.n_tiles * NS, clock0 = seg.tile0 * NS;
        f32x16 acc[8];
        f16x8 fa[8];        // bank fragments of the current group (one set)
        f16x8 bq[4][2];     // query fragments of four stages, two k16 groups each

        // What one wave requests for one stage, in this order: query fragment B0 (group 0 of the stage; registers), two 1 KiB
        // bank pieces C0, C1 (row tile w, both groups; LDS-DMA) -- all three during group 1 of the stage four earlier -- and
        // query fragment B1 (group 1) during group 0 of the stage three earlier (its register is free only then).  Wave 0
        // adds the row-init values behind the first stage of a tile.
        // The fetch position, all wave-uniform: qf / bf / bi point at the NEXT stage to request (this wave's query fragments
        // [g16][1 KiB]; row tile w of the bank tile: [g16][1 KiB], the next row tile behind it; the tile's row-init values),
        // qf1 at the stage whose B1 is still to be requested.  A stage advances qf and bf by 2 KiB; at a tile's end qf returns
        // to the wave's first fragment and bf jumps from the end of row tile w to row tile w of the next tile of the segment.
        const char* const qbase = reinterpret_cast<const char*>(a.q16) + (size_t)(seg.q_tile * 8 + w) * g16 * 1024;
        const char* qf = qbase;
        const char* qf1 = qbase;
        const char* bf = reinterpret_cast<const char*>(a.bank16) + ((size_t)seg.b_tile0 * 8 + w) * g16 * 1024;
        const float* bi = a.binit + (size_t)seg.b_tile0 * HB_BT;
        const long long b_wrap = ((long long)seg.stride * 8 - 1) * g16 * 1024;
        const int bi_step = seg.stride * HB_BT;
        int fks = 0, left = total;
        unsigned slot_f = 0, fpar = 0;        // ring slot being filled (byte offset), parity of the row-init buffer being filled
#define F2_REQ_B0(U) F2_BL(bq[U][0], qf, 0)
#define F2_REQ_B1(U) F2_BL(bq[U][1], qf1, 1024)
#define F2_REQ_C(I) F2_DMA(bf, lds_w + slot_f + (I) * 1024u, lane_off + (I) * 1024u)
#define F2_ADVANCE()                                                                                                         \
        {                                                                                                                    \
            if (fks == 0 && w == 0) F2_DMA(bi, lds_0 + F2_BINIT + fpar, lane_off)                                            \
            qf1 = qf;                                                                                                        \
            if (--left > 0) {                                                                                                \
                qf += 2048; bf += 2048;                                                                                      \
                if (++fks == NS) { fks = 0; qf = qbase; bf += b_wrap; bi += bi_step; fpar ^= 1024u; }                        \
            }                                                                                                                \
            slot_f = (slot_f + F2_SLOT) & (F2_RING * F2_SLOT - 1);                                                           \
        }
        // vmcnt by hand.  Order of a wave's requests: ... B1(s+3) | B0(s+4) C0 C1 | B1(s+4) | B0(s+5) ... (wave 0: now and
        // then one more, which only makes a wait stricter).  At the barrier in the middle of stage s everything of stage
        // s + 1 must have landed; its youngest request is B1(s+1), behind it come B0 C0 C1 (s+2), B1(s+2), B0 C0 C1 (s+3) and
        // B1(s+3) -> "all but the newest 8".  Past the last stage the fetch position stays put (same requests again, results
        // unused), so the count never changes.
        F2_REQ_B0(0) F2_REQ_C(0) F2_REQ_C(1) F2_ADVANCE() F2_REQ_B1(0)
        F2_REQ_B0(1) F2_REQ_C(0) F2_REQ_C(1) F2_ADVANCE() F2_REQ_B1(1)
        F2_REQ_B0(2) F2_REQ_C(0) F2_REQ_C(1) F2_ADVANCE() F2_REQ_B1(2)
        F2_REQ_B0(3) F2_REQ_C(0) F2_REQ_C(1) F2_ADVANCE()
        // stage 0 (and B1 of it) has landed: eleven younger requests
        asm volatile("s_waitcnt vmcnt(11)" : "+v"(bq[0][0]), "+v"(bq[0][1]) :: "memory");
        __syncthreads();
        unsigned slot_c = 0;                  // ring slot being computed (byte offset)
        int ks = 0, bt = seg.b_tile0, cpar = 0;
        bool bulk = seg.tile0 < 16;   // loose floors at the start of a search: the epilogue's quarter loop right away (pool_epilogue_scan)
#define F2_INIT_TILE()                                                                                                       \
        {                                                                                                                    \
            const f32x4* bi_ = reinterpret_cast<const f32x4*>(smem + F2_BINIT + cpar * 1024);                                \
            _Pragma("unroll") for (int t = 0; t < 8; ++t)                                                                    \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                              \
                    const f32x4 v = bi_[8 * t + 2 * g + h];                                                                  \
                    acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1]; acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3]; \
                }                                                                                                            \
        }
        {
            const f16x8* A = reinterpret_cast<const f16x8*>(smem) + lane;
#pragma unroll
            for (int t = 0; t < 8; ++t) fa[t] = A[(t * 2) * 64];
        }
        F2_INIT_TILE()
#define F2_MM(T, B) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[T], B, acc[T], 0, 0, 0);
#define F2_STAGE(U)                                                                                                          \
        {                                                                                                                    \
            const f16x8* Ac = reinterpret_cast<const f16x8*>(smem + slot_c) + lane;                                          \
            const unsigned slot_n = (slot_c + F2_SLOT) & (F2_RING * F2_SLOT - 1);                                            \
            const f16x8* An = reinterpret_cast<const f16x8*>(smem + slot_n) + lane;                                          \
            /* group 0; filler after MFMA t: fragment t of group 1; B1 of the stage three ahead */                           \
            KN_FENCE F2_MM(0, bq[U][0]) KN_FENCE fa[0] = Ac[(0 * 2 + 1) * 64]; F2_REQ_B1((U + 3) & 3)                           \
            KN_FENCE F2_MM(1, bq[U][0]) KN_FENCE fa[1] = Ac[(1 * 2 + 1) * 64];                                                  \
            KN_FENCE F2_MM(2, bq[U][0]) KN_FENCE fa[2] = Ac[(2 * 2 + 1) * 64];                                                  \
            KN_FENCE F2_MM(3, bq[U][0]) KN_FENCE fa[3] = Ac[(3 * 2 + 1) * 64];                                                  \
            KN_FENCE F2_MM(4, bq[U][0]) KN_FENCE fa[4] = Ac[(4 * 2 + 1) * 64];                                                  \
            KN_FENCE F2_MM(5, bq[U][0]) KN_FENCE fa[5] = Ac[(5 * 2 + 1) * 64];                                                  \
            KN_FENCE F2_MM(6, bq[U][0]) KN_FENCE fa[6] = Ac[(6 * 2 + 1) * 64];                                                  \
            KN_FENCE F2_MM(7, bq[U][0]) KN_FENCE fa[7] = Ac[(7 * 2 + 1) * 64];                                                  \
            KN_FENCE                                                                                                         \
            /* the next stage has landed: my requests (and my query fragments of it), then everyone's */                    \
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(bq[(U + 1) & 3][0]), "+v"(bq[(U + 1) & 3][1]), "+v"(bq[U][1]) :: "memory"); \
            __builtin_amdgcn_s_barrier();   /* raw s_barrier: __syncthreads() would drain the LDS-DMA copies (vmcnt(0)) */   \
            /* group 1; filler after MFMA t: fragment t of the next stage's group 0; B0 C0 C1 of the stage four ahead */     \
            if ((U) == 0 && w == 0) cl_tick(cs, clock0 + st, lane);   /* cluster soft sync, ahead of the stage's requests; acts when the clock is a multiple of 4: st is, clock0 is (NS is) */ \
            KN_FENCE F2_MM(0, bq[U][1]) KN_FENCE fa[0] = An[(0 * 2) * 64]; F2_REQ_B0(U)                                         \
            KN_FENCE F2_MM(1, bq[U][1]) KN_FENCE fa[1] = An[(1 * 2) * 64];                                                      \
            KN_FENCE F2_MM(2, bq[U][1]) KN_FENCE fa[2] = An[(2 * 2) * 64]; F2_REQ_C(0)                                          \
            KN_FENCE F2_MM(3, bq[U][1]) KN_FENCE fa[3] = An[(3 * 2) * 64];                                                      \
            KN_FENCE F2_MM(4, bq[U][1]) KN_FENCE fa[4] = An[(4 * 2) * 64]; F2_REQ_C(1)                                          \
            KN_FENCE F2_MM(5, bq[U][1]) KN_FENCE fa[5] = An[(5 * 2) * 64];                                                      \
            KN_FENCE F2_MM(6, bq[U][1]) KN_FENCE fa[6] = An[(6 * 2) * 64];                                                      \
            KN_FENCE F2_MM(7, bq[U][1]) KN_FENCE fa[7] = An[(7 * 2) * 64];                                                      \
            KN_FENCE                                                                                                         \
            F2_ADVANCE()                                                                                                     \
            KN_FENCE                                                                                                         \
            slot_c = slot_n;                                                                                                 \
        }
        for (int st = 0; st < total; st += 4) {
            F2_STAGE(0) F2_STAGE(1) F2_STAGE(2) F2_STAGE(3)
            ks += 4;
            if (ks == NS) {
                // a slot's first tile: filter below the k'-th largest of its 256 scores (bisection over the accumulators)
                // instead of appending all 256 rows of every query through the overflow path
                if (seg.first && bt == seg.b_tile0 && cold_start_needed(thr)) thr = fmaxf(thr, cold_start_threshold(acc, k));
                pool_epilogue_scan<EMAX>(acc, thr, wl_s, wl_i, sc, w * 32, lane, k, (unsigned)bt, klw, pcnt, bulk);
                ks = 0; bt += seg.stride; cpar ^= 1;
                F2_INIT_TILE()
                {   // the next tile's first fragments again (rather than kept live across the epilogue: fewer registers)
                    const f16x8* A = reinterpret_cast<const f16x8*>(smem + slot_c) + lane;
#pragma unroll
                    for (int t = 0; t < 8; ++t) fa[t] = A[(t * 2) * 64];
                }
            }
        }
        // the run-ahead requests still target the query-fragment registers: drain them while those registers are live
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]),
                     "+v"(bq[3][0]), "+v"(bq[3][1]) :: "memory");
#undef F2_STAGE
#undef F2_MM
#undef F2_INIT_TILE
#undef F2_ADVANCE
#undef F2_REQ_B0
#undef F2_REQ_B1
#undef F2_REQ_C
        if (w == 0) cl_publish(cs, seg.next_tile0 == 0x7FFFFFFF ? 0x7FFFFFFF : seg.next_tile0 * NS, lane);   // covers idle units
        {
            const knn_args_pool_view pv{HB_KARG(knn16_args, state_cnt), HB_KARG(knn16_args, state_thr)};
            pool_end(pv, seg.slot, pcnt, thr, myq, lane);
        }
        if (lane < 32) floor_publish(HB_KARG(knn16_args, gthr), seg.q_tile * HB_QT + myq, thr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    cl_finish(cs, a.cl_stats, w == 0, lane);
    wg_stamp<knn16_args>(1);
}

// What the re-rank leaves for an ESCALATION of the queries whose certificate fails (hb_launch_knn), and what it takes from the pass before:
//   seed_in  [query] (second pass only): the fp16-score floor this pass ran under -- every row with an fp16 score >= it was offered to the
//            pools, so a candidate list that is NOT full holds every such row: the rows outside score below the floor in fp16, hence below
//            floor + E exactly, and the answer is exact when its k-th best exceeds that;
//   kth_out  [query]: the exact k-th best score among this pass's candidates -- a lower bound of the true k-th best: the floor of an fp32
//            search of this query (ties pass: floor_from_key);
//   floor_out[query]: kth - 1.001 E, a hair lower: every row that can still enter the top k has an exact score >= kth, hence an fp16 score
//            above this -- the floor of a second, wider fp16 pass (0.001 E is twenty times the rounding of these few operations).
struct hb_rerank_seeds { const float* seed_in; float* kth_out; float* floor_out; };
__device__ __forceinline__ bool hb_rerank_finish(const hb_rerank_seeds& sd, int64_t qi, bool ok, bool list_not_full, bool have_kth, float s, float E) {
    if (sd.seed_in && list_not_full && have_kth && !ok) ok = s > sd.seed_in[qi] + E;
    if (sd.kth_out) {
        const bool fin = have_kth && E < INFINITY && fabsf(s) < INFINITY;      // (false for NaN as well)
        float f = s - 1.001f * E;
        f = f - fabsf(f) * 2.4e-7f - 1e-37f;
        sd.kth_out[qi] = fin ? s : -INFINITY;
        sd.floor_out[qi] = fin ? f : -INFINITY;
    }
    return ok;
}

// Exact re-rank: one wave per query; lane j scores candidates j, j+64, ... with the fp32 chain arithmetic of the fp32
// kernel (acc = row init; acc = fmaf(q_k, b_k, acc) for k ascending over the fp32 fragment tiles), then the wave
// ranks them by (score desc, id asc) and writes the best k.
__global__ __launch_bounds__(256) void rerank_kernel(const float* __restrict__ tiles, const float* __restrict__ binit, int g8,
                                                     int d, const float* __restrict__ q, const float* __restrict__ qn2,
                                                     const int64_t* __restrict__ cand, const float* __restrict__ cand_score,
                                                     const float* __restrict__ qnorm, const float* __restrict__ bmax,
                                                     unsigned char* __restrict__ certified, int kc, int64_t nq, int k,
                                                     int64_t id_base, int metric, int out_metric, int64_t ntotal,
                                                     int64_t* __restrict__ out_idx, float* __restrict__ out_dist, hb_rerank_seeds sd) {
    __shared__ float s_sc[4][256];
    __shared__ int64_t s_id[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wv;
    if (qi >= nq) return;   // wave-uniform
    const float* qr = q + qi * (int64_t)d;
    // E >= |fp16 score - exact score| of any row of this query (both inputs rounded to fp16: relative 2^-10 per product,
    // Cauchy-Schwarz over the row; fp32 accumulation: D * 2^-23) -- the certificate below, and which candidates need an exact score at
    // all: the list comes sorted by fp16 score, its first k have exact scores >= (k-th fp16 score) - E, so a candidate whose fp16
    // score is more than 2E below the k-th's is exactly below k of them and cannot be in the answer.  Its row is not read (a row
    // is 2 x D/8 sixteen-byte pieces 512 B apart: the re-rank is bound by the sectors it touches; 50,176 x 384, 12,544 queries:
    // 0.96 ms of a 2.4 ms search).  A non-finite E (query norm) skips nothing, and fails the certificate.
    const float E = qnorm[qi] * bmax[0] * (1.05f / 1024.0f + (float)d * 2.4e-7f)
                    + (qnorm[qi] + bmax[0]) * sqrtf((float)d) * 6e-8f                 // fp16 subnormal inputs
                    + (metric == 1 ? (float)d * 1.2e-7f * 0.5f * bmax[0] * bmax[0] : 0.0f)  // |row init| in the sums
                    + 1e-30f;
    const float cut = (k <= kc && cand[qi * (int64_t)kc + (k - 1)] >= 0) ? cand_score[qi * (int64_t)kc + (k - 1)] - 2.0f * E : -INFINITY;
    for (int c = lane; c < kc; c += 64) {
        const int64_t row = cand[qi * (int64_t)kc + c];
        float acc = -INFINITY;
        if (row >= 0 && !(c >= k && cand_score[qi * (int64_t)kc + c] < cut)) {
            acc = binit[row];
            const float* base = tiles + ((row >> 5) * (int64_t)g8) * HB_BLK + (int)(row & 31) * 4;
            for (int g = 0; g < g8; ++g) {
                const f32x4 e = *reinterpret_cast<const f32x4*>(base + (int64_t)g * HB_BLK);         // k = 8g + 0,2,4,6
                const f32x4 o = *reinterpret_cast<const f32x4*>(base + (int64_t)g * HB_BLK + 128);   // k = 8g + 1,3,5,7
                const int k0 = 8 * g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (k0 + 2 * j < d) acc = fmaf(qr[k0 + 2 * j], e[j], acc);
                    if (k0 + 2 * j + 1 < d) acc = fmaf(qr[k0 + 2 * j + 1], o[j], acc);
                }
            }
        }
        s_sc[wv][c] = acc;
        s_id[wv][c] = row;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to its lanes
    for (int c = lane; c < kc; c += 64) {
        const float s = s_sc[wv][c];
        const int64_t id = s_id[wv][c];
        int rank = 0;
        for (int j = 0; j < kc; ++j) {
            const float sj = s_sc[wv][j];
            const int64_t ij = s_id[wv][j];
            bool better;
            if (ij < 0 || id < 0) better = (ij >= 0 && id < 0) || (ij < 0 && id < 0 && j < c);
            else better = (sj > s) || (sj == s && (ij < id || (ij == id && j < c)));
            rank += better;
        }
        if (rank == k - 1) {
            // Certificate: every row outside the candidate list has an fp16 score <= the kc-th candidate's, hence an
            // exact score <= that + E.  If the exact k-th best is strictly above that bound, no outside row can enter the
            // top k: the answer IS the fp32 answer.
            // The argument needs finite fp16 operands: a query with |q_i| > 65504 becomes inf in fp16 and its scores inf / NaN
            // (||q|| <= 65504 rules that out; a NaN / inf norm fails the test too), and a candidate list that is not full
            // although the bank has kc rows has lost rows to NaN / -inf fp16 scores that nothing bounds.
            const int64_t last = cand[qi * (int64_t)kc + kc - 1];
            const bool finite_q = qnorm[qi] <= 65504.0f;
            bool ok = last < 0 && ntotal < kc && finite_q;   // fewer than kc rows exist: every row was a candidate
            if (last >= 0 && id >= 0 && finite_q) {
                ok = s > cand_score[qi * (int64_t)kc + kc - 1] + E;
            }
            ok = hb_rerank_finish(sd, qi, ok, last < 0, id >= 0 && finite_q, s, E);
            certified[qi] = ok ? 1 : 0;
        }
        if (rank < k) {
            const int64_t o = qi * (int64_t)k + rank;
            if (id < 0) { out_idx[o] = -1; out_dist[o] = out_metric == 1 ? INFINITY : -INFINITY; }
            else {
                out_idx[o] = id + id_base;
                if (out_metric == 1) { const float d2 = fmaf(-2.0f, s, qn2[qi]); out_dist[o] = d2 > 0.0f ? d2 : 0.0f; }
                else out_dist[o] = s;
            }
        }
    }
}

int hb_launch_rerank(const float* tiles, const float* binit, int g8, int d, const float* q, const float* qn2,
                     const int64_t* cand, const float* cand_score, const float* qnorm, const float* bmax,
                     unsigned char* certified, int kc, int64_t nq, int k, int64_t id_base, int metric, int out_metric,
                     int64_t ntotal, int64_t* out_idx, float* out_dist, hipStream_t s, const float* seed_in, float* kth_out, float* floor_out) {
    if (nq == 0) return 0;
    if (kc > 256) return hb_fail("hb_index_search: too many candidates for the re-rank kernel");
    rerank_kernel<<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s>>>(tiles, binit, g8, d, q, qn2, cand, cand_score, qnorm, bmax,
                                                                     certified, kc, nq, k, id_base, metric, out_metric, ntotal, out_idx, out_dist,
                                                                     hb_rerank_seeds{seed_in, kth_out, floor_out});
    HB_HIP(hipGetLastError());
    return 0;
}

// ---- the re-rank on a row-major copy of the bank ---------------------------------------------------------------------
// In the fragment tiles a row is 2 x D/8 sixteen-byte pieces 512 B apart: the re-rank above pulls a 128-byte line for every piece it
// uses (FETCH_SIZE of the kernel at 300,000 x 768, 12,544 queries: 11.6 GB for 1.35 GB of rows, 1.72 of the search's 8.05 ms; k = 90:
// a third of the search).  Where the bank is small enough (hb_launch_knn: 2.5 x the bank within 55 % of the device) the bank is kept a second time as plain rows [row][rs] and the
// re-rank reads whole lines: eight lanes fetch one candidate row's 128 B (32 k) with one instruction, eight rows per instruction;
// the pieces go through a padded LDS image from which lane t takes row t's 32 values for its serial chain (same chain, same order, same
// bits as rerank_kernel).  The next 32 k of every row are in flight while a chunk is consumed; the query's chunk travels as one more
// row of the image.
__global__ __launch_bounds__(256) void tiles_to_rows_kernel(const float* __restrict__ t32, int g8, float* __restrict__ rows, int rs,
                                                            int64_t rt0) {
    __shared__ float s[4 * HB_BLK];
    const int64_t rt = rt0 + blockIdx.x;
    const int g0 = blockIdx.y * 4, ng = min(4, g8 - g0);
    const int t = threadIdx.x;
    if (t < ng * 64) reinterpret_cast<f32x4*>(s)[t] = reinterpret_cast<const f32x4*>(t32 + (rt * g8 + g0) * HB_BLK)[t];
    __syncthreads();
    const int i = t >> 3, c = t & 7;          // row of the tile, four consecutive k
    f32x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kq = 4 * c + j, gl = kq >> 3, kk = kq & 7;
        out[j] = gl < ng ? s[gl * HB_BLK + ((kk & 1) * 32 + i) * 4 + (kk >> 1)] : 0.0f;
    }
    if (8 * g0 + 4 * c < rs) *reinterpret_cast<f32x4*>(rows + (rt * 32 + i) * (int64_t)rs + 8 * g0 + 4 * c) = out;
}

int hb_launch_tiles_to_rows(const float* t32, int g8, float* rows, int rs, int64_t n_row_tiles, int64_t rt0, hipStream_t s) {
    if (n_row_tiles <= 0) return 0;
    tiles_to_rows_kernel<<<dim3((unsigned)n_row_tiles, (unsigned)((rs / 8 + 3) / 4)), dim3(256), 0, s>>>(t32, g8, rows, rs, rt0);
    HB_HIP(hipGetLastError());
    return 0;
}

#define RRW_NONE 0xFFFFFFFFu
#define RRW_STRIDE 36      // dwords per row of the LDS image: 128 B of values + 16 B (lane t's b128 reads of row t: no bank conflicts)
__global__ __launch_bounds__(256) void rerank_rows_kernel(const float* __restrict__ rows, int rs, const float* __restrict__ binit,
                                                          int d, const float* __restrict__ q, const float* __restrict__ qn2,
                                                          const int64_t* __restrict__ cand, const float* __restrict__ cand_score,
                                                          const float* __restrict__ qnorm, const float* __restrict__ bmax,
                                                          unsigned char* __restrict__ certified, int kc, int64_t nq, int k,
                                                          int64_t id_base, int metric, int out_metric, int64_t ntotal,
                                                          int64_t* __restrict__ out_idx, float* __restrict__ out_dist, hb_rerank_seeds sd) {
    __shared__ float s_sc[4][256];
    __shared__ unsigned s_id[4][256];                    // bank rows (below 2^32); RRW_NONE: no candidate
    __shared__ int s_act[4][256];
    __shared__ __attribute__((aligned(16))) float s_img[4][66 * RRW_STRIDE];   // 64 rows + the query's + one of padding: 50 KB per workgroup, three per CU
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wv;
    if (qi >= nq) return;   // wave-uniform
    const float* qr = q + qi * (int64_t)d;
    // (E and the skip rule: rerank_kernel)
    const float E = qnorm[qi] * bmax[0] * (1.05f / 1024.0f + (float)d * 2.4e-7f)
                    + (qnorm[qi] + bmax[0]) * sqrtf((float)d) * 6e-8f
                    + (metric == 1 ? (float)d * 1.2e-7f * 0.5f * bmax[0] * bmax[0] : 0.0f)
                    + 1e-30f;
    const float cut = (k <= kc && cand[qi * (int64_t)kc + (k - 1)] >= 0) ? cand_score[qi * (int64_t)kc + (k - 1)] - 2.0f * E : -INFINITY;
    // the candidates that need an exact score, compacted: s_act[0 .. n_act)
    int n_act = 0;
    for (int c0 = 0; c0 < kc; c0 += 64) {
        const int c = c0 + lane;
        int64_t row = -1;
        bool act = false;
        if (c < kc) {
            row = cand[qi * (int64_t)kc + c];
            act = row >= 0 && !(c >= k && cand_score[qi * (int64_t)kc + c] < cut);
            s_sc[wv][c] = -INFINITY;
            s_id[wv][c] = row >= 0 ? (unsigned)row : RRW_NONE;
        }
        const unsigned long long m = __ballot(act);
        if (act) s_act[wv][n_act + __popcll(m & ((1ull << lane) - 1ull))] = c;
        n_act += __popcll(m);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float* img = s_img[wv];
    const int rr = lane >> 3, chn = lane & 7;           // loader role: row rr of an instruction's eight, 16-byte piece chn of its 128 B
    const int nch = (d + 31) >> 5;
    for (int b0 = 0; b0 < n_act; b0 += 64) {
        const int nb = min(64, n_act - b0);             // rows of this batch; image row nb is the query's chunk
        const int nj = (nb + 1 + 7) >> 3;               // load instructions per chunk
        const int c = lane < nb ? s_act[wv][b0 + lane] : -1;
        float acc = c >= 0 ? binit[s_id[wv][c]] : 0.0f;
        const float* src[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int li = 8 * j + rr;
            src[j] = li < nb ? rows + (int64_t)s_id[wv][s_act[wv][b0 + li]] * rs + 4 * chn : (li == nb ? qr + 4 * chn : nullptr);
        }
        f32x4 R[9];
        // chunk `ch` of every row of the batch -> R (the query's last chunk may end inside the piece: d need not be a multiple of 4)
#define RRW_LOAD(CH)                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 9; ++j) {                                                                      \
            if (j < nj) {                                                                                                    \
                R[j] = f32x4{0.f, 0.f, 0.f, 0.f};                                                                            \
                if (src[j]) {                                                                                                \
                    const int k0 = 32 * (CH) + 4 * chn;                                                                      \
                    if (8 * j + rr < nb || k0 + 4 <= d) R[j] = *reinterpret_cast<const f32x4*>(src[j] + 32 * (CH));          \
                    else { _Pragma("unroll") for (int e = 0; e < 4; ++e) if (k0 + e < d) R[j][e] = src[j][32 * (CH) + e]; }   \
                }                                                                                                            \
            }                                                                                                                \
        }
        RRW_LOAD(0)
        for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
            for (int j = 0; j < 9; ++j)
                if (j < nj && 8 * j + rr <= nb) *reinterpret_cast<f32x4*>(img + (8 * j + rr) * RRW_STRIDE + 4 * chn) = R[j];
            if (ch + 1 < nch) { RRW_LOAD(ch + 1) }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the image is written
            __builtin_amdgcn_wave_barrier();
            if (c >= 0) {
                const f32x4* xr = reinterpret_cast<const f32x4*>(img + lane * RRW_STRIDE);
                const f32x4* qv = reinterpret_cast<const f32x4*>(img + nb * RRW_STRIDE);
                if (32 * ch + 32 <= d) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f32x4 x = xr[i], y = qv[i];
                        acc = fmaf(y[0], x[0], acc); acc = fmaf(y[1], x[1], acc); acc = fmaf(y[2], x[2], acc); acc = fmaf(y[3], x[3], acc);
                    }
                } else {
                    for (int i = 0; 32 * ch + i < d; ++i) acc = fmaf(img[nb * RRW_STRIDE + i], img[lane * RRW_STRIDE + i], acc);
                }
            }
            __builtin_amdgcn_wave_barrier();      // everyone has read the image before the next chunk overwrites it
        }
#undef RRW_LOAD
        if (c >= 0) s_sc[wv][c] = acc;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to its lanes
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < kc; c += 64) {
        const float s = s_sc[wv][c];
        const int64_t id = s_id[wv][c] == RRW_NONE ? -1 : (int64_t)s_id[wv][c];
        int rank = 0;
        for (int j = 0; j < kc; ++j) {
            const float sj = s_sc[wv][j];
            const int64_t ij = s_id[wv][j] == RRW_NONE ? -1 : (int64_t)s_id[wv][j];
            bool better;
            if (ij < 0 || id < 0) better = (ij >= 0 && id < 0) || (ij < 0 && id < 0 && j < c);
            else better = (sj > s) || (sj == s && (ij < id || (ij == id && j < c)));
            rank += better;
        }
        if (rank == k - 1) {   // the certificate: rerank_kernel
            const int64_t last = cand[qi * (int64_t)kc + kc - 1];
            const bool finite_q = qnorm[qi] <= 65504.0f;
            bool ok = last < 0 && ntotal < kc && finite_q;
            if (last >= 0 && id >= 0 && finite_q) ok = s > cand_score[qi * (int64_t)kc + kc - 1] + E;
            ok = hb_rerank_finish(sd, qi, ok, last < 0, id >= 0 && finite_q, s, E);
            certified[qi] = ok ? 1 : 0;
        }
        if (rank < k) {
            const int64_t o = qi * (int64_t)k + rank;
            if (id < 0) { out_idx[o] = -1; out_dist[o] = out_metric == 1 ? INFINITY : -INFINITY; }
            else {
                out_idx[o] = id + id_base;
                if (out_metric == 1) { const float d2 = fmaf(-2.0f, s, qn2[qi]); out_dist[o] = d2 > 0.0f ? d2 : 0.0f; }
                else out_dist[o] = s;
            }
        }
    }
}

int hb_launch_rerank_rows(const float* rows, int rs, const float* binit, int d, const float* q, const float* qn2,
                          const int64_t* cand, const float* cand_score, const float* qnorm, const float* bmax,
                          unsigned char* certified, int kc, int64_t nq, int k, int64_t id_base, int metric, int out_metric,
                          int64_t ntotal, int64_t* out_idx, float* out_dist, hipStream_t s, const float* seed_in, float* kth_out, float* floor_out) {
    if (nq == 0) return 0;
    if (kc > 256) return hb_fail("hb_index_search: too many candidates for the re-rank kernel");
    rerank_rows_kernel<<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s>>>(rows, rs, binit, d, q, qn2, cand, cand_score, qnorm, bmax,
                                                                          certified, kc, nq, k, id_base, metric, out_metric, ntotal, out_idx, out_dist,
                                                                          hb_rerank_seeds{seed_in, kth_out, floor_out});
    HB_HIP(hipGetLastError());
    return 0;
}

int hb_knn_f16_launch(const knn16_args& args, int grid, hipStream_t s) {
    // <4> compacts pools of up to 256 entries (k <= 64), <8> up to HB_POOL_MAX = 512 (its rare compaction path holds twice the registers)
    void (*fn)(knn16_args) = args.klw <= 256 ? knn_f16v2_kernel<4> : knn_f16v2_kernel<8>;
    if (args.klw > 512) return hb_fail("hb_index_search: candidate pools beyond 512 entries");
    if (hb_ensure_dyn_lds((const void*)fn, F2_LDS_TOTAL)) return -1;
    fn<<<dim3((unsigned)grid), dim3(HB_THREADS), F2_LDS_TOTAL, s>>>(args);
    HB_HIP(hipGetLastError());
    return 0;
}

// max over bank-row norms (all positive), kept in a device scalar for the certificate above
__global__ __launch_bounds__(256) void bnorm_max_kernel(const float* __restrict__ bnorm, int64_t n, float* __restrict__ bmax) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, bnorm[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(bmax), __float_as_int(m));   // positive floats order like ints
}

int hb_launch_bnorm_max(const float* bnorm, int64_t n, float* bmax, hipStream_t s) {
    if (n == 0) return 0;
    const int64_t blocks = std::min<int64_t>((n + 255) / 256, 1024);
    bnorm_max_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(bnorm, n, bmax);
    HB_HIP(hipGetLastError());
    return 0;
}

// out_idx[rows[i], :] = src_idx[i, :], out_dist likewise (results of the exact re-search of uncertified queries)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int64_t* __restrict__ rows, int64_t n, int k,
                                                           const int64_t* __restrict__ src_idx, const float* __restrict__ src_dist,
                                                           int64_t* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * k) return;
    const int64_t i = t / k;
    const int j = (int)(t % k);
    out_idx[rows[i] * k + j] = src_idx[t];
    out_dist[rows[i] * k + j] = src_dist[t];
}

int hb_launch_scatter_rows(const int64_t* rows, int64_t n, int k, const int64_t* src_idx, const float* src_dist,
                           int64_t* out_idx, float* out_dist, hipStream_t s) {
    if (n == 0) return 0;
    scatter_rows_kernel<<<dim3((unsigned)((n * k + 255) / 256)), dim3(256), 0, s>>>(rows, n, k, src_idx, src_dist, out_idx, out_dist);
    HB_HIP(hipGetLastError());
    return 0;
}
