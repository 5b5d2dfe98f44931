// The fp32 kNN kernel (default wherever it can run: tiles of a multiple of four k8 stages; hb_index_set_variant(ix, 4)
// forbids it): the same fused fp32-MFMA running top-k as hbird_knn.hip -- same tiles, same k-ascending
// fmaf chains, same lists / candidate pools, hence the same bits -- with the QUERY fragments loaded straight from global
// memory into registers instead of being staged through LDS.
//
// A wave's query fragment (32 queries x k8 = 1 KiB per stage) is read by that wave only, so the LDS stage buys no reuse: it
// costs an LDS-DMA write (the expensive side of the LDS) and a ds_read per stage.  Here the LDS ring holds bank fragments
// only (8 KiB per stage), the query fragment of stage s + 3 is requested during stage s by an inline-asm
// global_load_dwordx4 (invisible to hipcc's wait-count pass, counted by hand with the copies) into one of four register
// buffers; the stage loop is unrolled by four so that the buffers have static indices (g8 must be a multiple of 4: the
// launcher keeps other shapes on hbird_knn.hip).  10 M x 768, k = 30: 2377 -> 2288 ms on one box (0.895 -> 0.93 of the fp32
// MFMA peak); k = 90 (WIDE), 5 M rows: 1294 -> 1218 ms; 50,176 x 384 (COLD): 4.86 -> 4.62 ms.
#include "hbird_knn_dev.h"

#define BD_SLOT 8192                        // bank fragments of one k8 stage: 8 row tiles x 1 KiB
#define BD_RING 4
#define BD_BINIT (BD_RING * BD_SLOT)        // 2 x 1 KiB
#define BD_LISTS (BD_BINIT + 2048)
#define BD_SCRATCH (BD_LISTS + 2 * HB_QT * HB_KL * 4)
#define BD_CLWORDS (BD_SCRATCH + 8192)       // landing zone of the cluster progress poll
// Stages between two looks at the other cluster members.  Measured at 10 M x 768, 2 x 4, lag 16 (kernel ms over no clusters /
// FETCH_SIZE x 2): every 16 stages +1.3 %, 32: +0.9 % / 1.93 TB, 64: +0.8 %, 128: +0.3 % / 1.97 TB, 256: +0.4 % / 2.12 TB.
#define BD_CL_PERIOD 128
#define BD_LDS_TOTAL (BD_CLWORDS + 64)
#define BD_QF BD_LDS_TOTAL                    // small-search instantiation: quota floors, 2 KiB per wave
#define BD_LDS_TOTAL_COLD (BD_QF + 8 * 2048)

#define BD_MFMA(T, FR, B, S) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(FR[(T) & 3][S], B[S], acc[T], 0, 0, 0);
// the two wave classes have different numbers of requests in flight: ONE statement with the branch inside, so that the
// "+v" register is the same on both paths (hbird_knn_f16.hip: two statements in an if / else made hipcc copy it early)
#define BD_WAIT(N_ISSUER, N_OTHER, B)                                                                                   \
    asm volatile("s_cmp_lt_u32 %1, 4\n\ts_cbranch_scc1 .Lbdw_%=\n\ts_waitcnt vmcnt(" #N_OTHER ")\n\ts_branch .Lbdd_%=\n"  \
                 ".Lbdw_%=:\n\ts_waitcnt vmcnt(" #N_ISSUER ")\n.Lbdd_%=:"                                               \
                 : "+v"(B) : "s"(w) : "memory", "scc");
// (Rounds 2 / 3 carried timing-only ablation builds of this loop -- no bank copies / no query-fragment loads / no epilogue / no
// fragment reads -- and placement experiments -- every wave copying its own row tile, static wave priorities, the requests in the Y
// half, floors requested at a tile's start: their numbers are in profiles/LABBOOK.md section 4 and profiles/r02, the switches are gone.)
#define BD_BLOAD(B) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(B) : "v"(lane_off), "s"(qfp) : "memory");
// one 1 KiB LDS-DMA piece in the saddr form: wave-uniform 64-bit base + 32-bit lane offset -> LDS (wave-uniform address in M0 + 16 * lane);
// the builtin takes a per-lane flat address (a 64-bit VALU add per piece).  M0 is clobbered on purpose (hbird_knn_f16.hip: F2_DMA)
#define BD_DMA(SRC, LDS_ADDR, VOFF)                                                                                      \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(VOFF), "s"(SRC), "s"(LDS_ADDR) : "memory", "m0");
#define BD_RD(DST, SRC) DST = SRC;

// WIDE: k > HB_KL, candidate pools in global memory; CL: member of an L2-sharing cluster (strided segments on a common
// clock, soft sync from wave 0); COLD: small search (radix-select cold start, scan epilogue, per-tile floors) -- all as in
// hbird_knn.hip
// (Round 5's OL instantiations -- all phases of a pool search in ONE launch behind a grid barrier -- measured 2-10 % slower than a launch per
// phase and were removed in round 6: profiles/r05/one_launch_*.txt, git history.)
template <bool WIDE, bool CL, bool COLD = false>
__global__ __launch_bounds__(HB_THREADS, 2) void knn_fused_bd_kernel(knn_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    float* lst_s = reinterpret_cast<float*>(smem + BD_LISTS);
    unsigned* lst_i = reinterpret_cast<unsigned*>(smem + BD_LISTS + HB_QT * HB_KL * 4);
    float* sc = reinterpret_cast<float*>(smem + BD_SCRATCH) + w * 256;
    unsigned* qf = reinterpret_cast<unsigned*>(smem + BD_QF) + w * 512;   // COLD only
    int* pcnt = reinterpret_cast<int*>(smem + BD_LISTS);   // WIDE: pool fill counts in the (otherwise unused) list area
    const int g8 = a.g8, k = a.k;
    const int myq = w * 32 + (lane & 31);
    const unsigned lane_off = (unsigned)lane * 16u;
    const unsigned lane_off_hi = lane_off + 4u * (unsigned)g8 * 1024u;      // the same lane's piece of row tile w + 4
    const unsigned lds_0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    cl_sync cs;
    if constexpr (CL) cs = cl_init(a.wg_member, a.prog, a.cl, a.lag, blockIdx.x, w == 0, smem + BD_CLWORDS);

    wg_stamp<knn_args>(0);
    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_end[blockIdx.x];   // this launch's share of the block's segments (phases: hb_launch_knn)
    // "everything before my first segment is done" (a member without any work: everything)
    if constexpr (CL) { if (w == 0) cl_publish(cs, seg_begin < seg_end ? a.segs[seg_begin].tile0 * g8 : 0x7FFFFFFF, lane); }
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = a.segs[si];
        const int bstride = CL ? seg.stride : 1, clock0 = CL ? seg.tile0 * g8 : 0;
        // (arguments of the segment / tile boundaries come from the kernarg segment where they are used: HB_KARG)
        float thr;
        if constexpr (WIDE) {
            thr = pool_begin(knn_args_pool_view{HB_KARG(knn_args, state_cnt), HB_KARG(knn_args, state_thr)}, seg.slot, seg.first, pcnt, myq, lane);
        } else {
            const float* wl_s = HB_KARG(knn_args, state_s) + (size_t)seg.slot * HB_QT * HB_KL;
            const unsigned* wl_i = HB_KARG(knn_args, state_i) + (size_t)seg.slot * HB_QT * HB_KL;
            for (int e = lane; e < 1024; e += 64) {
                lst_s[w * 1024 + e] = seg.first ? -INFINITY : wl_s[w * 1024 + e];
                lst_i[w * 1024 + e] = seg.first ? HB_ID_NONE : wl_i[w * 1024 + e];
            }
            thr = lst_s[myq * HB_KL + (k - 1)];
        }
        thr = fmaxf(thr, floor_load(HB_KARG(knn_args, gthr), seg.q_tile * HB_QT + myq));
        [[maybe_unused]] bool bulk = seg.tile0 < 16;   // WIDE && COLD: loose floors at the start of a search (pool_epilogue_scan)
        // COLD lists: which tiles exchange floors with the query tile's other slots (small_floor_*).  Every tile while floors are loose -- the
        // first 64 tiles of the workgroup, the first 8 of a new slot --, then every 16th: the exchange costs a tile about 1 % (waves 4-7
        // drain their run-ahead loads to read it) and the floors of a slot that has seen 16,000 rows move slowly (tools/exp_size_sweep.py).
        auto floor_tile = [](int clock, bool first, int ti) { return clock < 64 || (first && ti < 8) || (clock & 15) == 0; };
        [[maybe_unused]] bool fx = floor_tile(seg.tile0, seg.first, 0);
        const char* qsrc = reinterpret_cast<const char*>(a.q_tiles + ((size_t)(seg.q_tile * 8 + w) * g8) * HB_BLK);
        const int total = seg.n_tiles * g8;
        f32x16 acc[8];
        f32x4 fa[4], fy[4];   // X-half / Y-half bank fragments
        f32x4 bq[4];          // query fragments of four stages

        // The fetch position as running wave-uniform pointers (round 4, as in hbird_knn_f16.hip: the per-stage address arithmetic -- a
        // 64-bit multiply chain and per-lane 64-bit adds for every copy -- was the largest part of the stage loop's scalar skeleton):
        // qfp = this wave's query fragment, bf = bank row tile w (row tile w + 4 through a second lane offset), bi = the tile's row-init
        // values, all of the NEXT stage to request.  A stage advances qfp and bf by 1 KiB; at a tile's end qfp returns to the wave's
        // first fragment and bf jumps from the end of row tile w to row tile w of the segment's next tile.
        // waves 0-3 copy the bank row tiles w and w + 4 of a stage (their SIMD partners 4-7 issue no copies, hbird_knn.hip)
        // (what only a tile's end needs -- the wrap distances, the row-init base -- is recomputed / re-read there: fewer live scalars)
        const char* qfp = qsrc;
        const char* bf = reinterpret_cast<const char*>(a.bank_tiles) + ((size_t)seg.b_tile0 * 8 + w) * g8 * 1024;
        int bt = seg.b_tile0, ks = 0;          // tile / stage being computed
        int fbt = seg.b_tile0, fks = 0;        // tile / stage being fetched
        int slot_c = 0, left = total;
        unsigned slot_f = 0, fpar = 0;         // ring slot being filled (byte offset), row-init buffer being filled (byte offset)
        int cpar = 0;                          // row-init double buffer: parity of the tile being computed
        auto issue_a = [&]() {
            if (w < 4) {
                BD_DMA(bf, lds_0 + slot_f + (unsigned)w * 1024u, lane_off)
                BD_DMA(bf, lds_0 + slot_f + (unsigned)(w + 4) * 1024u, lane_off_hi)
            }
        };
        auto advance_fetch = [&]() {
            if (fks == 0 && w == 0) BD_DMA(reinterpret_cast<const char*>(HB_KARG(knn_args, binit) + (size_t)fbt * HB_BT), lds_0 + BD_BINIT + fpar, lane_off)
            if (--left > 0) {
                qfp += 1024; bf += 1024;
                if (++fks == g8) {
                    fks = 0; fbt += bstride; fpar ^= 1024u;
                    qfp -= (size_t)g8 * 1024;
                    bf += ((long long)bstride * 8 - 1) * g8 * 1024;
                }
            }
            slot_f = (slot_f + BD_SLOT) & (BD_RING * BD_SLOT - 1);
        };
        // vmcnt by hand.  Per stage a wave requests, in this order: (waves 0-3) two bank pieces, then its query fragment
        // (wave 0, first stage of a tile: the row-init values behind it, which only makes a wait stricter).  The requests
        // of stage j are issued during stage j - 3; at the top of stage s those of stage s + 1 must have landed, those of
        // stage s + 2 may be in flight: "all but the newest 3" (waves 4-7: 1).  Past the last stage the fetch position stays
        // put (same requests again, results unused), so the count never changes.
        issue_a(); BD_BLOAD(bq[0]) advance_fetch();
        issue_a(); BD_BLOAD(bq[1]) advance_fetch();
        issue_a(); BD_BLOAD(bq[2]) advance_fetch();
        BD_WAIT(6, 2, bq[0])     // stage 0 landed; stages 1-2 in flight
        __syncthreads();
        {
            const f32x4* A = reinterpret_cast<const f32x4*>(smem);
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = A[t * 64 + lane];
        }
        // the requests are issued in the X half of the stage (in the Y half: no gain, 5 M x 768 1149 / 1147 / 1161 ms)
#define BD_REQ_X1 issue_a();
#define BD_REQ_X2(U) BD_BLOAD(bq[((U) + 3) & 3])
#define BD_REQ_X3 advance_fetch();
#define BD_REQ_Y1(U)
#define BD_REQ_Y2
// small searches: the floors are requested in the tile's LAST four stages (at its start they were one tile staler: 50,176 x 384
// 4.62 -> 4.53 ms, 200 k x 384 15.12 -> 14.98).  Four stages of counted waits cover the request for
// waves 0-3, waves 4-7 wait for it explicitly before they read.
#define BD_FLOOR_KS (g8 - 4)
#define BD_STAGE(U)                                                                                                     \
        {                                                                                                               \
            BD_WAIT(3, 1, bq[((U) + 1) & 3])   /* stage st + 1 has landed for me (st + 2 in flight) ... */              \
            __builtin_amdgcn_s_barrier();      /* ... and for everyone; the slot of stage st - 1 is free for st + 3 */   \
            /* small searches: this tile's floors, requested HERE so that they are older than the stage's own requests */ \
            if constexpr (COLD && !WIDE && (U) == 0) { if (ks == BD_FLOOR_KS && fx) small_floor_request(HB_KARG(knn_args, qfl), HB_KARG(knn_args, gthr), seg, w, lane, qf, sc); } \
            int slot_n = slot_c + 1; if (slot_n == BD_RING) slot_n = 0;                                                 \
            const f32x4* Ac = reinterpret_cast<const f32x4*>(smem + slot_c * BD_SLOT) + lane;                           \
            const f32x4* An = reinterpret_cast<const f32x4*>(smem + slot_n * BD_SLOT) + lane;                           \
            /* X half: tiles 0-3, k-steps 0-3; fillers: the four Y fragments, the copies, the query fragment of st + 3 */ \
            KN_FENCE BD_MFMA(0, fa, bq[U], 0) KN_FENCE BD_RD(fy[0], Ac[4 * 64])                                              \
            KN_FENCE BD_MFMA(1, fa, bq[U], 0) KN_FENCE BD_RD(fy[1], Ac[5 * 64])                                              \
            KN_FENCE BD_MFMA(2, fa, bq[U], 0) KN_FENCE BD_RD(fy[2], Ac[6 * 64])                                              \
            KN_FENCE BD_MFMA(3, fa, bq[U], 0) KN_FENCE BD_RD(fy[3], Ac[7 * 64])                                              \
            KN_FENCE BD_MFMA(0, fa, bq[U], 1) BD_MFMA(1, fa, bq[U], 1) KN_FENCE                                         \
            /* cluster soft sync (acts every 4th stage; the clock is a multiple of 4 at U == 0), AHEAD of the stage's requests */ \
            if constexpr (CL && (U) == 0) { if (w == 0) cl_tick<BD_CL_PERIOD>(cs, clock0 + st, lane); }                 \
            BD_REQ_X1                                                                                                   \
            KN_FENCE BD_MFMA(2, fa, bq[U], 1) BD_MFMA(3, fa, bq[U], 1) BD_MFMA(0, fa, bq[U], 2) BD_MFMA(1, fa, bq[U], 2) KN_FENCE \
            BD_REQ_X2(U)                                                                                                \
            KN_FENCE BD_MFMA(2, fa, bq[U], 2) BD_MFMA(3, fa, bq[U], 2) KN_FENCE                                         \
            BD_REQ_X3                                                                                                   \
            KN_FENCE BD_MFMA(0, fa, bq[U], 3) BD_MFMA(1, fa, bq[U], 3) BD_MFMA(2, fa, bq[U], 3) BD_MFMA(3, fa, bq[U], 3) KN_FENCE \
            /* Y half: tiles 4-7; fillers: the X fragments of stage st + 1 */                                           \
            KN_FENCE BD_MFMA(4, fy, bq[U], 0) KN_FENCE BD_RD(fa[0], An[0 * 64])                                              \
            KN_FENCE BD_MFMA(5, fy, bq[U], 0) KN_FENCE BD_RD(fa[1], An[1 * 64])                                              \
            KN_FENCE BD_MFMA(6, fy, bq[U], 0) KN_FENCE BD_RD(fa[2], An[2 * 64])                                              \
            KN_FENCE BD_MFMA(7, fy, bq[U], 0) KN_FENCE BD_RD(fa[3], An[3 * 64])                                              \
            KN_FENCE                                                                                                    \
            BD_MFMA(4, fy, bq[U], 1) BD_MFMA(5, fy, bq[U], 1) KN_FENCE BD_REQ_Y1(U) KN_FENCE BD_MFMA(6, fy, bq[U], 1) BD_MFMA(7, fy, bq[U], 1) \
            BD_MFMA(4, fy, bq[U], 2) BD_MFMA(5, fy, bq[U], 2) KN_FENCE BD_REQ_Y2 KN_FENCE BD_MFMA(6, fy, bq[U], 2) BD_MFMA(7, fy, bq[U], 2)    \
            BD_MFMA(4, fy, bq[U], 3) BD_MFMA(5, fy, bq[U], 3) BD_MFMA(6, fy, bq[U], 3) BD_MFMA(7, fy, bq[U], 3)         \
            KN_FENCE                                                                                                    \
            slot_c = slot_n;                                                                                            \
        }
        for (int st = 0; st < total; st += 4) {
            if (ks == 0) {   // accumulators start from the bank rows' init values (landed with the tile's first stage)
                // the init values of this tile were published by an earlier barrier (they ride with the tile's first stage)
                const f32x4* bi = reinterpret_cast<const f32x4*>(smem + BD_BINIT + cpar * 1024);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = bi[8 * t + 2 * g + h];
                        acc[t][4 * g + 0] = v[0]; acc[t][4 * g + 1] = v[1];
                        acc[t][4 * g + 2] = v[2]; acc[t][4 * g + 3] = v[3];
                    }
            }
            BD_STAGE(0) BD_STAGE(1) BD_STAGE(2) BD_STAGE(3)
            ks += 4;
            if (ks == g8) {
                if constexpr (WIDE) {
                    // the slot's pool pointers are derived HERE from one laundered scalar (hbird_knn.hip: kept live through
                    // the stage loop they push the loop's own pointers into spilled SGPRs, reloaded in every stage)
                    const hb_seg* sp_ = HB_KARG(knn_args, segs) + si;     // boundary-only fields: read again where they are used
                    asm volatile("" : "+s"(sp_));
                    const int slot_ = sp_->slot;
                    const int klw = HB_KARG(knn_args, klw);
                    float* ps = HB_KARG(knn_args, state_s) + (size_t)slot_ * HB_QT * klw;
                    unsigned* pi = HB_KARG(knn_args, state_i) + (size_t)slot_ * HB_QT * klw;
                    if constexpr (COLD) {
                        // small searches: a slot sees few rows, so its first tiles and its appends are a visible share -- cold start,
                        // then the fp16 candidate kernel's epilogue (register queue, one drain per tile; hbird_knn_dev.h)
                        if (sp_->first && bt == sp_->b_tile0 && cold_start_needed(thr)) thr = fmaxf(thr, cold_start_threshold(acc, k));
                        pool_epilogue_scan<HB_POOL_MAX / 64>(acc, thr, ps, pi, sc, w * 32, lane, k, (unsigned)bt, klw, pcnt, bulk);
                    } else tile_epilogue<true, true>(acc, thr, ps, pi, sc, w * 32, lane, k, (unsigned)bt, klw, pcnt);
                } else if constexpr (COLD) {
                    if (seg.first && bt == seg.b_tile0 && cold_start_needed(thr)) thr = fmaxf(thr, cold_start_threshold(acc, k));
                    // the floors requested at the tile's start (waves 4-7: behind their query fragments; waves 0-3 have passed
                    // counted waits that cover them)
                    if (fx) {
                        asm volatile("s_cmp_lt_u32 %0, 4\n\ts_cbranch_scc1 .Lbfl_%=\n\ts_waitcnt vmcnt(0)\n.Lbfl_%=:" :: "s"(w) : "memory", "scc");
                        thr = fmaxf(thr, small_floor_read(seg, qf, sc, lane));
                    }
                    list_epilogue_scan(acc, thr, lst_s, lst_i, w * 32, lane, k, (unsigned)bt);
                    if (fx) small_floor_publish(HB_KARG(knn_args, qfl), HB_KARG(knn_args, gthr), seg, lst_s, myq, k, thr, lane);
                    {   // the next tile's turn (boundary-only fields: read again, not kept in scalars through the loop)
                        const hb_seg* sn_ = HB_KARG(knn_args, segs) + si;
                        asm volatile("" : "+s"(sn_));
                        const int ti = bt + bstride - sn_->b_tile0;
                        fx = floor_tile(sn_->tile0 + ti, sn_->first != 0, ti);
                    }
                } else tile_epilogue<true, false>(acc, thr, lst_s, lst_i, sc, w * 32, lane, k, (unsigned)bt);
                ks = 0;
                bt += bstride; cpar ^= 1;
            }
        }
#undef BD_STAGE
        // the run-ahead requests still target the query-fragment registers: drain them while those registers are live
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) :: "memory");
        const hb_seg* se_ = HB_KARG(knn_args, segs) + si;     // boundary-only fields: read again (not kept in scalars through the loop)
        asm volatile("" : "+s"(se_));
        const int e_next = se_->next_tile0, e_slot = se_->slot, e_qt = se_->q_tile;
        if constexpr (CL) { if (w == 0) cl_publish(cs, e_next == 0x7FFFFFFF ? 0x7FFFFFFF : e_next * g8, lane); }   // covers idle units
        if constexpr (WIDE) pool_end(knn_args_pool_view{HB_KARG(knn_args, state_cnt), HB_KARG(knn_args, state_thr)}, e_slot, pcnt, thr, myq, lane);
        else {
            float* wl_s = HB_KARG(knn_args, state_s) + (size_t)e_slot * HB_QT * HB_KL;
            unsigned* wl_i = HB_KARG(knn_args, state_i) + (size_t)e_slot * HB_QT * HB_KL;
            for (int e = lane; e < 1024; e += 64) { wl_s[w * 1024 + e] = lst_s[w * 1024 + e]; wl_i[w * 1024 + e] = lst_i[w * 1024 + e]; }
        }
        if (lane < 32) floor_publish(HB_KARG(knn_args, gthr), e_qt * HB_QT + myq, thr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the ring and the lists are reused by the next segment
    }
    if constexpr (CL) cl_finish(cs, a.cl_stats, w == 0, lane);
    wg_stamp<knn_args>(1);
}

hb_knn_fn hb_knn_bd_kernel(bool wide, bool clustered, bool small) {
    if (small && !clustered) return wide ? knn_fused_bd_kernel<true, false, true> : knn_fused_bd_kernel<false, false, true>;
    if (clustered) return wide ? knn_fused_bd_kernel<true, true> : knn_fused_bd_kernel<false, true>;
    return wide ? knn_fused_bd_kernel<true, false> : knn_fused_bd_kernel<false, false>;
}
int hb_knn_bd_lds_bytes(bool small_lists) { return small_lists ? BD_LDS_TOTAL_COLD : BD_LDS_TOTAL; }
