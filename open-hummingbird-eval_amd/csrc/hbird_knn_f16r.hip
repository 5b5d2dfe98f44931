// fp16 candidate kernel, register-staged variant (hb_index_set_variant bit 1): 4 waves, one per SIMD.
//
// Why: the LDS is what bounds the 8-wave kernel (hbird_knn_f16.hip) -- an LDS-DMA byte occupies the LDS port about four
// times as long as a ds_read / ds_write byte and the two streams serialise (tools/ubench/lds_dma_contention.hip).  Here
//   * wave w owns queries [64w, 64w + 64) against all 256 bank rows: 16 accumulator tiles in AGPRs, 10 fragment reads
//     per 16 MFMAs (0.63 KiB of LDS reads per MFMA instead of 1.13);
//   * the copies go global -> VGPR (global_load_dwordx4) -> LDS (ds_write_b128): four register sets of 8 x 16 B per
//     lane keep FOUR k32 stages (128 KiB per CU) in flight, the LDS holds only the stage being read and the next one.
//     The candidate kernels are bound by Little's law -- loaded memory latency (~3 us) over the bytes in flight -- and
//     the LDS-DMA ring of the 8-wave kernel cannot hold more than three stages;
//   * per k16 group the 16 MFMAs are interleaved with the fillers (fragment reads of the next group; in group 0 the
//     ds_writes of the next stage, in group 1 the global loads of the stage five ahead), pinned with sched_barrier.
// The compiler tracks the VGPR loads, so vmcnt needs no hand counting here.
// Needs Dp16 % 128 == 0 (k32 stages per tile divisible by the four register sets, which are compile-time constants in
// the unrolled loop) and pools of at most 256 entries; other searches use the 8-wave kernel.
#include "hbird_knn_dev.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define R_THREADS 256
#define R_EMAX 4                               // pools of at most 256 entries (k' <= 128): fewer registers in the compaction
#define R_HALF 16384                          // 16 KiB of bank fragments of a k32 stage, then 16 KiB of query fragments
#define R_SLOT (2 * R_HALF)
#define R_BINIT (2 * R_SLOT)                  // two LDS slots: the stage being read and the next one
#define R_SCRATCH (R_BINIT + 2048)
#define R_PCNT (R_SCRATCH + 4096)
#define R_LDS_TOTAL (R_PCNT + 1024)

template <int N> struct ic { static constexpr int value = N; };

// One accumulator tile out of the AGPRs, element by element and only when asked: without this the register allocator
// moves all 256 accumulators to VGPRs at the loop exit (their VALU uses in the epilogue) and spills most of them.
__device__ __forceinline__ f32x16 acc_tile_to_vgprs(const f32x16& acc) {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float x;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i]));
        v[i] = x;
    }
    return v;
}

__global__ __launch_bounds__(R_THREADS) void knn_f16r_kernel(knn16_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    float* sc = reinterpret_cast<float*>(smem + R_SCRATCH) + w * 256;
    int* pcnt = reinterpret_cast<int*>(smem + R_PCNT);
    const int g16 = a.g16, k = a.k, klw = a.klw;
    const int NS = g16 / 2;   // k32 stages per bank tile, a multiple of 4
    const int q0 = w * 64 + (lane & 31), q1 = q0 + 32;
    const unsigned lane_off = (unsigned)lane * 16u;

    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_off[blockIdx.x + 1];
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = a.segs[si];
        float* wl_s = a.state_s + (size_t)seg.slot * HB_QT * klw;
        unsigned* wl_i = a.state_i + (size_t)seg.slot * HB_QT * klw;
        const knn_args_pool_view pv{a.state_cnt, a.state_thr};
        float thr0 = pool_begin(pv, seg.slot, seg.first, pcnt, q0, lane);
        float thr1 = pool_begin(pv, seg.slot, seg.first, pcnt, q1, lane);
        thr0 = fmaxf(thr0, floor_load(a.gthr, seg.q_tile * HB_QT + q0));
        thr1 = fmaxf(thr1, floor_load(a.gthr, seg.q_tile * HB_QT + q1));
        const int total = seg.n_tiles * NS;
        f32x16 acc0[8], acc1[8];   // query column blocks 2w and 2w+1
        u32x4 R[4][8];             // four stages of copies in flight: 8 x 16 B per lane each
        u32x4 rb;                  // wave 0: the next tile's row-init values

        // copy i of a stage for wave w: row tile 2w + (i >> 2), k16 group (i >> 1) & 1, bank (i even) or query (i odd)
        const char* bank_w = reinterpret_cast<const char*>(a.bank16) + (size_t)(2 * w) * g16 * 1024;
        const char* query_w = reinterpret_cast<const char*>(a.q16) + (size_t)(seg.q_tile * 8 + 2 * w) * g16 * 1024;
        auto src_of = [&](int i, int bt, int ks) -> const u32x4* {
#if defined(F16_ABL) && (F16_ABL & 2)
            bt &= 3;   // timing only: 4 bank tiles, L2-resident
#endif
            const size_t off = ((size_t)(i >> 2) * g16 + ks * 2 + ((i >> 1) & 1)) * 1024 + lane_off;
#if defined(F16_ABL) && (F16_ABL & 8)
            const char* qw = reinterpret_cast<const char*>(a.q16) + (size_t)(2 * w) * g16 * 1024;   // timing only: one query tile for all
#else
            const char* qw = query_w;
#endif
            return reinterpret_cast<const u32x4*>((i & 1) ? qw + off : bank_w + (size_t)bt * 8 * g16 * 1024 + off);
        };
        auto dst_of = [&](int i, int slot) -> u32x4* {
            return reinterpret_cast<u32x4*>(smem + slot * R_SLOT + (i & 1) * R_HALF + ((2 * w + (i >> 2)) * 2 + ((i >> 1) & 1)) * 1024 + lane_off);
        };

        int bt = seg.b_tile0;                  // tile being computed
        int fbt = seg.b_tile0, fks = 0;        // next stage to fetch (stays on the last stage past the end)
        int left = total;
        auto advance_fetch = [&]() { if (--left > 0) { if (++fks == NS) { fks = 0; ++fbt; } } };

        // prologue: stages 0-3 into the four register sets, stage 0 on to LDS slot 0, then stage 4 into set 0
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int i = 0; i < 8; ++i) R[st][i] = *src_of(i, fbt, fks);
            advance_fetch();
        }
        if (w == 0) *reinterpret_cast<u32x4*>(smem + R_BINIT + (bt & 1) * 1024 + lane_off) =
                        *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.binit + (size_t)bt * HB_BT) + lane_off);
#pragma unroll
        for (int i = 0; i < 8; ++i) *dst_of(i, 0) = R[0][i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) R[0][i] = *src_of(i, fbt, fks);
        advance_fetch();

        // Fragments: a bank fragment feeds exactly two MFMAs (the wave's two query blocks), so a ring of four -- read three
        // MFMA pairs ahead -- hides the LDS latency; the two query fragments of a k16 group are read one group ahead.
        f16x8 ar[4], b0[2], b1[2];
        // One k32 stage whose copies sit in register set SET = stage % 4, in LDS slot SLOT = stage & 1.
        // 16 MFMA pairs p = 8 g + t (k16 group g, bank row tile t); fillers after pair p: the bank fragment of pair p + 3,
        // the query fragments of the next group (p = 0, 8), and one copy: the next stage on its way from the registers
        // to LDS (p < 8) or the global load of the stage five ahead (p >= 8).  The barrier after pair 7 says: this stage's
        // group 0 ... everything of this stage has been read into registers or is being read from LDS only by... see below.
        auto stage = [&](auto set_c, auto slot_c) {
            constexpr int SET = decltype(set_c)::value, SLOT = decltype(slot_c)::value;
            constexpr int NXT = (SET + 1) % 4, NSLOT = SLOT ^ 1;
            const f16x8* Ac = reinterpret_cast<const f16x8*>(smem + SLOT * R_SLOT) + lane;
            const f16x8* An = reinterpret_cast<const f16x8*>(smem + NSLOT * R_SLOT) + lane;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int g = p >> 3, t = p & 7;
                KN_FENCE
                acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[p & 3], b0[g], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[p & 3], b1[g], acc1[t], 0, 0, 0);
                KN_FENCE
                const int pn = p + 3;
                if (pn < 16) ar[pn & 3] = Ac[((pn & 7) * 2 + (pn >> 3)) * 64];
                else ar[pn & 3] = An[(((pn - 16) & 7) * 2) * 64];          // p >= 13: after the barrier, next stage
                if (p == 0) { b0[1] = Ac[R_HALF / 16 + ((2 * w) * 2 + 1) * 64]; b1[1] = Ac[R_HALF / 16 + ((2 * w + 1) * 2 + 1) * 64]; }
                if (p == 8) { b0[0] = An[R_HALF / 16 + ((2 * w) * 2) * 64]; b1[0] = An[R_HALF / 16 + ((2 * w + 1) * 2) * 64]; }
                if (p < 8) *dst_of(p, NSLOT) = R[NXT][p];
                else R[NXT][p - 8] = *src_of(p - 8, fbt, fks);
                if (p == 7) {
                    KN_FENCE
                    __syncthreads();   // the next stage is in LDS for everyone; its slot's previous content was read before
                }
            }
            advance_fetch();
            KN_FENCE
        };
        for (int tl = 0; tl < seg.n_tiles; ++tl, ++bt) {
            {   // first fragments of the tile (read again rather than kept live across the epilogue: fewer registers)
                const f16x8* A = reinterpret_cast<const f16x8*>(smem) + lane;   // a tile starts on an even stage: slot 0
#pragma unroll
                for (int t = 0; t < 3; ++t) ar[t] = A[(t * 2) * 64];
                b0[0] = A[R_HALF / 16 + ((2 * w) * 2) * 64];
                b1[0] = A[R_HALF / 16 + ((2 * w + 1) * 2) * 64];
            }
            {   // accumulators start from the bank rows' init values
                const f32x4* bi = reinterpret_cast<const f32x4*>(smem + R_BINIT + (bt & 1) * 1024);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = bi[8 * t + 2 * g + h];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { acc0[t][4 * g + r] = v[r]; acc1[t][4 * g + r] = v[r]; }
                    }
            }
            // wave 0 fetches the next tile's row-init values now and publishes them before the tile's last barrier
            const int nbt = tl + 1 < seg.n_tiles ? bt + 1 : bt;
            if (w == 0) rb = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.binit + (size_t)nbt * HB_BT) + lane_off);
#pragma nounroll
            for (int ks = 0; ks < NS; ks += 4) {
                stage(ic<0>{}, ic<0>{});
                stage(ic<1>{}, ic<1>{});
                stage(ic<2>{}, ic<0>{});
                if (w == 0 && ks + 4 == NS) *reinterpret_cast<u32x4*>(smem + R_BINIT + (nbt & 1) * 1024 + lane_off) = rb;
                stage(ic<3>{}, ic<1>{});
            }
#if defined(F16_ABL) && (F16_ABL & 1)
#pragma unroll
            for (int t = 0; t < 8; ++t) asm volatile("" :: "a"(acc0[t]), "a"(acc1[t]));   // timing only: no epilogue
#else
            // per accumulator tile (16 values out of the AGPRs at a time), ascending bank rows for each query block
#pragma unroll
            for (int t = 0; t < 8; ++t) { KN_FENCE tile_epilogue_pool_one<R_EMAX>(acc_tile_to_vgprs(acc0[t]), t, thr0, wl_s, wl_i, sc, w * 64, lane, k, (unsigned)bt, klw, pcnt); }
#pragma unroll
            for (int t = 0; t < 8; ++t) { KN_FENCE tile_epilogue_pool_one<R_EMAX>(acc_tile_to_vgprs(acc1[t]), t, thr1, wl_s, wl_i, sc, w * 64 + 32, lane, k, (unsigned)bt, klw, pcnt); }
            KN_FENCE
#endif
        }
        pool_end(pv, seg.slot, pcnt, thr0, q0, lane);
        pool_end(pv, seg.slot, pcnt, thr1, q1, lane);
        if (lane < 32) { floor_publish(a.gthr, seg.q_tile * HB_QT + q0, thr0); floor_publish(a.gthr, seg.q_tile * HB_QT + q1, thr1); }
        __syncthreads();   // the LDS slots and the fill counts are reused by the next segment
    }
}

int hb_knn_f16r_launch(const knn16_args& args, int grid, hipStream_t s) {
    static bool attr = false;
    if (!attr) { HB_HIP(hipFuncSetAttribute((const void*)knn_f16r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_TOTAL)); attr = true; }
    knn_f16r_kernel<<<dim3((unsigned)grid), dim3(R_THREADS), R_LDS_TOTAL, s>>>(args);
    HB_HIP(hipGetLastError());
    return 0;
}
