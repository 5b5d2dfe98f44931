// K4, 4-wave variant: one wave per SIMD, 256 accumulator registers per lane (AGPR file).
//
// Same pair tile (256 queries x 256 bank rows), same LDS ring / work list / partial-list format as
// knn_fused_kernel, but the workgroup has 4 waves and wave w owns query columns [64w, 64w+64) x all 256 bank
// rows: acc[8 row tiles][2 query tiles] = 256 registers.  Per k8 stage a wave issues 64 MFMAs against 10
// fragment reads and 4 LDS-DMA copies (the 8-wave kernel: 32 MFMAs against 9 reads and 2 copies), i.e. about
// half the non-matrix instructions per MFMA; with only one wave per SIMD nothing else competes for the issue port.
#include "hbird_knn_dev.h"

#define W4_THREADS 256

// dump 8 registers (one half tile of one query tile) of accumulator tile (T, NB) to the wave's scratch
#define W4_DUMP_CASE(T, NB, H)                                                                         \
    case (4 * (T) + 2 * (NB) + (H)):                                                                   \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) sc[r * 64 + lane] = acc[2 * (T) + (NB)][8 * (H) + r]; \
        break;
#define W4_DUMP_TILE(T) W4_DUMP_CASE(T, 0, 0) W4_DUMP_CASE(T, 0, 1) W4_DUMP_CASE(T, 1, 0) W4_DUMP_CASE(T, 1, 1)

__device__ __forceinline__ void w4_epilogue(f32x16 (&acc)[16], float (&thr)[2], float* lst_s, unsigned* lst_i, float* sc,
                                            int w, int lane, int k, unsigned bt, int klw) {
    // phase 1: which (row tile, query tile, half) has a score above its query's threshold
    unsigned hmask = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            bool any0 = false, any1 = false;
#pragma unroll
            for (int r = 0; r < 8; ++r) { any0 |= acc[2 * t + nb][r] > thr[nb]; any1 |= acc[2 * t + nb][8 + r] > thr[nb]; }
            if (__ballot(any0) != 0ull) hmask |= 1u << (4 * t + 2 * nb);
            if (__ballot(any1) != 0ull) hmask |= 1u << (4 * t + 2 * nb + 1);
        }
    // phase 2 (rare): bits are visited in ascending order = ascending bank row for each query
    while (hmask) {
        const int bit = __builtin_ctz(hmask);
        hmask &= hmask - 1;
        switch (bit) {
            W4_DUMP_TILE(0) W4_DUMP_TILE(1) W4_DUMP_TILE(2) W4_DUMP_TILE(3)
            W4_DUMP_TILE(4) W4_DUMP_TILE(5) W4_DUMP_TILE(6) W4_DUMP_TILE(7)
        }
        const int t = bit >> 2, nb = (bit >> 1) & 1, hf = bit & 1;
        const unsigned row_base = bt * HB_BT + t * 32 + hf * 16;
        const int qbase = w * 64 + nb * 32;
        float th = nb ? thr[1] : thr[0];
        for (int gg = 0; gg < 2; ++gg)
            for (int hh = 0; hh < 2; ++hh)
                for (int j = 0; j < 4; ++j) {
                    const float v = sc[(gg * 4 + j) * 64 + lane];
                    unsigned long long m = __ballot(v > th);
                    m &= hh ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
                        const int n = l & 31;
                        list_insert(lst_s, lst_i, qbase + n, k, s, row_base + gg * 8 + hh * 4 + j, lane);
                        const float kth = lst_s[(qbase + n) * HB_KL + (k - 1)];
                        if ((lane & 31) == n) th = kth;
                    }
                }
        if (nb) thr[1] = th; else thr[0] = th;
    }
}

#define W4_MFMA(T, NB, FR, FB, S) \
    acc[2 * (T) + (NB)] = __builtin_amdgcn_mfma_f32_32x32x2f32(FR[(T) & 3][S], FB[NB][S], acc[2 * (T) + (NB)], 0, 0, 0);
#define W4_MFMA2(T, FR, FB, S) W4_MFMA(T, 0, FR, FB, S) W4_MFMA(T, 1, FR, FB, S)

__global__ __launch_bounds__(W4_THREADS) void knn_fused_w4_kernel(knn_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    float* lst_s = reinterpret_cast<float*>(smem + KN_LISTS);
    unsigned* lst_i = reinterpret_cast<unsigned*>(smem + KN_LISTS + HB_QT * HB_KL * 4);
    float* sc = reinterpret_cast<float*>(smem + KN_SCRATCH) + w * 512;
    const int g8 = a.g8, k = a.k, klw = a.klw;

    const int seg_begin = a.wg_off[blockIdx.x], seg_end = a.wg_end[blockIdx.x];   // this launch's share of the block's segments (phases: hb_launch_knn)
    for (int si = seg_begin; si < seg_end; ++si) {
        const hb_seg seg = a.segs[si];
        float* wl_s = a.state_s + (size_t)seg.slot * HB_QT * klw;
        unsigned* wl_i = a.state_i + (size_t)seg.slot * HB_QT * klw;
        float thr[2];
        {
#pragma nounroll
            for (int e = lane; e < 2048; e += 64) {   // this wave's 64 queries x 32 entries
                lst_s[w * 2048 + e] = seg.first ? -INFINITY : wl_s[w * 2048 + e];
                lst_i[w * 2048 + e] = seg.first ? HB_ID_NONE : wl_i[w * 2048 + e];
            }
            thr[0] = lst_s[(w * 64 + (lane & 31)) * HB_KL + (k - 1)];
            thr[1] = lst_s[(w * 64 + 32 + (lane & 31)) * HB_KL + (k - 1)];
        }
        const int total = seg.n_tiles * g8;
        f32x16 acc[16];
        f32x4 fa[4], fy[4], fb[2], fbk[2];

        // wave w stages bank row-tiles 2w, 2w+1 and query row-tiles 2w, 2w+1 of one k8 group (4 x 1 KiB)
        const float* bsrc0 = a.bank_tiles + (size_t)(2 * w) * g8 * HB_BLK + lane * 4;
        const float* qsrc0 = a.q_tiles + ((size_t)(seg.q_tile * 8 + 2 * w) * g8) * HB_BLK + lane * 4;
        auto issue = [&](int which, int bt, int ks, int slot) {
            char* sb = smem + slot * KN_SLOT_BYTES;
            if (which == 0) glds16(bsrc0 + ((size_t)bt * 8 * g8 + ks) * HB_BLK, sb + (2 * w) * 1024);
            else if (which == 1) glds16(bsrc0 + ((size_t)bt * 8 * g8 + g8 + ks) * HB_BLK, sb + (2 * w + 1) * 1024);
            else if (which == 2) glds16(qsrc0 + (size_t)ks * HB_BLK, sb + 8192 + (2 * w) * 1024);
            else {
                glds16(qsrc0 + (size_t)(g8 + ks) * HB_BLK, sb + 8192 + (2 * w + 1) * 1024);
                if (ks == 0 && w == 0) glds16(a.binit + (size_t)bt * HB_BT + lane * 4, smem + KN_BINIT + (bt & 1) * 1024);
            }
        };

        int bt = seg.b_tile0;
        int fbt = seg.b_tile0, fks = 0;
        int slot_c = 0, slot_f = 0;
        int left = total;
        auto advance_fetch = [&]() {
            if (--left > 0) { if (++fks == g8) { fks = 0; ++fbt; } }
            if (++slot_f == KN_RING) slot_f = 0;
        };
        // three stages in flight; 4 copies per stage per wave (wave 0: a 5th at the first stage of a tile)
        for (int p = 0; p < 3; ++p) {
            issue(0, fbt, fks, slot_f); issue(1, fbt, fks, slot_f); issue(2, fbt, fks, slot_f); issue(3, fbt, fks, slot_f);
            advance_fetch();
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        {
            const f32x4* A = reinterpret_cast<const f32x4*>(smem) + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = A[t * 64];
            fb[0] = A[8192 / 16 + (2 * w) * 64];
            fb[1] = A[8192 / 16 + (2 * w + 1) * 64];
        }
        for (int tl = 0; tl < seg.n_tiles; ++tl, ++bt) {
            // accumulators start from the bank rows' init values (published with the tile's first stage)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();
            {
                const f32x4* bi = reinterpret_cast<const f32x4*>(smem + KN_BINIT + (bt & 1) * 1024);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = bi[8 * t + 2 * g + h];
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            acc[2 * t + nb][4 * g + 0] = v[0]; acc[2 * t + nb][4 * g + 1] = v[1];
                            acc[2 * t + nb][4 * g + 2] = v[2]; acc[2 * t + nb][4 * g + 3] = v[3];
                        }
                    }
            }
#pragma nounroll
            for (int ks = 0; ks < g8; ++ks) {
                if (ks > 0) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // my copies of the next stage have landed
                    __syncthreads();
                }
                int slot_n = slot_c + 1; if (slot_n == KN_RING) slot_n = 0;
                const f32x4* Ac = reinterpret_cast<const f32x4*>(smem + slot_c * KN_SLOT_BYTES) + lane;
                const f32x4* An = reinterpret_cast<const f32x4*>(smem + slot_n * KN_SLOT_BYTES) + lane;
                // ---- X half: row tiles 0-3 x both query tiles, k-steps 0-3 (32 MFMAs) ----
                KN_FENCE W4_MFMA2(0, fa, fb, 0) KN_FENCE fy[0] = Ac[4 * 64];
                KN_FENCE W4_MFMA2(1, fa, fb, 0) KN_FENCE fy[1] = Ac[5 * 64];
                KN_FENCE W4_MFMA2(2, fa, fb, 0) KN_FENCE fy[2] = Ac[6 * 64];
                KN_FENCE W4_MFMA2(3, fa, fb, 0) KN_FENCE fy[3] = Ac[7 * 64];
                KN_FENCE W4_MFMA2(0, fa, fb, 1) W4_MFMA2(1, fa, fb, 1) KN_FENCE
                issue(0, fbt, fks, slot_f);
                KN_FENCE W4_MFMA2(2, fa, fb, 1) W4_MFMA2(3, fa, fb, 1) KN_FENCE
                issue(1, fbt, fks, slot_f);
                KN_FENCE W4_MFMA2(0, fa, fb, 2) W4_MFMA2(1, fa, fb, 2) KN_FENCE
                issue(2, fbt, fks, slot_f);
                KN_FENCE W4_MFMA2(2, fa, fb, 2) W4_MFMA2(3, fa, fb, 2) KN_FENCE
                issue(3, fbt, fks, slot_f);
                KN_FENCE W4_MFMA2(0, fa, fb, 3) W4_MFMA2(1, fa, fb, 3) KN_FENCE
                advance_fetch();
                KN_FENCE W4_MFMA2(2, fa, fb, 3) W4_MFMA2(3, fa, fb, 3) KN_FENCE
                fbk[0] = fb[0]; fbk[1] = fb[1];
                // ---- Y half: row tiles 4-7; fillers: next stage's X fragments and query fragments ----
                KN_FENCE W4_MFMA2(4, fy, fbk, 0) KN_FENCE fa[0] = An[0 * 64];
                KN_FENCE W4_MFMA2(5, fy, fbk, 0) KN_FENCE fa[1] = An[1 * 64];
                KN_FENCE W4_MFMA2(6, fy, fbk, 0) KN_FENCE fa[2] = An[2 * 64];
                KN_FENCE W4_MFMA2(7, fy, fbk, 0) KN_FENCE fa[3] = An[3 * 64];
                KN_FENCE W4_MFMA2(4, fy, fbk, 1) KN_FENCE fb[0] = An[8192 / 16 + (2 * w) * 64];
                KN_FENCE W4_MFMA2(5, fy, fbk, 1) KN_FENCE fb[1] = An[8192 / 16 + (2 * w + 1) * 64];
                KN_FENCE
                W4_MFMA2(6, fy, fbk, 1) W4_MFMA2(7, fy, fbk, 1)
                W4_MFMA2(4, fy, fbk, 2) W4_MFMA2(5, fy, fbk, 2) W4_MFMA2(6, fy, fbk, 2) W4_MFMA2(7, fy, fbk, 2)
                W4_MFMA2(4, fy, fbk, 3) W4_MFMA2(5, fy, fbk, 3) W4_MFMA2(6, fy, fbk, 3) W4_MFMA2(7, fy, fbk, 3)
                KN_FENCE
                slot_c = slot_n;
            }
            w4_epilogue(acc, thr, lst_s, lst_i, sc, w, lane, k, (unsigned)bt, HB_KL);
        }
        {
#pragma nounroll
            for (int e = lane; e < 2048; e += 64) { wl_s[w * 2048 + e] = lst_s[w * 2048 + e]; wl_i[w * 2048 + e] = lst_i[w * 2048 + e]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

hb_knn_fn hb_knn_w4_kernel(bool wide) {   // LDS-list path only (k <= HB_KL); the caller keeps k > HB_KL on the 8-wave kernel
    (void)wide;
    return (hb_knn_fn)knn_fused_w4_kernel;
}
