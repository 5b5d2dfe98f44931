// Device-side helpers shared by the kNN kernel variants (hbird_knn.hip, hbird_knn_w4.hip).
#pragma once
#include "hbird_internal.h"

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
    int klw;  // list row stride in the state buffer: HB_KL, or k rounded up to 64 when k > HB_KL
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

// Same insertion for lists that live in global memory (k > HB_KL): row stride KLW = k rounded up to 64,
// each lane holds KLW/64 entries.  Only the owning wave ever touches a query's list; loads bypass the L1 and
// the stores are drained before the next insertion reads the list again.
__device__ __forceinline__ void list_insert_wide(float* gs, unsigned* gi, int k, int klw, float s, unsigned id, int lane) {
    float es[4];
    unsigned ei[4];
    const int E = klw >> 6;
    int p = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e < E) {
            const int j = e * 64 + lane;
            es[e] = __hip_atomic_load(gs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ei[e] = __hip_atomic_load(gi + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool better = j < k && ((es[e] > s) || (es[e] == s && ei[e] < id));
            p += __popcll(__ballot(better));
        }
    }
    if (p >= k) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e < E) {
            const int j = e * 64 + lane;
            if (j >= p && j < k - 1) { gs[j + 1] = es[e]; gi[j + 1] = ei[e]; }
        }
    }
    if (lane == 0) { gs[p] = s; gi[p] = id; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// dump the 4 registers of quarter Q (rows 8Q .. 8Q+7 of the 32-row tile) of accumulator tile T
#define HB_DUMP_CASE(T, Q)                                                                  \
    case (4 * (T) + (Q)):                                                                   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) sc[r * 64 + lane] = acc[T][4 * (Q) + r]; \
        break;
#define HB_DUMP_TILE(T) HB_DUMP_CASE(T, 0) HB_DUMP_CASE(T, 1) HB_DUMP_CASE(T, 2) HB_DUMP_CASE(T, 3)

// Epilogue of one (query tile, bank tile) pair for one wave: filter the wave's 256 x 32 scores against the
// per-query thresholds (phase 1, always) and insert the rare survivors into the lists (phase 2).
template <bool SLOW = true, bool WIDE = false>
__device__ __forceinline__ void tile_epilogue(f32x16 (&acc)[8], float& thr, float* lst_s, unsigned* lst_i, float* sc,
                                              int w, int lane, int k, unsigned bt, int klw = HB_KL) {
    unsigned qmask = 0;   // bit 4t+q: quarter q (8 bank rows) of row tile t holds a score above its query's threshold
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool any = (acc[t][4 * q] > thr) | (acc[t][4 * q + 1] > thr) | (acc[t][4 * q + 2] > thr) | (acc[t][4 * q + 3] > thr);
            if (__ballot(any) != 0ull) qmask |= 1u << (4 * t + q);
        }
    if (!SLOW) { asm volatile("" :: "s"(qmask)); return; }
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
                    float kth;
                    if constexpr (WIDE) {
                        float* gs = lst_s + (size_t)(w * 32 + n) * klw;
                        list_insert_wide(gs, lst_i + (size_t)(w * 32 + n) * klw, k, klw, s, row_base + hh * 4 + j, lane);
                        kth = __hip_atomic_load(gs + (k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        list_insert(lst_s, lst_i, w * 32 + n, k, s, row_base + hh * 4 + j, lane);
                        kth = lst_s[(w * 32 + n) * HB_KL + (k - 1)];
                    }
                    if ((lane & 31) == n) thr = kth;
                }
            }
    }
}

// LDS map shared by the variants (bytes): a 4-slot ring of k8 stages, two row-init buffers, the lists, the scratch
#define KN_SLOT_BYTES 16384                 // 8 KiB bank fragments + 8 KiB query fragments (32 rows x 8 k blocks)
#define KN_RING 4
#define KN_BINIT (KN_RING * KN_SLOT_BYTES)  // 2 x 1 KiB
#define KN_LISTS (KN_BINIT + 2048)
#define KN_SCRATCH (KN_LISTS + 2 * HB_QT * HB_KL * 4)
#define KN_LDS_TOTAL (KN_SCRATCH + 8192)      // scratch: 1 KiB per wave (8 waves) / 2 KiB per wave (4 waves)
#define KN_FENCE __builtin_amdgcn_sched_barrier(0);

typedef void (*hb_knn_fn)(knn_args);
hb_knn_fn hb_knn_w4_kernel(bool wide);   // 4-wave (one wave per SIMD) variant, hbird_knn_w4.hip
