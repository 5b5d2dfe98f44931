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
