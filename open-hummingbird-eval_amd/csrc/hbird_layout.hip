// Layout kernels: row-major fp32 rows <-> MFMA-fragment tiles, plus the per-row constants the
// kNN kernel needs (accumulator init, norms).  gfx950 only.
//
// K1 of SURVEY.md 2.3 ("bank_normalize_append"): `features / torch.norm(features, dim=2,
// keepdim=True)` (reference hbird_eval.py:324, 335 -- no eps) is fused into the append: rows are
// read once from the extractor's output and written straight into the index's device-resident,
// fragment-tiled bank (no host round trip, reference 328-329/353).
//
// Fragment-tile layout (one 1 KiB block = 32 rows x 8 k):
//   block(rt, g) at float offset ((rt * G8) + g) * 256,  rt = row / 32, g = k / 8, G8 = Dp / 8
//   element (i = row % 32, kk = k % 8) at  ((kk & 1) * 32 + i) * 4 + (kk >> 1)
// so that lane l = h*32 + i of a wave reads ONE float4 at 16*l bytes holding k = 8g + {h, 2+h, 4+h,
// 6+h}; four successive v_mfma_f32_32x32x2_f32 then consume k = (8g, 8g+1), (8g+2, 8g+3), ... in
// ascending order, i.e. every score is a k-ascending fmaf chain (bit-exact oracle definition).
#include "hbird_internal.h"
#include <algorithm>

__device__ __forceinline__ double wave8_sum(double v) {
    // reduce over aligned groups of 8 lanes
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

// One block = 32 consecutive source rows. 256 threads: 8 per row for the reductions.
// src may be null (zero fill only) for rows >= n_valid. Destination rows are row0 + local row.
template <bool NORMALIZE, bool IS_BANK>
__global__ __launch_bounds__(256) void rows_to_tiles_kernel(const float* __restrict__ src, int64_t n_valid,
                                                            int64_t n_cover, int d, int dp, int64_t row0,
                                                            float* __restrict__ tiles, float* __restrict__ binit,
                                                            float* __restrict__ bnorm, int metric) {
    __shared__ float s_scale[32];
    const int tid = threadIdx.x;
    const int64_t r_base = (int64_t)blockIdx.x * 32;
    const int g8 = dp >> 3;

    // ---- pass 1: row norms (double accumulation, one rounding to fp32) ----
    {
        const int i = tid >> 3, sub = tid & 7;
        const int64_t r = r_base + i;
        double acc = 0.0;
        if (r < n_valid) {
            const float* row = src + r * (int64_t)d;
            if ((d & 3) == 0) {
                const float4* row4 = reinterpret_cast<const float4*>(row);
                for (int k4 = sub; k4 < (d >> 2); k4 += 8) {
                    float4 v = row4[k4];
                    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
                }
            } else {
                for (int k = sub; k < d; k += 8) { float v = row[k]; acc += (double)v * v; }
            }
        }
        acc = wave8_sum(acc);
        float nrm = (float)sqrt(acc);
        float scale_div = NORMALIZE ? nrm : 1.0f;   // stored value = x / scale_div
        if (sub == 0) s_scale[i] = scale_div;
        if (IS_BANK) {
            // norm of the STORED row (what F.normalize(k) of hbird_eval.py:595 would recompute)
            float stored_nrm = nrm;
            if (NORMALIZE) {
                double a2 = 0.0;
                if (r < n_valid) {
                    const float* row = src + r * (int64_t)d;
                    for (int k = sub; k < d; k += 8) { float v = row[k] / nrm; a2 += (double)v * v; }
                }
                a2 = wave8_sum(a2);
                stored_nrm = (float)sqrt(a2);
            }
            if (sub == 0 && r < n_valid) bnorm[row0 + r] = stored_nrm;
        }
    }
    __syncthreads();

    // ---- pass 2: scatter into fragment tiles ----
    {
        const int s = tid & 3, i = (tid >> 2) & 31, h = tid >> 7;
        const int kk = 2 * s + h;
        const int64_t r = r_base + i;
        if (r < n_cover) {
            const int64_t dr = row0 + r;
            const bool valid = r < n_valid;
            const float sc = s_scale[i];
            const float* row = valid ? src + r * (int64_t)d : nullptr;
            float* dst = tiles + ((dr >> 5) * (int64_t)g8) * HB_BLK + ((kk & 1) * 32 + (int)(dr & 31)) * 4 + (kk >> 1);
            for (int g = 0; g < g8; ++g) {
                const int k = 8 * g + kk;
                float v = 0.0f;
                if (valid && k < d) { v = row[k]; if (NORMALIZE) v = v / sc; }
                dst[(int64_t)g * HB_BLK] = v;
            }
        }
    }

    // ---- pass 3 (bank): accumulator init value per row ----
    if (IS_BANK && tid < 32) {
        const int64_t r = r_base + tid;
        if (r < n_valid) {
            float init = 0.0f;
            if (metric == 1) {
                // -0.5 * ||b||^2 with ||b||^2 as ONE k-ascending fmaf chain over the stored values
                const float* row = src + r * (int64_t)d;
                const float sc = s_scale[tid];
                float acc = 0.0f;
                for (int k = 0; k < d; ++k) { float v = row[k]; if (NORMALIZE) v = v / sc; acc = fmaf(v, v, acc); }
                init = -0.5f * acc;
            }
            binit[row0 + r] = init;
        }
    }
}

// K1, second form (round 6): the block's 32 rows go through LDS once.  The first form reads its rows three times with 4-byte strided
// accesses (the norms, the stored rows' norms, the scatter: 0.39 of the HBM roofline on 500 k x 768 chunks); here they come in by coalesced
// 16-byte loads (eight in flight per thread), every pass of the first form then reads LDS -- the same per-thread loops in the same order, so
// the same bits in tiles, bnorm and binit (tests/test_ops_gpu.py holds the two forms to each other bit for bit) -- and the tiles leave as
// whole 16-byte pieces.  Twice the first form's rate: 0.49-0.59 of the HBM roofline on whole 500 k-row appends at D = 384 / 768 / 1024.  Rows are padded by four floats in LDS (lane i reads row i: without
// the padding all 32 rows of a width that is a multiple of 32 start in one bank).  For widths that are multiples of 16 (then Dp == D) up
// to 1152 (32 x (D + 4) x 4 bytes of LDS) and 16-byte aligned sources; everything else keeps the first form.
#define K1L_MAX_D 1152
template <bool NORMALIZE, bool IS_BANK, int ROWS>
__global__ __launch_bounds__(256) void rows_to_tiles_lds_kernel(const float* __restrict__ src, int64_t n_valid, int64_t n_cover, int d, int64_t row0,
                                                                float* __restrict__ tiles, float* __restrict__ binit, float* __restrict__ bnorm,
                                                                int metric) {
    extern __shared__ __attribute__((aligned(16))) float s_rows[];     // [ROWS][d + 4]
    __shared__ float s_scale[ROWS];
    const int tid = threadIdx.x;
    const int64_t r_base = (int64_t)blockIdx.x * ROWS;
    const int ld = d + 4, d4 = d >> 2, g8 = d >> 3;
    // ---- pass 0: the block's rows -> LDS (rows beyond n_valid: zeros) ----
    {
        const int n_here = (int)std::max<int64_t>(0, std::min<int64_t>(ROWS, n_valid - r_base));
        const float4* src4 = reinterpret_cast<const float4*>(src + r_base * (int64_t)d);
        const int total = ROWS * d4;
        for (int f0 = tid; f0 < total; f0 += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * 256;
                v[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (f < total && f / d4 < n_here) v[u] = src4[f];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * 256;
                if (f < total) { const int row = f / d4, c4 = f - row * d4; *reinterpret_cast<float4*>(s_rows + row * ld + 4 * c4) = v[u]; }
            }
        }
    }
    __syncthreads();
    // ---- pass 1: row norms (double accumulation, one rounding to fp32): the first form's loops, from LDS (eight threads per row) ----
    if (tid < ROWS * 8) {
        const int i = tid >> 3, sub = tid & 7;
        const int64_t r = r_base + i;
        const float* row = s_rows + i * ld;
        double acc = 0.0;
        if (r < n_valid)
            for (int k4 = sub; k4 < d4; k4 += 8) {
                const float4 v = *reinterpret_cast<const float4*>(row + 4 * k4);
                acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            }
        acc = wave8_sum(acc);
        const float nrm = (float)sqrt(acc);
        if (sub == 0) s_scale[i] = NORMALIZE ? nrm : 1.0f;
        if (IS_BANK) {
            float stored_nrm = nrm;
            if (NORMALIZE) {
                double a2 = 0.0;
                if (r < n_valid)
                    for (int k = sub; k < d; k += 8) { const float v = row[k] / nrm; a2 += (double)v * v; }
                a2 = wave8_sum(a2);
                stored_nrm = (float)sqrt(a2);
            }
            if (sub == 0 && r < n_valid) bnorm[row0 + r] = stored_nrm;
        }
    }
    __syncthreads();
    // ---- pass 2: fragment tiles, 16 bytes per thread (row i, half h: k = 8 g + {h, 2 + h, 4 + h, 6 + h}); 2 ROWS threads per k8 group ----
    {
        constexpr int PER = 2 * ROWS;
        const int slot = tid % PER, grp = tid / PER, h = slot / ROWS, i = slot % ROWS;
        const int64_t r = r_base + i;
        if (r < n_cover) {
            const int64_t dr = row0 + r;
            const bool valid = r < n_valid;
            const float sc = s_scale[i];
            const float* row = s_rows + i * ld + h;
            float* dst = tiles + ((dr >> 5) * (int64_t)g8) * HB_BLK + (h * 32 + (int)(dr & 31)) * 4;
            for (int g = grp; g < g8; g += 256 / PER) {
                float4 o;
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                if (valid) {
                    v0 = row[8 * g]; v1 = row[8 * g + 2]; v2 = row[8 * g + 4]; v3 = row[8 * g + 6];
                    if (NORMALIZE) { v0 = v0 / sc; v1 = v1 / sc; v2 = v2 / sc; v3 = v3 / sc; }
                }
                o.x = v0; o.y = v1; o.z = v2; o.w = v3;
                *reinterpret_cast<float4*>(dst + (int64_t)g * HB_BLK) = o;
            }
        }
    }
    // ---- pass 3 (bank): accumulator init value per row ----
    if (IS_BANK && tid < ROWS) {
        const int64_t r = r_base + tid;
        if (r < n_valid) {
            float init = 0.0f;
            if (metric == 1) {
                const float* row = s_rows + tid * ld;
                const float sc = s_scale[tid];
                float acc = 0.0f;
                for (int k = 0; k < d; ++k) { float v = row[k]; if (NORMALIZE) v = v / sc; acc = fmaf(v, v, acc); }
                init = -0.5f * acc;
            }
            binit[row0 + r] = init;
        }
    }
}

static int g_layout_form = 0;      // 0 = automatic (the LDS form where it applies), 1 = the first form everywhere (tests: the two forms' bits)
static int g_layout_rows = 0;      // rows per workgroup of the LDS form (0 = automatic; forms 32 / 16 / 8 force one: A/B runs)
extern "C" int hb_set_layout_form(int form) {
    if (form != 0 && form != 1 && form != 8 && form != 16 && form != 32) return hb_fail("hb_set_layout_form: 0 (automatic), 1 (the first form) or 8 / 16 / 32 (rows per workgroup of the LDS form)");
    g_layout_form = form == 1 ? 1 : 0;
    g_layout_rows = form > 1 ? form : 0;
    return 0;
}

int hb_launch_rows_to_tiles(const float* src, int64_t n_rows, int d, int dp, int64_t row0, float* tiles, float* binit,
                            float* bnorm, int metric, int normalize, int is_bank, hipStream_t s) {
    // queries: cover up to the next HB_QT boundary with zeros so that stale workspace never leaks in
    int64_t n_cover = is_bank ? n_rows : ((n_rows + HB_QT - 1) / HB_QT) * HB_QT;
    if (n_cover == 0) return 0;
    dim3 grid((unsigned)((n_cover + 31) / 32)), block(256);
    if (g_layout_form == 0 && d % 16 == 0 && d == dp && d <= K1L_MAX_D && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // rows per workgroup: 32 / 16 / 8 by the width (LDS: rows x (D + 4) x 4 bytes of the CU's 160 KiB)
        // (measured on 500 k-row chunks, whole appends, of 8 TB/s: D = 384: first form 0.27, 32 rows 0.55, 16 rows 0.59, 8 rows 0.54; D = 768: 0.26 / 0.32 /
        // 0.49 / 0.50; D = 1024: 0.25 / 0.32 / 0.39 / 0.48 -- what counts is several resident workgroups per CU: about 50 KB of LDS each)
        const int rows = g_layout_rows > 0 ? g_layout_rows : (d <= 192 ? 32 : d <= 828 ? 16 : 8);
        const int lds = rows * (d + 4) * 4;
        typedef void (*k1_fn)(const float*, int64_t, int64_t, int, int64_t, float*, float*, float*, int);
        k1_fn fn;
        if (rows == 32) fn = is_bank ? (normalize ? (k1_fn)rows_to_tiles_lds_kernel<true, true, 32> : (k1_fn)rows_to_tiles_lds_kernel<false, true, 32>) : (k1_fn)rows_to_tiles_lds_kernel<false, false, 32>;
        else if (rows == 16) fn = is_bank ? (normalize ? (k1_fn)rows_to_tiles_lds_kernel<true, true, 16> : (k1_fn)rows_to_tiles_lds_kernel<false, true, 16>) : (k1_fn)rows_to_tiles_lds_kernel<false, false, 16>;
        else fn = is_bank ? (normalize ? (k1_fn)rows_to_tiles_lds_kernel<true, true, 8> : (k1_fn)rows_to_tiles_lds_kernel<false, true, 8>) : (k1_fn)rows_to_tiles_lds_kernel<false, false, 8>;
        if (hb_ensure_dyn_lds((const void*)fn, lds)) return -1;
        fn<<<dim3((unsigned)((n_cover + rows - 1) / rows)), block, lds, s>>>(src, n_rows, n_cover, d, row0, tiles, binit, bnorm, metric);
        HB_HIP(hipGetLastError());
        return 0;
    }
    if (is_bank) {
        if (normalize) rows_to_tiles_kernel<true, true><<<grid, block, 0, s>>>(src, n_rows, n_cover, d, dp, row0, tiles, binit, bnorm, metric);
        else rows_to_tiles_kernel<false, true><<<grid, block, 0, s>>>(src, n_rows, n_cover, d, dp, row0, tiles, binit, bnorm, metric);
    } else {
        rows_to_tiles_kernel<false, false><<<grid, block, 0, s>>>(src, n_rows, n_cover, d, dp, row0, tiles, nullptr, nullptr, metric);
    }
    HB_HIP(hipGetLastError());
    return 0;
}

// Per-query constants: qn2 = k-ascending fmaf chain of q_k^2 (L2 distances), qnorm = fp32 L2 norm
// (double accumulation) used by the cosine aggregation (F.normalize(q), hbird_eval.py:594).
// One thread per query keeps both sums in their sequential order (qn2 is part of the bit-exact L2 distances).  One WAVE per block:
// the kernel is two dependent chains of d steps per thread, so what it needs is every CU busy (12,544 queries in blocks of 256 were 49
// blocks: 57 us; staging the rows through LDS for coalesced reads made it 96 us -- more latency in the chain, not less), and 16 bytes
// per load (a thread's row stays in its CU's L1 between the loads that share a cache line).
__global__ __launch_bounds__(64) void query_aux_kernel(const float* __restrict__ q, int64_t nq, int d,
                                                       float* __restrict__ qn2, float* __restrict__ qnorm) {
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= nq) return;
    const float* row = q + r * (int64_t)d;
    float acc = 0.0f;
    double a2 = 0.0;
    int k = 0;
    if ((d & 3) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
#pragma unroll 4
        for (; k < d; k += 4) {
            const float4 v4 = *reinterpret_cast<const float4*>(row + k); const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc = fmaf(v[j], v[j], acc); a2 += (double)v[j] * v[j]; }
        }
    }
    for (; k < d; ++k) { const float v = row[k]; acc = fmaf(v, v, acc); a2 += (double)v * v; }
    qn2[r] = acc;
    qnorm[r] = (float)sqrt(a2);
}

// sharded searches: ordering scores s = q.b - |b|^2/2 of the L2 metric -> squared distances, exactly as the
// single-index merge converts them: d = max(0, fma(-2, s, |q|^2)) with the chain |q|^2; missing neighbours -> +inf
__global__ __launch_bounds__(256) void scores_to_l2_kernel(const float* __restrict__ qn2, int64_t n, int k, float* __restrict__ dist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = dist[i];
    if (s == -INFINITY) { dist[i] = INFINITY; return; }
    const float d2 = fmaf(-2.0f, s, qn2[i / k]);
    dist[i] = d2 > 0.0f ? d2 : 0.0f;
}

int hb_launch_scores_to_l2(const float* qn2, int64_t nq, int k, float* dist_inout, hipStream_t s) {
    const int64_t n = nq * k;
    if (n == 0) return 0;
    scores_to_l2_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(qn2, n, k, dist_inout);
    HB_HIP(hipGetLastError());
    return 0;
}

int hb_launch_query_aux(const float* q, int64_t nq, int d, float* qn2, float* qnorm, hipStream_t s) {
    if (nq == 0) return 0;
    query_aux_kernel<<<dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, s>>>(q, nq, d, qn2, qnorm);
    HB_HIP(hipGetLastError());
    return 0;
}

// Gather bank rows back to row-major (return_knn_details: key_features of hbird_eval.py:632,635).
// ids < 0 (missing neighbour) produce zero rows.  One wave per output row.
__global__ __launch_bounds__(256) void tiles_to_rows_kernel(const float* __restrict__ tiles, int g8, int d,
                                                            const int64_t* __restrict__ ids, int64_t n,
                                                            int64_t id_base, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n) return;
    const int64_t gid = ids[o];
    float* dst = out + o * (int64_t)d;
    if (gid < 0) { for (int k = lane; k < d; k += 64) dst[k] = 0.0f; return; }
    const int64_t r = gid - id_base;
    const float* base = tiles + ((r >> 5) * (int64_t)g8) * HB_BLK;
    const int i = (int)(r & 31);
    for (int k = lane; k < d; k += 64) {
        const int g = k >> 3, kk = k & 7;
        dst[k] = base[(int64_t)g * HB_BLK + ((kk & 1) * 32 + i) * 4 + (kk >> 1)];
    }
}

int hb_launch_tiles_to_rows(const float* tiles, int g8, int d, const int64_t* ids, int64_t n, int64_t id_base,
                            float* out, hipStream_t s) {
    if (n == 0) return 0;
    tiles_to_rows_kernel<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s>>>(tiles, g8, d, ids, n, id_base, out);
    HB_HIP(hipGetLastError());
    return 0;
}

// Standalone row normalisation (row-major in, row-major out) -- used by the bounded-memory build
// where sampled rows are normalised before the label gather (hbird_eval.py:335).
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, int64_t n, int d,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* row = x + r * (int64_t)d;
    double acc = 0.0;
    for (int k = lane; k < d; k += 64) { float v = row[k]; acc += (double)v * v; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float nrm = (float)sqrt(acc);
    for (int k = lane; k < d; k += 64) out[r * (int64_t)d + k] = row[k] / nrm;
}

int hb_launch_normalize_rows(const float* x, int64_t n, int d, float* out, hipStream_t s) {
    if (n == 0) return 0;
    normalize_rows_kernel<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s>>>(x, n, d, out);
    HB_HIP(hipGetLastError());
    return 0;
}

// out[i, :] = src[ids[i], :] for row-major fp32 tables (label rows: hbird_eval.py:344-346, 633).
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t src_rows, int width,
                                                          const int64_t* __restrict__ ids, int64_t n,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n) return;
    const int64_t r = ids[o];
    const bool ok = r >= 0 && r < src_rows;
    for (int c = lane; c < width; c += 64) out[o * (int64_t)width + c] = ok ? src[r * (int64_t)width + c] : 0.0f;
}

int hb_launch_gather_rows(const float* src, int64_t src_rows, int width, const int64_t* ids, int64_t n, float* out,
                          hipStream_t s) {
    if (n == 0) return 0;
    gather_rows_kernel<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s>>>(src, src_rows, width, ids, n, out);
    HB_HIP(hipGetLastError());
    return 0;
}
