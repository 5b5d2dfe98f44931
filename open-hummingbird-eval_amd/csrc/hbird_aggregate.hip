// K5 of SURVEY.md 2.3: soft-label aggregation over the k neighbours of every query patch
// (reference hbird_eval.py:575-609 `_cross_attention` after the CPU index_select of 632-633).
//
//   q^ = q / max(||q||, 1e-12), k^_j = b_j / max(||b_j||, 1e-12)           (F.normalize, 594-595)
//   attn = softmax_j( (q^ . k^_j) / beta ),  label_hat = sum_j attn_j * label_j   (603-608)
//
// The kNN kernel already produced ip_j = q . b_j for the k neighbours, so q^.k^_j = ip_j / (||q|| ||b_j||)
// and only the k label rows (k*C*4 bytes per query) are gathered -- the k x D neighbour features
// the reference gathers on the CPU are never touched.  HBM-bound gather: one wave per query.
#include "hbird_internal.h"

#define AGG_MAX_K 256

// Two id ranges: the NORM table covers global ids [norm_base, norm_base + nnorm) and decides which neighbours take part in the
// softmax; the LABEL table covers [id_base, id_base + nlabels) and decides whose label rows are summed here.  They coincide for an
// ordinary index.  Label-sharded aggregation (hb_index_aggregate_partial): the norms of ALL rows are replicated (4 B per row), the
// label rows stay with their owners; every rank computes the same weights and the partial sum over the neighbours it owns, the
// all-reduce of the partial sums is label_hat ("distributed softmax + all-reduce").
// Label rows come as fp32 values or (U16) as uint16 counts j of values j / P -- what K2 produces: (float)j / (float)P -- at half the
// gather traffic and half the table in HBM (6.2 -> 3.1 GB at cfg-3, per rank when the table is replicated).  The very same fp32 value
// comes back from three instructions instead of a division: r = RN(1 / P) once, q' = RN(j r), e = fma(-q', P, j) (exact), q = fma(e, r, q')
// -- correctly rounded for every 0 <= j <= P <= 2048 (checked exhaustively: tests/test_ops_gpu.py::test_count_quotients_are_exact); larger
// denominators divide in place.  (Until round 5 a table of the P + 1 quotients in LDS: 64 lanes looking up random entries conflict three-
// to four-way, and the kernel ran slower on counts than on fp32 rows although it moved half the bytes.)
#define AGG_LUT 2048
template <bool U16>
__global__ __launch_bounds__(256) void aggregate_kernel(const void* __restrict__ labels_v, int ls, int wide, int P, int64_t nlabels, int C,
                                                        const float* __restrict__ bnorm, int64_t norm_base, int64_t nnorm,
                                                        const float* __restrict__ qnorm,
                                                        const int64_t* __restrict__ idx,
                                                        const float* __restrict__ dist, int64_t nq, int k,
                                                        int64_t id_base, int metric, const float* __restrict__ qn2,
                                                        float beta, float* __restrict__ out) {
    __shared__ float s_w[4][AGG_MAX_K];
    __shared__ int64_t s_row[4][AGG_MAX_K];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wv;
    const float* labels = reinterpret_cast<const float*>(labels_v);
    const unsigned short* counts = reinterpret_cast<const unsigned short*>(labels_v);
    const float Pf = (float)P, Pr = 1.0f / Pf;
    if (q >= nq) return;   // whole wave exits together (no block-level barrier below)
    float* wgt = s_w[wv];
    int64_t* rows = s_row[wv];
    // logits of the k neighbours (lane-strided), running maximum
    float mx = -INFINITY;
    for (int j = lane; j < k; j += 64) {
        float logit = -INFINITY;
        int64_t row = -1;
        const int64_t gid = idx[q * (int64_t)k + j];
        const int64_t r = gid - id_base, rn = gid - norm_base;
        if (gid >= 0 && rn >= 0 && rn < nnorm) {
            if (r >= 0 && r < nlabels) row = r;
            const float bn = fmaxf(bnorm[rn], 1e-12f);
            const float qn = fmaxf(qnorm[q], 1e-12f);
            float ip = dist[q * (int64_t)k + j];
            if (metric == 1) ip = 0.5f * (qn2[q] + bnorm[rn] * bnorm[rn] - ip);   // squared L2 -> inner product
            logit = (ip / (qn * bn)) / beta;
        }
        wgt[j] = logit;
        rows[j] = row;
        mx = fmaxf(mx, logit);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.0f;
    for (int j = lane; j < k; j += 64) {
        const float e = wgt[j] > -INFINITY ? expf(wgt[j] - mx) : 0.0f;   // every neighbour with a norm takes part (owned or not)
        wgt[j] = e;
        den += e;
    }
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
    const float inv = den > 0.0f ? 1.0f / den : 0.0f;
    // the weights as they enter the sum (attn_j = e_j / den), and row 0 with weight 0 for a neighbour whose label row is not here: the
    // gather below is then branch-free, so that a batch of loads is in flight before the first is used (the kernel is bound by the
    // gather's latency: 90 dependent-looking two-byte loads per lane at C = 151, k = 30 until round 5)
    for (int j = lane; j < k; j += 64) {
        const bool own = rows[j] >= 0;
        wgt[j] = own ? wgt[j] * inv : 0.0f;
        if (!own) rows[j] = 0;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to all its lanes
    auto label_at = [&](int64_t rj, int c) -> float {
        if (U16) {
            const float jf = (float)counts[rj * (int64_t)ls + c];
            if (P > AGG_LUT) return jf / Pf;
            const float q1 = jf * Pr;
            return fmaf(fmaf(-q1, Pf, jf), Pr, q1);
        }
        return labels[rj * (int64_t)ls + c];
    };
    constexpr int UB = 8;                  // label rows in flight per lane
    if (C <= 32) {
        // few classes (VOC 21, Cityscapes 19, COCO-Stuff 15 ...): G = 64 / C neighbours at a time, lane = (neighbour group g, class c);
        // group g sums the neighbours j = g, g + G, ... in ascending order, the G partial sums are added in group order
        const int G = 64 / C, g = lane / C, c = lane - g * C;
        const bool act = g < G;
        float accv = 0.0f;
        for (int j0 = 0; j0 < k; j0 += G * UB) {
            float lv[UB], wj[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = j0 + u * G + g;
                const bool in = act && j < k;
                wj[u] = in ? wgt[j] : 0.0f;
                lv[u] = in ? label_at(rows[j], c) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) accv = fmaf(wj[u], lv[u], accv);
        }
        float total = accv;                 // lanes of group 0: + group 1 + group 2 ...
        for (int gg = 1; gg < G; ++gg) total += __shfl(accv, gg * C + c);
        if (g == 0) out[q * (int64_t)C + c] = total;
        return;
    }
    if (U16 && wide) {
        // count rows of 16-byte granules (the index's own table, padded; hb_launch_aggregate checks stride, alignment, P and C): lane l gathers the eight counts 8 l .. 8 l + 7 of a row with ONE 16-byte load
        // -- ceil(C / 8) lanes cover a row (19 of 64 at C = 151), k loads per lane instead of 3 k two-byte ones: the kernel is bound by the
        // number of gather instructions in flight, not by lanes or bytes.  Every class still sums its neighbours in ascending order with
        // the same fmaf chain, so the bits equal the narrow path's (and the fp32 table's).
        const int nl = (C + 7) >> 3;
        float a8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a8[i] = 0.0f;
        if (lane < nl) {
            for (int j0 = 0; j0 < k; j0 += UB) {
                uint4 raw[UB];
                float wj[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = j0 + u;
                    wj[u] = j < k ? wgt[j] : 0.0f;
                    raw[u] = *reinterpret_cast<const uint4*>(counts + (j < k ? rows[j] : 0) * (int64_t)ls + 8 * lane);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const unsigned wds[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float jf = (float)((wds[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu);
                        const float q1 = jf * Pr;
                        a8[i] = fmaf(wj[u], fmaf(fmaf(-q1, Pf, jf), Pr, q1), a8[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * lane + i < C) out[q * (int64_t)C + 8 * lane + i] = a8[i];
        }
        return;
    }
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int cc = c < C ? c : C - 1;   // lanes past the last class repeat it (no store)
        float accv = 0.0f;
        for (int j0 = 0; j0 < k; j0 += UB) {
            float lv[UB], wj[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int j = j0 + u;
                wj[u] = j < k ? wgt[j] : 0.0f;
                lv[u] = j < k ? label_at(rows[j], cc) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) accv = fmaf(wj[u], lv[u], accv);
        }
        if (c < C) out[q * (int64_t)C + c] = accv;
    }
}

int hb_launch_aggregate(const hb_index* ix, const float* qnorm, const int64_t* idx, const float* dist, int64_t nq,
                        int k, int64_t id_base, float beta, float* out, hipStream_t s, const float* norms_all, int64_t n_all) {
    if (nq == 0) return 0;
    if (k > AGG_MAX_K) return hb_fail("hb_index_search_aggregate: k must be <= 256");
    const bool own16 = ix->label_P > 0;
    const void* labels = own16 ? (const void*)ix->labels16 : (const void*)ix->labels;
    const float* bnorm = ix->bnorm; int64_t nlab = ix->nlabels;
    int P = ix->label_P;
    const dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    // K5's wide gather (aggregate_kernel): rows of whole 16-byte granules at a 16-byte aligned base, the three-instruction quotient's range, at
    // most 64 lanes per row, and more classes than the (neighbour group, class) form takes
    auto wide_ok = [&](const void* tab, int stride, int PP) {
        return (stride & 7) == 0 && (reinterpret_cast<uintptr_t>(tab) & 15) == 0 && PP > 0 && PP <= AGG_LUT && ix->c > 32 && ix->c <= 512 ? 1 : 0;
    };
    if (norms_all) {   // label-sharded: this index's own label rows, everybody's norms
        if (!labels || ix->nlabels < ix->ntotal) return hb_fail("hb_index_aggregate_partial: label rows missing (hb_index_add_labels)");
        if (own16) aggregate_kernel<true><<<grid, block, 0, s>>>(labels, ix->lab_stride(), wide_ok(labels, ix->lab_stride(), P), P, ix->ntotal, ix->c, norms_all, 0, n_all, qnorm, idx, dist, nq, k, id_base, ix->metric, ix->q_aux, beta, out);
        else aggregate_kernel<false><<<grid, block, 0, s>>>(labels, ix->c, 0, 0, ix->ntotal, ix->c, norms_all, 0, n_all, qnorm, idx, dist, nq, k, id_base, ix->metric, ix->q_aux, beta, out);
        HB_HIP(hipGetLastError());
        return 0;
    }
    bool u16 = own16;
    int ls = ix->lab_stride();
    if (ix->ext_labels || ix->ext_labels16) {
        ls = ix->c;                                  // borrowed tables are dense [n, C]
        u16 = ix->ext_labels16 != nullptr;
        labels = u16 ? (const void*)ix->ext_labels16 : (const void*)ix->ext_labels; P = ix->ext_P;
        bnorm = ix->ext_bnorm; nlab = ix->ext_n; id_base = ix->ext_base;
    } else if (!labels || ix->nlabels < ix->ntotal) return hb_fail("hb_index_search_aggregate: label rows missing (hb_index_add_labels)");
    if (u16) aggregate_kernel<true><<<grid, block, 0, s>>>(labels, ls, wide_ok(labels, ls, P), P, nlab, ix->c, bnorm, id_base, nlab, qnorm, idx, dist, nq, k, id_base, ix->metric, ix->q_aux, beta, out);
    else aggregate_kernel<false><<<grid, block, 0, s>>>(labels, ls, 0, 0, nlab, ix->c, bnorm, id_base, nlab, qnorm, idx, dist, nq, k, id_base, ix->metric, ix->q_aux, beta, out);
    HB_HIP(hipGetLastError());
    return 0;
}

// fp32 label values -> uint16 counts: j = round(v P); the stored j is exact iff (float)j / (float)P == v (every value K2 produces);
// anything else raises the sticky flag (read once after the table grew: hb_labels_checked).  Dense [rows, c] in, rows of dst_stride out.
__global__ __launch_bounds__(256) void labels_to_counts_kernel(const float* __restrict__ src, int64_t n, int c, int dst_stride, int P,
                                                               unsigned short* __restrict__ dst, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = src[i], Pf = (float)P;
    const float r = rintf(v * Pf);
    const bool ok = r >= 0.0f && r <= Pf && r / Pf == v;
    dst[(i / c) * dst_stride + (i % c)] = ok ? (unsigned short)r : 0;
    if (!ok) *flag = 1;
}

int hb_launch_labels_to_counts(const float* src, int64_t rows, int c, int dst_stride, int P, uint16_t* dst, int* flag, hipStream_t s) {
    const int64_t n = rows * c;
    if (n == 0) return 0;
    labels_to_counts_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(src, n, c, dst_stride, P, dst, flag);
    HB_HIP(hipGetLastError());
    return 0;
}

// out[i, :] = counts[ids[i], :] / P as fp32 (label_memory.index_select on a table stored as counts); ids outside the table give zeros
__global__ __launch_bounds__(256) void gather_label_counts_kernel(const unsigned short* __restrict__ src, int64_t src_rows, int c, int src_stride, int P,
                                                                  const int64_t* __restrict__ ids, int64_t n, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * c) return;
    const int64_t i = t / c, r = ids ? ids[i] : i;      // (ids == nullptr: the rows in order -- hb_index_labels_to_fp32)
    out[t] = (r >= 0 && r < src_rows) ? (float)src[r * src_stride + (t % c)] / (float)P : 0.0f;
}

int hb_launch_gather_label_counts(const uint16_t* src, int64_t src_rows, int c, int src_stride, int P, const int64_t* ids, int64_t n, float* out, hipStream_t s) {
    if (n == 0) return 0;
    gather_label_counts_kernel<<<dim3((unsigned)((n * c + 255) / 256)), dim3(256), 0, s>>>(src, src_rows, c, src_stride, P, ids, n, out);
    HB_HIP(hipGetLastError());
    return 0;
}
