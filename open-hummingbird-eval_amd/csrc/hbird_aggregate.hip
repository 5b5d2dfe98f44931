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
__global__ __launch_bounds__(256) void aggregate_kernel(const float* __restrict__ labels, int64_t nlabels, int C,
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
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible to all its lanes
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float accv = 0.0f;
        if (c < C)
            for (int j = 0; j < k; ++j) {
                const int64_t rj = rows[j];
                if (rj >= 0) accv = fmaf(wgt[j] * inv, labels[rj * (int64_t)C + c], accv);
            }
        if (c < C) out[q * (int64_t)C + c] = accv;
    }
}

int hb_launch_aggregate(const hb_index* ix, const float* qnorm, const int64_t* idx, const float* dist, int64_t nq,
                        int k, int64_t id_base, float beta, float* out, hipStream_t s, const float* norms_all, int64_t n_all) {
    if (nq == 0) return 0;
    if (k > AGG_MAX_K) return hb_fail("hb_index_search_aggregate: k must be <= 256");
    const float* labels = ix->labels; const float* bnorm = ix->bnorm; int64_t nlab = ix->nlabels;
    if (norms_all) {   // label-sharded: this index's own label rows, everybody's norms
        if (!ix->labels || ix->nlabels < ix->ntotal) return hb_fail("hb_index_aggregate_partial: label rows missing (hb_index_add_labels)");
        aggregate_kernel<<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s>>>(labels, ix->ntotal, ix->c, norms_all, 0, n_all, qnorm, idx, dist, nq, k,
                                                                             id_base, ix->metric, ix->q_aux, beta, out);
        HB_HIP(hipGetLastError());
        return 0;
    }
    if (ix->ext_labels) { labels = ix->ext_labels; bnorm = ix->ext_bnorm; nlab = ix->ext_n; id_base = ix->ext_base; }
    else if (!ix->labels || ix->nlabels < ix->ntotal) return hb_fail("hb_index_search_aggregate: label rows missing (hb_index_add_labels)");
    aggregate_kernel<<<dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s>>>(labels, nlab, ix->c, bnorm, id_base, nlab, qnorm,
                                                                         idx, dist, nq, k, id_base, ix->metric,
                                                                         ix->q_aux, beta, out);
    HB_HIP(hipGetLastError());
    return 0;
}
