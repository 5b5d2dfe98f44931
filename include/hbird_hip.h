/*
 * hbird_hip.h -- C ABI of libhbird_hip.so: the MI355X (gfx950) drop-in for the kNN backend plugin and
 * the bank-build / label-aggregation hot path of vpariza/open-hummingbird-eval.
 *
 * The reference's boundary for this path is a Python class (hbird/nn/search_base.py:3-31) constructed
 * by HbirdEvaluation._create_nn (hbird/hbird_eval.py:267-281) and called once per validation batch from
 * _find_nearest_key_to_query (hbird_eval.py:628).  Its Faiss implementation (hbird/nn/search_faiss.py)
 * binds the third-party faiss-gpu C++/CUDA library; the entry points below are what that binding is
 * replaced with.  Citations are file:line under the reference tree.
 *
 * Conventions: every function returns 0 on success and a negative status on failure, in which case
 * hb_last_error() (thread-local) describes it.  All matrices are dense row-major fp32 / int64.  One
 * index handle (hb_index_t) lives on ONE GPU; one process per GPU composes shards and replicas above this ABI
 * with RCCL, one process with several GPUs uses an hb_multi_t, declared below.  Calls on one handle must be serialised by the caller; work is
 * enqueued on the handle's stream (hb_index_set_stream) and host-pointer variants synchronise it.
 * `*_on_device` = 1 means the pointer is device memory of the handle's GPU, 0 means host memory.
 */
#ifndef HBIRD_HIP_H
#define HBIRD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hb_index hb_index_t;

#define HB_METRIC_IP 0 /* faiss.GpuIndexFlatIP, search_faiss.py:43-44 ("dot_product")        */
#define HB_METRIC_L2 1 /* faiss.GpuIndexFlatL2, search_faiss.py:45-46 ("l2" / "euclidean")   */
#define HB_MAX_K 2048  /* neighbours per query of a search -- faiss-gpu's own limit (the reference forwards any k, search_faiss.py:84-85; its default
                        * is 30).  k <= 32 keeps the lists in LDS, k <= 256 is one pass over candidate pools, beyond that ceil(k / 256) passes,
                        * each behind the last neighbour of the one before: a search of k = 1024 costs four searches of k = 256. */
#define HB_MAX_K_AGGREGATE 256   /* ... of the fused search + label aggregation (K5 keeps a query's k weights in LDS) and of sharded merges */

const char* hb_last_error(void);
/* faiss.get_num_gpus(), search_faiss.py:14 */
int hb_device_count(int* n);

/* faiss.StandardGpuResources + GpuIndexFlatConfig{device} + GpuIndexFlat{IP,L2}(res, d, config),
 * search_faiss.py:34-48.  `device` is validated like gpu_ids at search_faiss.py:23-25. */
int hb_index_create(int d, int metric, int device, hb_index_t** out);
int hb_index_free(hb_index_t* ix);
/* hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream. */
int hb_index_set_stream(hb_index_t* ix, void* hip_stream);
/* Pre-size the device-resident bank (rows); optional, add() grows geometrically otherwise. */
int hb_index_reserve(hb_index_t* ix, int64_t n_rows);
/* index.add(np.float32[n, d]), search_faiss.py:78-81.  Appendable (successive ids, like
 * faiss.IndexShards' successive-id convention, search_faiss.py:56-63).  normalize = 1 fuses
 * `features / torch.norm(features, dim=2, keepdim=True)` (hbird_eval.py:324, 335: no eps) into the append
 * so the per-batch device->host copy of hbird_eval.py:328 disappears. */
int hb_index_add(hb_index_t* ix, const float* x, int64_t n, int x_on_device, int normalize);
/* label_memory rows (hbird_eval.py:329, 354), co-indexed with the bank rows. */
int hb_index_add_labels(hb_index_t* ix, const float* labels, int64_t n, int c, int on_device);
/* Every label value of the reference is j / P, P = patch_size^2: one_hot(...).float().mean(dim=3) over the P pixels of a patch
 * (hbird_eval.py:319-320, computed as (float)j / (float)P).  With a denominator set BEFORE the first label row, the index stores the
 * uint16 count j instead of the fp32 value -- half the table (6.2 -> 3.1 GB at cfg-3) and half the aggregation's gather traffic --
 * and hands back exactly the same fp32 values.  A value that is not such a multiple fails the next call that reads the labels.
 * 0 (default) = fp32 storage. */
int hb_index_set_label_denominator(hb_index_t* ix, int P);
int hb_index_label_denominator(const hb_index_t* ix, int* P);
/* The stored counts back to fp32 rows, in place and in one pass (the value (float)j / (float)P of every count; denominator 0 afterwards):
 * for a bank whose training batches come in more than one input size -- the reference recomputes patch_size per batch (hbird_eval.py:313-314),
 * and a table of counts holds ONE denominator.  A no-op on an fp32 table. */
int hb_index_labels_to_fp32(hb_index_t* ix);
int64_t hb_index_ntotal(const hb_index_t* ix);
int64_t hb_index_nlabels(const hb_index_t* ix);
/* Drop all rows and labels (keeps allocations). */
int hb_index_reset(hb_index_t* ix);

/* index.search(np.float32[nq, d], k) -> (D, I), search_faiss.py:84-90.  out_idx[nq, k] int64 (id_base +
 * local row, -1 when fewer than k rows exist), out_dist[nq, k] fp32 (inner product, descending, or
 * squared L2, ascending), rows best-first, ties by lower id.  1 <= k <= HB_MAX_K. */
int hb_index_search(hb_index_t* ix, const float* q, int64_t nq, int k, int64_t id_base, int64_t* out_idx,
                    float* out_dist, int io_on_device);
/* search + hbird_eval.py:632-633 (label gather) + 575-609 (_cross_attention, beta = 0.02) fused:
 * out_label_hat[nq, c].  out_idx_opt / out_dist_opt may be NULL. */
int hb_index_search_aggregate(hb_index_t* ix, const float* q, int64_t nq, int k, int64_t id_base, float beta,
                              float* out_label_hat, int64_t* out_idx_opt, float* out_dist_opt, int io_on_device);
/* The aggregation step alone, on given (merged) neighbours -- used after a multi-GPU top-k merge.
 * idx are global ids; q, idx, dist, out are device pointers (io_on_device must be 1). */
int hb_index_aggregate(hb_index_t* ix, const float* q, int64_t nq, const int64_t* idx, const float* dist, int k,
                       int64_t id_base, float beta, float* out_label_hat, int io_on_device);
/* Label-sharded aggregation (multi-GPU without replicating label_memory: 6.2 GB at cfg-3, 16.7 GB for the full ADE20K bank): the
 * bank-row norms of ALL rows are replicated (norms_all[n_all], global ids from 0: 4 B per row), the label rows stay with their
 * owners.  Every rank calls this with the same merged neighbours and gets the softmax-weighted sum over the neighbours IT owns
 * (global ids [id_base, id_base + ntotal)); the sum of the ranks' out_partial[nq, c] (an all-reduce) is label_hat.  The weights are
 * computed identically everywhere; only the order of the fp32 sum differs from hb_index_aggregate (within 1e-6).  Device pointers. */
int hb_index_aggregate_partial(hb_index_t* ix, const float* q, int64_t nq, const int64_t* idx, const float* dist, int k,
                               int64_t id_base, float beta, const float* norms_all, int64_t n_all, float* out_partial);
/* feature_memory.index_select(0, idx) (hbird_eval.py:632) for return_knn_details; out[n, d]. */
int hb_index_reconstruct(hb_index_t* ix, const int64_t* ids, int64_t n, int64_t id_base, float* out,
                         int io_on_device);
/* label_memory.index_select(0, idx) (hbird_eval.py:633); out[n, c]. */
int hb_index_gather_labels(hb_index_t* ix, const int64_t* ids, int64_t n, int64_t id_base, float* out,
                           int io_on_device);

/* Multi-GPU aggregation tables: borrow device arrays labels[n, c] / norms[n] that cover the GLOBAL id range
 * [id_base, id_base + n) (the all-gathered label_memory and bank-row norms); NULL restores the index's own
 * tables.  hb_index_copy_norms exports this shard's row norms (ntotal floats) for that all-gather. */
int hb_index_set_label_table(hb_index_t* ix, const float* labels, const float* norms, int64_t n, int c,
                             int64_t id_base);
int hb_index_copy_norms(hb_index_t* ix, float* out, int on_device);
/* The same for a table held as counts (hb_index_set_label_denominator): hb_index_copy_label_counts exports this shard's nlabels x c
 * uint16 counts for the all-gather, hb_index_set_label_count_table borrows the gathered counts[n, c] / norms[n] of denominator P. */
int hb_index_copy_label_counts(hb_index_t* ix, uint16_t* out, int on_device);
int hb_index_set_label_count_table(hb_index_t* ix, const uint16_t* counts, const float* norms, int64_t n, int c, int P,
                                   int64_t id_base);

/* Sharded searches (faiss.IndexShards, search_faiss.py:53-63): with score output enabled a search returns the
 * ORDERING score in out_dist (inner product: q.b; L2: q.b - |b|^2/2; larger is better; missing neighbours -inf)
 * instead of the metric's distance.  Squared L2 distances round away differences that the ordering still sees, so the
 * cross-shard merge must run on the scores (hb_merge_topk with metric 0) to reproduce the single-index order bit for
 * bit; hb_index_distances_from_scores then converts the merged scores in place (device pointers; a no-op for the
 * inner product) exactly as the single-index search does. */
int hb_index_set_score_output(hb_index_t* ix, int enable);
int hb_index_distances_from_scores(hb_index_t* ix, const float* q, int64_t nq, int k, float* dist_inout);
/* k-way merge of per-shard results laid out [parts][nq][k] (faiss.IndexShards' merge, search_faiss.py:
 * 53-63; here fed by an RCCL all-gather).  Device pointers. */
int hb_merge_topk(const float* dist_parts, const int64_t* idx_parts, int parts, int64_t nq, int k, int metric,
                  int64_t* out_idx, float* out_dist, void* hip_stream);
/* The same merge on PACKED per-shard lists, so that one rank's result travels in ONE all-gather message: a packed
 * list is hb_packed_list_bytes(nq, k) bytes = [nq*k int64 ids][nq*k fp32 scores] (+ padding to 16 B); a search writes
 * it when out_idx = buf and out_dist = (float*)((char*)buf + nq*k*8).  packed_parts holds `parts` such lists
 * part_bytes apart (the all-gather's output). */
int64_t hb_packed_list_bytes(int64_t nq, int k);
int hb_merge_topk_packed(const void* packed_parts, int64_t part_bytes, int parts, int64_t nq, int k, int metric,
                         int64_t* out_idx, float* out_dist, void* hip_stream);

/* ---- several GPUs behind one handle ------------------------------------------------------------------ */
/* faiss.index_cpu_to_gpu_multiple_py(resources, index_cpu, gpus=gpu_ids) with co.shard = idx_shard, search_faiss.py:50-76:
 * one hb_index_t per entry of gpu_ids (an id may repeat) driven by one host thread each (faiss `threaded = True`, 57).
 * shard = 1: faiss.IndexShards (53-63) -- contiguous row ranges with successive ids, every GPU searches all queries, the
 * [nq, k] lists are merged by (ordering score descending, id ascending) and converted to the metric's distances as the
 * single-index search converts them: the single-index result bit for bit.  shard = 0: faiss.IndexReplicas (65-74) -- every
 * GPU holds all rows, the queries are split.  All buffers are HOST memory (the reference hands numpy arrays, 80-81, 88).
 * hb_multi_reserve plans the row count (shards: equal contiguous ranges; rows beyond the plan go to the last shard; without a
 * plan everything goes to the first).  hb_multi_shard_rows: rows held by the first n entries.  hb_multi_set_fp16 as
 * hb_index_set_fp16.  Host composition of the entries above; in Python the same is hbird_mi.nn.search_hip.HipMultiIndex. */
typedef struct hb_multi hb_multi_t;
int hb_multi_create(int d, int metric, const int* gpu_ids, int n_gpus, int shard, hb_multi_t** out);
int hb_multi_free(hb_multi_t* m);
int hb_multi_reserve(hb_multi_t* m, int64_t n_rows);
int hb_multi_add(hb_multi_t* m, const float* x, int64_t n, int normalize);
int64_t hb_multi_ntotal(const hb_multi_t* m);
int hb_multi_shard_rows(const hb_multi_t* m, int64_t* rows, int n);
int hb_multi_set_fp16(hb_multi_t* m, int enable);
int hb_multi_search(hb_multi_t* m, const float* q, int64_t nq, int k, int64_t* out_idx, float* out_dist);

/* ---- bank build (device pointers, enqueued on hip_stream) -------------------------------------- */
/* features / torch.norm(features, dim=-1, keepdim=True), hbird_eval.py:324, 335. */
int hb_normalize_rows(const float* x, int64_t n, int d, float* out, void* hip_stream);
/* _patchify_gt + one_hot(...).float().mean(dim=3), hbird_eval.py:555-573, 319-320; y[B,1,H,W] int64 ->
 * out[B, H/ps, W/ps, C]; map255 = 1 applies `y[y == 255] = 0` (hbird_eval.py:310) on the fly.  A class value outside
 * [0, C) fails the call (F.one_hot raises for it, hbird_eval.py:319): the error flag is read back after the kernel, so
 * this entry synchronises hip_stream. */
int hb_patch_label_hist(const int64_t* y, int64_t B, int H, int W, int ps, int C, int map255, float* out,
                        void* hip_stream);
/* _sample_features scores, hbird_eval.py:471-493 (presence read from the soft labels). */
int hb_patch_scores(const float* label, int64_t B, int SS, int C, float* scores, int* nonempty, int* nz_count,
                    void* hip_stream);
/* noise multiply + K smallest, hbird_eval.py:497-511; r drawn by the caller from torch's CPU generator. */
int hb_patch_select(const float* scores, const int* nonempty, const float* r, const int64_t* r_off, int64_t B,
                    int SS, int K, int64_t* out_idx, float* out_scores, void* hip_stream);
/* out[i, :] = src[ids[i], :] (features.gather / label.gather, hbird_eval.py:515, 344-346). */
int hb_gather_rows(const float* src, int64_t src_rows, int width, const int64_t* ids, int64_t n, float* out,
                   void* hip_stream);

/* ---- after the aggregation ------------------------------------------------------------------------ */
/* reshape/permute + F.interpolate(bilinear) + argmax, hbird_eval.py:235-243; label_hat[B, S*S, C] ->
 * out[B, 1, h, w] int64. */
int hb_upsample_argmax(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out,
                       void* hip_stream);
/* The two steps fused (hbird_eval.py:235-243 + eval_metrics.py:73-104): label_hat[B, S*S, C] is upsampled, its argmax is counted
 * straight into conf[num_gt, num_pred] against the masks gt[B, 1, h, w] (gt == ignore_index and out-of-range pairs dropped, as
 * hb_confusion_update) -- the int64 class map is never written unless out_map_opt asks for it. */
int hb_upsample_argmax_confusion(const float* label_hat, int64_t B, int S, int C, int h, int w, const int64_t* gt, int num_gt,
                                 int num_pred, int64_t ignore_index, int has_ignore, uint64_t* conf, int64_t* out_map_opt,
                                 void* hip_stream);
/* Sliding-window frames (BASELINE cfg-5; the reference has no tiler): one window's label_hat[B, S*S, C] is upsampled
 * like hbird_eval.py:240 (bilinear, align_corners=False) to win_h x win_w and added into acc[B, H, W, C] (fp32,
 * channels last, zeroed by the caller) at (y0, x0).  Windows of one frame must be accumulated in a fixed order
 * (fp32 sums).  hb_argmax_channels then gives the frame's class map: out[n] = argmax_c acc[n, c], first maximum
 * wins (the argmax of hbird_eval.py:243). */
int hb_upsample_accumulate(const float* label_hat, int64_t B, int S, int C, int win_h, int win_w, float* acc, int H,
                           int W, int y0, int x0, void* hip_stream);
int hb_argmax_channels(const float* acc, int64_t n, int C, int64_t* out, void* hip_stream);
/* PredsmIoU.update, hbird/utils/eval_metrics.py:73-104; conf[num_gt, num_pred] uint64 accumulated. */
int hb_confusion_update(const int64_t* gt, const int64_t* pred, int64_t n, int num_gt, int num_pred,
                        int64_t ignore_index, int has_ignore, uint64_t* conf, void* hip_stream);

/* ---- measurement / tuning ------------------------------------------------------------------------- */
/* When enabled, every search brackets the kNN kernel with HIP events on the handle's stream. */
int hb_index_set_timing(hb_index_t* ix, int enable);
int hb_index_last_knn_ms(const hb_index_t* ix, double* ms);
/* Overrides for tests: number of workgroups (0 = one per CU) and bank tiles per panel (0 = auto); negative values are errors. */
int hb_index_set_tuning(hb_index_t* ix, int workgroups, int panel_tiles);
/* GpuIndexFlatConfig.useFloat16 (search_faiss.py:40): 1 = searches run an fp16 candidate pass (fp16 copies of the
 * fragment tiles, fp16 MFMA, k' >= 2k candidates) followed by an exact fp32 re-rank of the candidates, so the
 * returned indices / distances are those of the fp32 search.  Every query carries a certificate (exact k-th score >
 * k'-th fp16 score + rounding bound); queries that fail it are searched again with the fp32 kernel, so the result is
 * ALWAYS the fp32 result.  Applies to k <= 128; larger k use the fp32 kernel.  2 = the same, but only for banks of at
 * least 4,096 rows with rows x queries x D >= 1.5e10 x (k' / 64)^2, k' = 2k -- below that the fp32 kernel is the faster way to the same result (what the
 * plugin's use_fp16 sets). */
int hb_index_set_fp16(hb_index_t* ix, int enable);
/* Number of queries of the last fp16-mode search that needed the exact fp32 re-search. */
int hb_index_last_fp16_fallbacks(const hb_index_t* ix, int64_t* n);
/* What happens to a query whose certificate fails.  mode 0 (default): ESCALATION -- the failing queries, compacted, get a second fp16 pass
 * with k' = 256 candidates whose pools start from the floor (exact k-th best of the first pass - 1.001 E): every row that can still enter
 * the top k scores above it in fp16, so the pass appends little, a list that does not fill up is complete by construction, and a full one
 * is certified against a rank four times further down; only what fails again is searched by the fp32 kernel, from the exact k-th best
 * found so far as its floor.  mode 1: straight to the fp32 kernel (round 5's behaviour; every started tile of 256 failing queries costs a
 * whole-bank fp32 pass: 27 ms at 10 M x 768).  Same results either way: always the fp32 search's bits.
 * hb_index_last_fp16_escalated: queries of the last use_fp16 search whose FIRST certificate failed (hb_index_last_fp16_fallbacks: those
 * that reached the fp32 kernel). */
int hb_index_set_fp16_escalation(hb_index_t* ix, int mode);
int hb_index_last_fp16_escalated(const hb_index_t* ix, int64_t* n);
/* kNN kernel variant, for A/B runs and tests (same results): 0 = default; 3 = the fp32 kernel with register-resident query fragments
 * wherever it applies (D padded to a multiple of 32: what the default does too); 4 = never that kernel (both operands staged through
 * LDS); 6 = small fp32 searches with k <= 32 on sorted LDS lists as until round 3 (the default runs them on phased candidate pools).
 * 1 (4-wave fp32 kernel), 2 (first design of the fp16 candidate kernel) and 5 (16x16x32 fp16 kernel) no longer exist: same bits,
 * not faster (profiles/LABBOOK.md). */
int hb_index_set_variant(hb_index_t* ix, int variant);
/* Two more A/B switches (same results): phases = 0 launches a pool search (k > 32, use_fp16, small fp32 searches) once instead of in
 * phases; small_limit_stages > 0 moves the size (k8 stages per workgroup) below which a search takes the small-search kernels (0 =
 * the built-in 400,000). */
int hb_index_set_search_options(hb_index_t* ix, int phases, int64_t small_limit_stages);
/* (Round 5's opt-in "one launch per phased search" -- hb_index_set_one_launch / _one_launch_stats / _one_launch_trace -- measured 2-10 % slower
 * than a launch per phase at every size and was removed in round 6; numbers: profiles/r05/README.md.) */
/* Work shares per XCD.  The eight XCDs of one MI355X do not run the fp32 kNN kernel at one speed (with equal work the workgroups of the odd
 * XCDs finish 1-2 % after those of the even ones, and a launch lasts as long as its slowest workgroup: profiles/r05/xcd_speed_stamps_
 * headline.txt), so the work list gives group x -- the blocks equal to x mod 8, which the hardware deals to one XCD -- the share
 * w[x] / sum(w) of every panel's pairs instead of one eighth.  Speed only: any shares give the same results.
 * mode 0 (default) = calibrated: searches from about 30 ms of fp32 kernel stamp their workgroups' start and end, and the next such search turns
 * each group's median duration into its share (2275 -> 2258 ms at 10 M x 768; never waits for the stamps).  The fp32 kernels and the fp16
 * candidate kernel (274 -> 270 ms) calibrate shares of their own; a phased search's cuts follow the shares.  1 = equal shares; 2 = the eight
 * shares given (searches of any size, both families).  hb_index_xcd_weights: the shares in use by the fp32 kernels (fp16_kernel = 0) or the
 * fp16 candidate kernel (1) and the calibration rounds so far.
 * hb_schedule_plan_weighted: the host-only planner with such shares (tests; shared bit 0 = XCD-level query sharing, bit 1 = a phased list,
 * whose cuts follow the shares). */
int hb_index_set_xcd_weights(hb_index_t* ix, int mode, const double* w8);
int hb_index_xcd_weights(const hb_index_t* ix, int fp16_kernel, double* w8, int* rounds);
int hb_schedule_plan_weighted(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, int shared,
                              const double* xcd_w8, int* segs_out, int64_t max_segs, int64_t stats[8]);
/* Diagnostics: with hb_index_set_timing on, every workgroup of the last kNN launch (the last phase's, for a phased search) stamps the 100 MHz
 * real-time counter when it starts and when it ends, and notes the XCD it ran on: out[block][4] = {start, end, XCC id, 0} (low 32 bits).
 * Shows per-XCD speed differences: the launch lasts as long as its slowest workgroup. */
int hb_index_wg_stamps(hb_index_t* ix, uint32_t* out, int max_blocks, int* workgroups);
/* The clock the last stamped kNN launch ran at, WITHOUT a profiler attached: every workgroup also stamps the shader-cycle counter
 * (s_memtime) beside the real-time counter, and (cycles / 10 ns ticks) x 100 MHz is the clock its CU held over the launch.  out[0] = median
 * over the workgroups (GHz), [1] / [2] = slowest / fastest workgroup, [3] = the launch's span in ms by the stamps (first start to last end).
 * Zeros when the last launch did not stamp (hb_index_set_timing off and no share calibration).  One stream synchronisation. */
int hb_index_kernel_clock(hb_index_t* ix, double out[4]);
/* State of the share calibration of one kernel family (fp16_kernel as above): out[0] = calibration rounds, [1] = 1 when the GUARD has locked
 * the shares -- a share set whose launches (same shape, shortest of at least two) measured 0.15 % slower than the best set seen (the fp16
 * candidate kernel, whose launches scatter by 0.5 %: three launches, 0.8 %) is dropped, the
 * best set returns and this index stops calibrating --, 2 when the map from block groups to XCDs kept moving (equal shares from then on),
 * [2] = reverts by the guard, [3] = stamp sets read, [4] = stamp sets rejected (a workgroup that did not stamp, blocks equal mod 8 that did not
 * share an XCD, a time far from the others'), [5] / [6] = shortest launch in ms with the best / the current share set, [7] = work lists built
 * for this index so far (a re-plan costs host time: 8 ms at 10 M x 768), [8] = moves of the group -> XCD map, [9] = the XCD block 0 was last
 * seen on, [10] (fp32 family) = the measured decision about the automatic L2-sharing clusters of the biggest fp32 searches: -1 still
 * measuring, 1 kept, 0 dropped (two calibrated launches with, two without, the faster form stays: hb_index_set_cluster), [11] = shortest
 * launch with minus without clusters in ms.  In calibrated mode the shares (hb_index_xcd_weights) belong to the PHYSICAL XCDs 0-7 as
 * HW_REG_XCC_ID numbers them. */
int hb_index_xcd_stats(const hb_index_t* ix, int fp16_kernel, double out[12]);
/* The calibration's decisions without a GPU (tests; like hb_schedule_plan* for the planner): a state as an index keeps per kernel family, fed with
 * the stamp sets of imagined launches.  hb_calibration_new(fp16_kernel) -> handle; hb_calibration_state: the GROUP shares the next launch would
 * run with (group g = blocks equal to g mod 8) and out[0..11] = rounds, locked, reverts, samples, rejected, moves of the group -> XCD map, cluster
 * state (0 measuring with, 1 measuring without, 2 decided), cluster choice, launches of the current share set, the XCD of block 0, launches measured
 * with / without clusters; hb_calibration_feed: one launch's stamps [G][2][4] = per block at its start and end {100 MHz ticks, XCC id, shader
 * cycles lo, hi}, the group shares it ran with, its shape key {query tiles, bank tiles, workgroups, phases, k, cluster shape q * 16 + b}, whether
 * that cluster shape was the automatic choice, the launch's part of the search's work -> flags: 1 rebuild the work list, 2 remember the shares, 4
 * remember the cluster decision, 8 the set was rejected (negative: bad arguments). */
void* hb_calibration_new(int fp16_kernel);
void hb_calibration_free(void* h);
int hb_calibration_state(const void* h, double shares8[8], int64_t out[12]);
int hb_calibration_feed(void* h, const uint32_t* stamps, int G, const double run_shares8[8], const int key6[6], int auto_cluster, double frac);
/* ... and the adaptive use of use_fp16 (mode 2) on a stream of n imagined searches of nq queries: f1[i] / f2[i] = the share of queries that would
 * fail the first certificate / of those the second pass at search i; out_how[i] = how the index would run search i given what it saw before:
 * 0 the chain (first pass, second pass for its failures, fp32 for the rest), 1 one pass with k' = 256 for all queries, 2 the fp32 kernel right away. */
int hb_f16_adapt_replay(int n, const double* f1, const double* f2, int64_t nq, int* out_how);
/* K1 (the fused normalise + fragment-tiled append, also the query re-tiling of every search) has two forms with the same bits: the rows staged
 * through LDS once (widths that are multiples of 16 up to 1152, 16-byte aligned sources) and the first form for everything else.  form 1 forces
 * the first form everywhere (process-wide; tests hold the two forms to each other), 0 = automatic, 8 / 16 / 32 = the LDS form with that many rows
 * per workgroup (A/B runs).  Twice the first form's rate on 500 k-row appends (0.49-0.59 of the HBM roofline at D = 384 / 768 / 1024). */
int hb_set_layout_form(int form);
/* use_fp16 searches re-rank their candidates in exact fp32 arithmetic.  In the fragment tiles a bank row is 2 x D/8 sixteen-byte pieces
 * 512 B apart, so that pass pulls eight times the bytes it uses; a second, row-major fp32 copy of the bank lets it read whole lines (a
 * use_fp16 search at 300,000 x 768: 8.0 -> 6.6 ms, k = 90: 18.0 -> 12.4; results identical).  What it saves is a few ms of re-rank per search whatever
 * the bank's size, what it costs is the bank once more: 17 % of a search for 0.9 GB at 300,000 x 768, 1.5 % for 30.7 GB at 10 M x 768, 0.4 % for
 * 83 GB at 20 M x 1024 (profiles/r06/final/fp16_residency.json).  mode 0 = automatic: the copy is made at the first use_fp16 search for banks of up
 * to 16 GB (5.2 M x 768) when fp32 tiles + fp16 tiles + this copy stay within 55 % of the device's memory with room to spare -- a bigger bank's
 * use_fp16 index holds 1.5 x the bank, not 2.5 x (the whole ADE20K bank, 27.7 M x 768: 130 GB) --; 1 = always (an error if the allocation fails);
 * 2 = never (an existing copy is released).  hb_index_rerank_copy_bytes: what the copy occupies now (0 = none). */
int hb_index_set_rerank_copy(hb_index_t* ix, int mode);
int hb_index_rerank_copy_bytes(const hb_index_t* ix, int64_t* bytes);
/* Work-list statistics of the last search: out[0]=workgroups, [1]=segments, [2]=slots, [3]=panel tiles,
 * [4]=max slots per query tile, [5]=query tiles, [6]=bank tiles, [7]=cluster shape (query ways * 16 + bank ways). */
int hb_index_schedule_info(const hb_index_t* ix, int64_t out[8]);
/* L2-sharing clusters of the kNN work list (speed only; results never depend on it): cluster_q x cluster_b workgroups
 * of one XCD walk (cluster_q query tiles) x (cluster_b interleaved bank tiles) in lockstep, so that one L2 fill serves
 * several workgroups.  0 x 0 = automatic (the fp16 candidate kernel of big searches: 8 x 1, or 4 x 2 / 2 x 2 when that
 * idles fewer pairs; the fp32 kernel: 2 x 4 or 2 x 2, only from one million k8 stages per workgroup up -- it is bound by
 * the matrix pipe, clusters cut its fabric reads by 60 % and cost cycles: a box held below its nominal clock by its power budget ran 2.5 %
 * faster with them, boxes at 2.38-2.39 GHz 0.2-0.8 % slower, so the calibration times both forms during an index's first five big searches and
 * keeps the faster one: hb_index_xcd_stats [10]), 1 x 1 = off, q x b with q * b <= 8 otherwise.  sync_lag: stages a member may run ahead of the slowest one before it waits (-1 = 16, 0 = never). */
int hb_index_set_cluster(hb_index_t* ix, int cluster_q, int cluster_b, int sync_lag);
/* How a clustered work list is dealt (speed only): 2 = every run of one query group is split over ALL clusters of an XCD, so that
 * the XCD's workgroups keep re-reading the same cluster_q query tiles -- they stay in its L2 instead of being streamed through the
 * fabric once per pair (the clusters of an XCD need no sync among themselves for that); 1 = each cluster takes its own range of
 * the q-major unit list; 0 = automatic. */
int hb_index_set_cluster_sharing(hb_index_t* ix, int mode);
/* Soft-sync statistics of the last clustered search (synchronises the stream): out[0] = progress checks, [1] = waits
 * (re-polls while a member was behind), [2] = members that gave up waiting (bounded spin); zeros without clusters. */
int hb_index_cluster_stats(hb_index_t* ix, int64_t out[4]);

/* Host-only (no GPU needed): the work list the kNN kernel would run for nqt query tiles (256 rows) x nbt bank tiles
 * (256 rows) on `workgroups` workgroups; panel_tiles = 0 selects the automatic panel; cluster_q / cluster_b as in
 * hb_index_set_cluster (-1 = the fp16 candidate kernel's automatic shape, -2 = the fp32 kernel's).  segs_out (may be NULL) receives up to max_segs rows {block, q_tile,
 * b_tile0, n_tiles, slot, first, tile stride, cluster clock at the first tile, cluster clock of the block's next segment,
 * progress word of the block (-1: no cluster)}; stats as hb_index_schedule_info. */
int hb_schedule_plan(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, int* segs_out,
                     int64_t max_segs, int64_t stats[8]);
/* The work list of a PHASED search (pools: k > 32 and the fp16 candidate pass): launched in phases of growing size, between
 * which the k-th best of all rows seen so far becomes every partial pool's floor.  Every block's segments are cut at the same
 * clocks: clocks_out receives up to max_cuts of them (*n_cuts: how many there are), bounds_out [cut][block] the position,
 * within the block's own segments, of the first segment of the next phase. */
int hb_schedule_plan_phased(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, int* segs_out,
                            int64_t max_segs, int64_t stats[8], int* clocks_out, int max_cuts, int* n_cuts, int* bounds_out);

/* hb_schedule_plan / _phased for the work list of hb_index_set_cluster_sharing(ix, 2). */
int hb_schedule_plan_shared(int nqt, int nbt, int workgroups, int panel_tiles, int d, int cluster_q, int cluster_b, int phased,
                            int* segs_out, int64_t max_segs, int64_t stats[8]);

#ifdef __cplusplus
}
#endif
#endif /* HBIRD_HIP_H */
