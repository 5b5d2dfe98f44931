// Post-kNN kernels ("next" row f1 of SURVEY.md 8): bilinear upsample + argmax (K6) and the streaming
// confusion matrix (K7).
#include "hbird_internal.h"

// The interpolation arithmetic below is specified step by step in fp32 (as ATen and the oracle evaluate it), so
// contraction into FMAs is switched off for this file.  (The __f*_rn intrinsics do not help: they are inline
// functions defined under the default contraction mode and fuse after inlining.)
#pragma clang fp contract(off)

// K6 -- reference hbird_eval.py:235-243: label_hat[B, S*S, C] -> reshape [B,S,S,C] -> permute
// [B,C,S,S] -> F.interpolate(size=(h,w), mode="bilinear", align_corners=False) -> argmax(dim=1).
// Fused: the [B,C,h,w] fp32 tensor (2.6 GB at cfg-3) is never materialised.  Source index as ATen's
// area_pixel_compute_source_index: src = max(0, scale*(dst+0.5)-0.5), scale = S/h in fp32;
// value = ly0*(lx0*v00 + lx1*v01) + ly1*(lx0*v10 + lx1*v11); ties -> lowest class.
//
// Work decomposition.  All output rows whose upper source row is j (a "band": ~h/S rows) interpolate between the
// same two source rows j and j+1, and `top` / `bot` of the formula depend on (source row, x, class) only.  One
// workgroup = (image, band j, chunk of <= R rows of the band, 64..256 output columns); it stages the two source rows'
// needed columns x C classes in LDS (25 KB at cfg-3), and each lane (= one output column) walks the classes once:
// 4 LDS reads (broadcast-friendly: a wave's 64 columns touch ~6 source columns) + top/bot per class, then for each of its
// rows v = ly0*top + ly1*bot and a running (best, argbest) in registers.  Same fp32 steps in the same order as the
// per-pixel formula (contraction off), so the class maps are bit-identical to the one-thread-per-pixel kernel this
// replaces (1.31 ms -> see profiles/r03 at the cfg-3 batch; 649 M class evaluations, 4 global loads each before).
__device__ __forceinline__ int k6_src0(float scale, int dst, int S) {
    const float f = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
    return min((int)floorf(f), S - 1);
}
// first output index whose upper source index is >= j (bands are contiguous: the source index is monotone)
__device__ __forceinline__ int k6_band_begin(float scale, int j, int S, int n) {
    if (j <= 0) return 0;
    if (j > S - 1) return n;
    int g = (int)(((float)j + 0.5f) / scale - 0.5f);
    g = max(0, min(g, n));
    while (g > 0 && k6_src0(scale, g - 1, S) >= j) --g;
    while (g < n && k6_src0(scale, g, S) < j) ++g;
    return g;
}

// K6 + K7 fused (row f1 of SURVEY.md 8: "upsample + argmax + confusion matrix"): with `conf` the kernel also counts its pixels into the
// confusion matrix conf[gt, pred] (PredsmIoU.update, eval_metrics.py:73-104: gt == ignore_index and out-of-range pairs dropped) and
// `out` may be NULL -- the int64 class map (8 B per pixel written, 16 B per pixel re-read by K7 with the masks) then never exists.
// A wave holds 64 neighbouring pixels of one row: segmentation maps are piecewise constant, so the wave first groups equal (gt, pred)
// pairs (leader election by ballot: usually one or two groups) and issues ONE 64-bit global atomic per group -- K7's per-pixel LDS
// atomics serialise exactly there (64 lanes on one bin).
// Where the counts go (`hbits`): few classes (G x P bins within 16 KiB: C <= 64) -> a plain [G x P] histogram in LDS (hbits = 0), so that
// noise-like maps -- every lane its own pair -- cost one LDS atomic per pixel; until round 5 they overflowed a 256-entry hash table and
// sent 3.2 M same-bin global atomics through the L2 (0.26 ms at the cfg-2 batch against 0.09 for the two separate kernels).  More
// classes -> a (pair, count) hash table of 2^hbits entries in the staging area (open addressing, four probes); a wave that still holds
// more than 16 ungrouped pixels after its three leader rounds is looking at noise, where a table buys nothing (as many distinct pairs as
// pixels): those lanes add to the matrix directly instead of probing a full table eight times each.
#define K6_HIST_BYTES (16 * 1024)
template <int K6_R>
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* __restrict__ lh, int S, int C, int h, int w,
                                                              float sy, float sx, int chunks, int cmax,
                                                              int64_t* __restrict__ out, const int64_t* __restrict__ gt, int G, int P,
                                                              int64_t ignore, int has_ignore, unsigned long long* __restrict__ conf, int hbits) {
    extern __shared__ __attribute__((aligned(16))) float k6_sm[];
    const int j = blockIdx.y / chunks, rc = blockIdx.y % chunks;
    const int64_t b = blockIdx.z;
    const int r0 = k6_band_begin(sy, j, S, h) + rc * K6_R;
    const int r1 = min(k6_band_begin(sy, j + 1, S, h), r0 + K6_R);
    if (r0 >= r1) return;                                   // block-uniform
    const int bw = blockDim.x;                              // 64 .. 256 columns per block (whole waves)
    const int xs = blockIdx.x * bw, xe = min(w, xs + bw) - 1;
    const int cx_lo = k6_src0(sx, xs, S), cx_hi = min(k6_src0(sx, xe, S) + 1, S - 1);
    const int ncols = cx_hi - cx_lo + 1;
    const int j1 = min(j + 1, S - 1);
    const int x = min(xs + (int)threadIdx.x, w - 1);        // lanes past the right border repeat the last column (no store)
    const float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.0f);
    const int x0 = min((int)floorf(fx), S - 1), x1 = min(x0 + 1, S - 1);
    const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
    float ly0[K6_R], ly1[K6_R], best[K6_R];
    int arg[K6_R];
#pragma unroll
    for (int r = 0; r < K6_R; ++r) {
        const int yy = min(r0 + r, r1 - 1);                 // rows past the chunk's end repeat its last row (no store)
        const float fy = fmaxf(sy * ((float)yy + 0.5f) - 0.5f, 0.0f);
        ly1[r] = fy - (float)j;                             // j = min(floor(fy), S - 1) for every row of the band
        ly0[r] = 1.0f - ly1[r];
        best[r] = 0.0f; arg[r] = 0;
    }
    const float* base = lh + b * (int64_t)S * S * C;
    for (int c0 = 0; c0 < C; c0 += cmax) {
        const int cn = min(cmax, C - c0);
        if (c0) __syncthreads();
        // stage [2 source rows][ncols][cn]: for cn == C a source row's columns are one contiguous run in global memory
        const int per_row = ncols * cn;
        if (cn == C) {                                      // the columns of a source row are one contiguous run
            const float* s0 = base + ((int64_t)j * S + cx_lo) * C;
            const float* s1 = base + ((int64_t)j1 * S + cx_lo) * C;
            for (int e = threadIdx.x; e < per_row; e += bw) { k6_sm[e] = s0[e]; k6_sm[per_row + e] = s1[e]; }
        } else
            for (int e = threadIdx.x; e < 2 * per_row; e += bw) {
                const int row = e >= per_row, rem = e - row * per_row;
                const int col = rem / cn, c = rem - col * cn;
                k6_sm[e] = base[((int64_t)(row ? j1 : j) * S + cx_lo + col) * C + c0 + c];
            }
        __syncthreads();
        if (xs + (int)(threadIdx.x & ~63u) >= w) continue;  // a wave with no column left of the border only helps staging
        const float* p00 = k6_sm + (x0 - cx_lo) * cn;
        const float* p01 = k6_sm + (x1 - cx_lo) * cn;
        const float* p10 = p00 + per_row;
        const float* p11 = p01 + per_row;
        int c = 0;
        if (c0 == 0) {                                      // class 0 starts every running maximum (first max wins)
            const float top = lx0 * p00[0] + lx1 * p01[0];
            const float bot = lx0 * p10[0] + lx1 * p11[0];
#pragma unroll
            for (int r = 0; r < K6_R; ++r) best[r] = ly0[r] * top + ly1[r] * bot;
            c = 1;
        }
        for (; c < cn; ++c) {
            const float top = lx0 * p00[c] + lx1 * p01[c];
            const float bot = lx0 * p10[c] + lx1 * p11[c];
#pragma unroll
            for (int r = 0; r < K6_R; ++r) {
                const float v = ly0[r] * top + ly1[r] * bot;
                if (v > best[r]) { best[r] = v; arg[r] = c0 + c; }   // NaN-free inputs; first max wins
            }
        }
    }
    if (out && xs + (int)threadIdx.x < w) {
#pragma unroll
        for (int r = 0; r < K6_R; ++r)
            if (r0 + r < r1) out[(b * h + (r0 + r)) * (int64_t)w + x] = arg[r];
    }
    if (conf) {
        // counts go through LDS -- a plain histogram or a (pair, count) table, see above -- and reach the matrix as ONE global atomic per
        // distinct pair of the workgroup's <= 16 x 256 pixels: same-address global atomics from every wave would serialise at the L2
        // (uniform regions send most pixels to one bin)
        __syncthreads();                                    // the staging area is free now
        const int bins = G * P, hsize = 1 << hbits;
        int* hkey = reinterpret_cast<int*>(k6_sm);
        unsigned* hcnt = hbits ? reinterpret_cast<unsigned*>(k6_sm) + hsize : reinterpret_cast<unsigned*>(k6_sm);
        if (hbits) { for (int e = threadIdx.x; e < hsize; e += blockDim.x) { hkey[e] = -1; hcnt[e] = 0u; } }
        else for (int e = threadIdx.x; e < bins; e += blockDim.x) hcnt[e] = 0u;
        __syncthreads();
        const int lane = threadIdx.x & 63;
        auto count = [&](int key, unsigned n) {
            if (!hbits) { atomicAdd(&hcnt[key], n); return; }
            unsigned slot = ((unsigned)key * 2654435761u) >> (32 - hbits);
            for (int probe = 0; probe < 4; ++probe) {
                const int old = atomicCAS(&hkey[slot], -1, key);
                if (old == -1 || old == key) { atomicAdd(&hcnt[slot], n); return; }
                slot = (slot + 1) & (hsize - 1);
            }
            atomicAdd(&conf[key], (unsigned long long)n);
        };
#pragma unroll
        for (int r = 0; r < K6_R; ++r) {
            if (r0 + r >= r1) break;                        // block-uniform
            bool valid = xs + (int)threadIdx.x < w;
            const int64_t g = valid ? gt[(b * h + (r0 + r)) * (int64_t)w + x] : -1;
            valid = valid && !(has_ignore && g == ignore) && g >= 0 && g < G && arg[r] < P;
            const int key = valid ? (int)g * P + arg[r] : -1;
            // a wave's 64 neighbouring pixels of one row mostly share one or two pairs: up to three groups are counted by their
            // leaders, whatever is left lane by lane
            unsigned long long todo = __ballot(valid);
            bool noise = false;
#pragma unroll 1
            for (int round = 0; round < 8 && todo; ++round) {
                const int leader = __builtin_ctzll(todo);
                const int k0 = __builtin_amdgcn_readlane(key, leader);
                const unsigned long long same = __ballot(key == k0) & todo;
                // after three groups: a fourth leader that stands alone among more than 16 ungrouped lanes is noise -- as many pairs as
                // pixels, nothing for a table to collect; a fourth GROUP (a row that crosses several regions) keeps being grouped, so that
                // equal pairs never go to the matrix lane by lane (17 same-address global atomics: 0.171 -> 0.194 ms on rectangle masks)
                if (round >= 3 && hbits && __popcll(same) == 1 && __popcll(todo) > 16) { noise = true; break; }
                if (lane == leader) count(k0, (unsigned)__popcll(same));
                todo &= ~same;
            }
            if ((todo >> lane) & 1ull) {
                if (noise) atomicAdd(&conf[key], 1ull);
                else count(key, 1u);
            }
        }
        __syncthreads();
        if (hbits) {
            for (int e = threadIdx.x; e < hsize; e += blockDim.x)
                if (hkey[e] >= 0 && hcnt[e]) atomicAdd(&conf[hkey[e]], (unsigned long long)hcnt[e]);
        } else
            for (int e = threadIdx.x; e < bins; e += blockDim.x)
                if (hcnt[e]) atomicAdd(&conf[e], (unsigned long long)hcnt[e]);
    }
}

int hb_launch_upsample_argmax(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out,
                              hipStream_t s) {
    return hb_launch_upsample_argmax_confusion(label_hat, B, S, C, h, w, out, nullptr, 0, 0, 0, 0, nullptr, s);
}

int hb_launch_upsample_argmax_confusion(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out, const int64_t* gt,
                                        int num_gt, int num_pred, int64_t ignore, int has_ignore, unsigned long long* conf, hipStream_t s) {
    if (B == 0) return 0;
    if (S < 1 || C < 1 || h < 1 || w < 1) return hb_fail("hb_upsample_argmax: bad shape");
    if (!out && !conf) return hb_fail("hb_upsample_argmax_confusion: neither a class map nor a confusion matrix was asked for");
    if (conf && (!gt || num_gt < 1 || num_pred < 1 || (long long)num_gt * num_pred > 0x7FFFFFFFLL)) return hb_fail("hb_upsample_argmax_confusion: bad confusion-matrix arguments");
    const float sy = (float)S / (float)h, sx = (float)S / (float)w;
    // tallest band (same fp32 arithmetic as the kernel: this file is compiled with contraction off on both sides)
    int maxband = 1, run = 0, prev = -1;
    for (int yy = 0; yy < h; ++yy) {
        const float f = std::fmax(sy * ((float)yy + 0.5f) - 0.5f, 0.0f);
        const int y0 = std::min((int)std::floor(f), S - 1);
        run = y0 == prev ? run + 1 : 1;
        prev = y0;
        maxband = std::max(maxband, run);
    }
    // rows per chunk: the instantiation that computes the fewest padded rows over all bands (h = 14 S: bands of 14 -> R = 14)
    static const int r_opts[] = {16, 14, 12, 10, 8};
    int R = 16;
    long long best_cost = -1;
    for (int r : r_opts) {
        long long cost = 0;
        int n = 0, pv = -1;
        for (int yy = 0; yy <= h; ++yy) {
            int y0 = -2;
            if (yy < h) {
                const float f = std::fmax(sy * ((float)yy + 0.5f) - 0.5f, 0.0f);
                y0 = std::min((int)std::floor(f), S - 1);
            }
            if (y0 != pv) { cost += (long long)((n + r - 1) / r) * (r + 1); n = 0; pv = y0; }   // + 1: per-chunk overhead (top / bot, staging)
            ++n;
        }
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; R = r; }
    }
    const int chunks = (maxband + R - 1) / R;
    if ((long long)S * chunks > 65535 || B > 65535) return hb_fail("hb_upsample_argmax: grid too large");
    // whole waves of 64 columns, dealt evenly to blocks of at most four (w = 518: 9 waves -> 3 blocks of 3, none idle)
    const int waves = (w + 63) / 64, blocks_x = (waves + 3) / 4, bw = (waves + blocks_x - 1) / blocks_x * 64;
    // widest column window of a block, and the classes per LDS pass that keep it within 64 KiB
    const int ncols = std::min(S, (int)std::ceil((double)bw * S / w) + 2);
    const int cmax = std::max(1, std::min(C, (64 * 1024) / (2 * ncols * 4)));
    // the confusion counts' home in the (then free) staging area: a plain histogram where G x P bins fit 16 KiB, else a hash table of
    // 256 or 512 (pair, count) entries
    const size_t stage_bytes = (size_t)2 * ncols * cmax * 4;
    int hbits = 0;
    size_t table_bytes = 0;
    if (conf) {
        if ((size_t)num_gt * num_pred * 4 <= K6_HIST_BYTES) table_bytes = (size_t)num_gt * num_pred * 4;
        else {
            hbits = 8;
            while (hbits < 9 && ((size_t)8 << (hbits + 1)) <= stage_bytes) ++hbits;     // (2048 entries: their per-block init + flush cost more than they catch, 0.175 -> 0.200 ms at cfg-3)
            table_bytes = (size_t)8 << hbits;
        }
    }
    const size_t lds = std::max<size_t>(stage_bytes, table_bytes);
    const dim3 grid((unsigned)blocks_x, (unsigned)(S * chunks), (unsigned)B);
#define K6_LAUNCH(RR) upsample_argmax_kernel<RR><<<grid, dim3(bw), lds, s>>>(label_hat, S, C, h, w, sy, sx, chunks, cmax, out, gt, num_gt, num_pred, ignore, has_ignore, conf, hbits)
    switch (R) {
        case 8: K6_LAUNCH(8); break;
        case 10: K6_LAUNCH(10); break;
        case 12: K6_LAUNCH(12); break;
        case 14: K6_LAUNCH(14); break;
        default: K6_LAUNCH(16); break;
    }
#undef K6_LAUNCH
    HB_HIP(hipGetLastError());
    return 0;
}

// K6w -- sliding-window evaluation (BASELINE cfg-5: 1024 x 2048 Cityscapes frames evaluated through input_size
// windows; the reference has no tiler).  A window's label_hat[B, S*S, C] is upsampled exactly as
// in K6 (the reference's per-image F.interpolate, hbird_eval.py:235-243) to win_h x win_w and ADDED into the frame
// accumulator acc[B, H, W, C] (channels last) at (y0, x0); the frame's prediction is argmax_c acc (dividing by the
// per-pixel window count would not change it).  One thread per (pixel, class) with the class fastest: accumulator
// traffic and the 4 source tokens are contiguous across a wavefront.  HBM-bound: 8 B per accumulator element.
__global__ __launch_bounds__(256) void upsample_accumulate_kernel(const float* __restrict__ lh, int S, int C, int win_h,
                                                                  int win_w, float sy, float sx, float* __restrict__ acc,
                                                                  int H, int W, int y0f, int x0f) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (e >= (int64_t)win_h * win_w * C) return;
    const int c = (int)(e % C);
    const int pix = (int)(e / C);
    const int x = pix % win_w, yy = pix / win_w;
        float fy = fmaxf(sy * ((float)yy + 0.5f) - 0.5f, 0.0f);
    float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.0f);
    int y0 = min((int)floorf(fy), S - 1), x0 = min((int)floorf(fx), S - 1);
    int y1 = min(y0 + 1, S - 1), x1 = min(x0 + 1, S - 1);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* base = lh + b * (int64_t)S * S * C + c;
    const float v00 = base[((int64_t)y0 * S + x0) * C], v01 = base[((int64_t)y0 * S + x1) * C];
    const float v10 = base[((int64_t)y1 * S + x0) * C], v11 = base[((int64_t)y1 * S + x1) * C];
    const float top = lx0 * v00 + lx1 * v01;
    const float bot = lx0 * v10 + lx1 * v11;
    const float v = ly0 * top + ly1 * bot;
    float* a = acc + ((b * H + (y0f + yy)) * (int64_t)W + (x0f + x)) * C + c;
    *a = *a + v;
}

int hb_launch_upsample_accumulate(const float* label_hat, int64_t B, int S, int C, int win_h, int win_w, float* acc, int H,
                                  int W, int y0, int x0, hipStream_t s) {
    if (B == 0) return 0;
    if (y0 < 0 || x0 < 0 || y0 + win_h > H || x0 + win_w > W) return hb_fail("hb_upsample_accumulate: window outside the frame");
    const float sy = (float)S / (float)win_h, sx = (float)S / (float)win_w;
    const int64_t n = (int64_t)win_h * win_w * C;
    upsample_accumulate_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, s>>>(label_hat, S, C, win_h, win_w, sy,
                                                                                                  sx, acc, H, W, y0, x0);
    HB_HIP(hipGetLastError());
    return 0;
}

// argmax over the last (class) dimension of acc[n, C]; first maximum wins (lowest class), as torch.argmax on the
// reference's [B, C, h, w] tensor.  One wavefront per 64 / C' pixels would coalesce better; at C <= 151 the rows are
// 76..604 B and the kernel moves 4 C + 8 bytes per pixel once -- not worth more.
__global__ __launch_bounds__(256) void argmax_channels_kernel(const float* __restrict__ acc, int64_t n, int C,
                                                              int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* a = acc + i * C;
    float best = a[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float v = a[c];
        if (v > best) { best = v; bi = c; }
    }
    out[i] = bi;
}

int hb_launch_argmax_channels(const float* acc, int64_t n, int C, int64_t* out, hipStream_t s) {
    if (n == 0) return 0;
    if (C < 1) return hb_fail("hb_argmax_channels: C must be >= 1");
    argmax_channels_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(acc, n, C, out);
    HB_HIP(hipGetLastError());
    return 0;
}

// K7 -- reference hbird/utils/eval_metrics.py:73-104 (PredsmIoU.update): drop gt == ignore_index, drop
// out-of-range pairs, conf[gt, pred] += 1.  Block-private LDS histogram (u32) flushed with global
// 64-bit atomics; falls back to direct global atomics when num_gt*num_pred does not fit LDS.
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ gt, const int64_t* __restrict__ pred,
                                                        int64_t n, int G, int P, int64_t ignore, int has_ignore,
                                                        int use_lds, unsigned long long* __restrict__ conf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* hist = reinterpret_cast<unsigned*>(smem);
    const int bins = G * P;
    if (use_lds) {
        for (int e = threadIdx.x; e < bins; e += 256) hist[e] = 0;
        __syncthreads();
    }
    // a wave's 64 neighbouring pixels mostly share one or two (gt, pred) pairs (piecewise-constant maps), and 64 per-pixel atomics on
    // one bin serialise: up to two groups of equal pairs are counted by their leaders, whatever is left (noise-like maps: conflict-free
    // bins anyway) lane by lane
    const int lane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + lane;
        bool valid = i < n;
        const int64_t g = valid ? gt[i] : -1, p = valid ? pred[i] : -1;
        valid = valid && !(has_ignore && g == ignore) && g >= 0 && g < G && p >= 0 && p < P;
        const int key = valid ? (int)(g * P + p) : -1;
        unsigned long long todo = __ballot(valid);
#pragma unroll 1
        for (int round = 0; round < 2 && todo; ++round) {
            const int leader = __builtin_ctzll(todo);
            const int k0 = __builtin_amdgcn_readlane(key, leader);
            const unsigned long long same = __ballot(key == k0) & todo;
            if (lane == leader) {
                if (use_lds) atomicAdd(&hist[k0], (unsigned)__popcll(same));
                else atomicAdd(&conf[k0], (unsigned long long)__popcll(same));
            }
            todo &= ~same;
        }
        if ((todo >> lane) & 1ull) {
            if (use_lds) atomicAdd(&hist[key], 1u);
            else atomicAdd(&conf[key], 1ull);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int e = threadIdx.x; e < bins; e += 256)
            if (hist[e]) atomicAdd(&conf[e], (unsigned long long)hist[e]);
    }
}

int hb_launch_confusion(const int64_t* gt, const int64_t* pred, int64_t n, int num_gt, int num_pred, int64_t ignore,
                        int has_ignore, unsigned long long* conf, hipStream_t s) {
    if (n == 0) return 0;
    if ((long long)num_gt * num_pred > 0x7FFFFFFFLL) return hb_fail("hb_confusion_update: too many classes");
    const size_t bins = (size_t)num_gt * num_pred;
    const int use_lds = bins * 4 <= 120 * 1024;
    const size_t sh = use_lds ? bins * 4 : 0;
    if (hb_ensure_dyn_lds((const void*)confusion_kernel, 120 * 1024)) return -1;   // per (kernel, device)
    int64_t blocks = std::min<int64_t>((n + 255) / 256, 2048);
    confusion_kernel<<<dim3((unsigned)blocks), dim3(256), sh, s>>>(gt, pred, n, num_gt, num_pred, ignore, has_ignore, use_lds, conf);
    HB_HIP(hipGetLastError());
    return 0;
}
