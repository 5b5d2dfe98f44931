// Post-kNN kernels ("next" row f1 of SURVEY.md 8): bilinear upsample + argmax (K6) and the streaming
// confusion matrix (K7).
#include "hbird_internal.h"

// The interpolation arithmetic below is specified step by step in fp32 (as ATen and the oracle evaluate it), so
// contraction into FMAs is switched off for this file.  (The __f*_rn intrinsics do not help: they are inline
// functions defined under the default contraction mode and fuse after inlining.)
#pragma clang fp contract(off)

// K6 -- reference hbird_eval.py:235-243: label_hat[B, S*S, C] -> reshape [B,S,S,C] -> permute
// [B,C,S,S] -> F.interpolate(size=(h,w), mode="bilinear", align_corners=False) -> argmax(dim=1).
// Fused: the [B,C,h,w] fp32 tensor (2.6 GB at cfg-3) is never materialised.  One thread per output
// pixel; source index as ATen's area_pixel_compute_source_index: src = max(0, scale*(dst+0.5)-0.5),
// scale = S/h in fp32; value = ly0*(lx0*v00 + lx1*v01) + ly1*(lx0*v10 + lx1*v11); ties -> lowest class.
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* __restrict__ lh, int S, int C, int h, int w,
                                                              float sy, float sx, int64_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int yy = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (x >= w) return;
        float fy = fmaxf(sy * ((float)yy + 0.5f) - 0.5f, 0.0f);
    float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.0f);
    int y0 = min((int)floorf(fy), S - 1), x0 = min((int)floorf(fx), S - 1);
    int y1 = min(y0 + 1, S - 1), x1 = min(x0 + 1, S - 1);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* base = lh + b * (int64_t)S * S * C;
    const float* p00 = base + ((int64_t)y0 * S + x0) * C;
    const float* p01 = base + ((int64_t)y0 * S + x1) * C;
    const float* p10 = base + ((int64_t)y1 * S + x0) * C;
    const float* p11 = base + ((int64_t)y1 * S + x1) * C;
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < C; ++c) {
        const float top = lx0 * p00[c] + lx1 * p01[c];
        const float bot = lx0 * p10[c] + lx1 * p11[c];
        const float v = ly0 * top + ly1 * bot;
        if (v > best || c == 0) { best = v; bi = c; }   // NaN-free inputs; first max wins
    }
    out[(b * h + yy) * (int64_t)w + x] = bi;
}

int hb_launch_upsample_argmax(const float* label_hat, int64_t B, int S, int C, int h, int w, int64_t* out,
                              hipStream_t s) {
    if (B == 0) return 0;
    const float sy = (float)S / (float)h, sx = (float)S / (float)w;
    upsample_argmax_kernel<<<dim3((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)B), dim3(256), 0, s>>>(label_hat, S, C, h, w, sy, sx, out);
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
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = gt[i], p = pred[i];
        if (has_ignore && g == ignore) continue;
        if (g < 0 || g >= G || p < 0 || p >= P) continue;
        if (use_lds) atomicAdd(&hist[g * P + p], 1u);
        else atomicAdd(&conf[g * P + p], 1ull);
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
    const size_t bins = (size_t)num_gt * num_pred;
    const int use_lds = bins * 4 <= 120 * 1024;
    const size_t sh = use_lds ? bins * 4 : 0;
    if (hb_ensure_dyn_lds((const void*)confusion_kernel, 120 * 1024)) return -1;   // per (kernel, device)
    int64_t blocks = std::min<int64_t>((n + 255) / 256, 2048);
    confusion_kernel<<<dim3((unsigned)blocks), dim3(256), sh, s>>>(gt, pred, n, num_gt, num_pred, ignore, has_ignore, use_lds, conf);
    HB_HIP(hipGetLastError());
    return 0;
}
