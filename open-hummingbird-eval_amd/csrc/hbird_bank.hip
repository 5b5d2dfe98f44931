// Bank-build kernels: K2 (patch soft labels) and K3 (bounded-memory patch sampling) of SURVEY.md 2.3.
#include "hbird_internal.h"
#include <mutex>

// K2 -- reference hbird_eval.py:309-310 (`y[y == 255] = 0`, optional), 555-573 (`_patchify_gt`) and
// 319-320 (`F.one_hot(patches, C).float().mean(dim=3)`): per-patch class histogram / P, computed
// directly from the int64 mask without materialising the [B,S,S,P,C] one-hot tensor.
// One wave per patch, LDS histogram of C bins; out[b, i, j, c] = fl(count / P).
// Pixels whose class is outside [0, C) are not counted and raise *err (the reference's one_hot
// raises for them).
__global__ __launch_bounds__(256) void patch_label_hist_kernel(const int64_t* __restrict__ y, int64_t n_patches, int H,
                                                               int W, int ps, int C, int map255,
                                                               float* __restrict__ out, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* hist = reinterpret_cast<int*>(smem) + wv * C;
    const int64_t p = (int64_t)blockIdx.x * 4 + wv;
    if (p >= n_patches) return;   // wave-uniform
    const int Sw = W / ps, Sh = H / ps;
    const int64_t b = p / ((int64_t)Sh * Sw);
    const int pi = (int)((p / Sw) % Sh), pj = (int)(p % Sw);
    for (int c = lane; c < C; c += 64) hist[c] = 0;
    const int P = ps * ps;
    const int64_t* base = y + (b * H + (int64_t)pi * ps) * W + (int64_t)pj * ps;
    bool bad = false;
    for (int e = lane; e < P; e += 64) {
        const int u = e / ps, v = e % ps;
        int64_t c = base[(int64_t)u * W + v];
        if (map255 && c == 255) c = 0;
        if (c >= 0 && c < C) atomicAdd(&hist[(int)c], 1);
        else bad = true;
    }
    if (bad && err) *err = 1;
    const float Pf = (float)P;
    float* o = out + p * (int64_t)C;
    for (int c = lane; c < C; c += 64) o[c] = (float)hist[c] / Pf;
}

int hb_launch_patch_label_hist(const int64_t* y, int64_t B, int H, int W, int ps, int C, int map255, float* out,
                               hipStream_t s) {
    if (ps <= 0 || H % ps || W % ps) return hb_fail("hb_patch_label_hist: H and W must be multiples of the patch size");
    const int64_t np = B * (H / ps) * (W / ps);
    if (np == 0) return 0;
    if ((size_t)C * 16 > 60000) return hb_fail("hb_patch_label_hist: too many classes");
    // out-of-range classes: F.one_hot of the reference raises (hbird_eval.py:319), and so does this call -- the flag is
    // read back after the kernel (one stream synchronisation per training batch, as one_hot's own range check costs)
    // one persistent error word per device (a 4-byte hipMalloc each, never freed: nothing to leak on a failing call)
    static std::mutex mu;
    static int* words[64] = {nullptr};
    int dev = 0;
    HB_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return hb_fail("hb_patch_label_hist: device index out of range");
    // the mutex covers the whole clear + kernel + read-back: two calls on one device (other streams / threads) share the word, and one
    // must not clear or read the other's flag; the call synchronises its stream anyway, so nothing is lost by serialising it
    std::lock_guard<std::mutex> lock(mu);
    if (!words[dev]) HB_HIP(hipMalloc((void**)&words[dev], 4));
    int* err = words[dev];
    HB_HIP(hipMemsetAsync(err, 0, 4, s));
    patch_label_hist_kernel<<<dim3((unsigned)((np + 3) / 4)), dim3(256), (size_t)C * 16, s>>>(y, np, H, W, ps, C, map255, out, err);
    HB_HIP(hipGetLastError());
    int bad = 0;
    HB_HIP(hipMemcpyAsync(&bad, err, 4, hipMemcpyDeviceToHost, s));
    HB_HIP(hipStreamSynchronize(s));
    if (bad) return hb_fail("hb_patch_label_hist: class values must be in [0, " + std::to_string(C) + ") (out-of-range class in the mask)");
    return 0;
}

// K3a -- reference hbird_eval.py:471-493: presence[p,c] = class c occurs in patch p; class_freq[c] =
// number of patches of the image containing c; score[p] = sum_c presence[p,c] * class_freq[c];
// empty patches get the 1e6 sentinel.  Presence is read from the K2 soft labels (label > 0).
// One block per image.  nonempty[b*SS + p] = 1/0; nz_count[b] = number of non-empty patches.
__global__ __launch_bounds__(256) void patch_scores_kernel(const float* __restrict__ label, int SS, int C,
                                                           float* __restrict__ scores, int* __restrict__ nonempty,
                                                           int* __restrict__ nz_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* freq = reinterpret_cast<int*>(smem);
    __shared__ int s_nz;
    const int b = blockIdx.x;
    const float* lb = label + (int64_t)b * SS * C;
    for (int c = threadIdx.x; c < C; c += 256) freq[c] = 0;
    if (threadIdx.x == 0) s_nz = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < SS * C; e += 256)
        if (lb[e] > 0.0f) atomicAdd(&freq[e % C], 1);
    __syncthreads();
    for (int p = threadIdx.x; p < SS; p += 256) {
        int sum = 0;
        bool any = false;
        for (int c = 0; c < C; ++c)
            if (lb[(int64_t)p * C + c] > 0.0f) { sum += freq[c]; any = true; }
        scores[(int64_t)b * SS + p] = any ? (float)sum : 1e6f;
        nonempty[(int64_t)b * SS + p] = any ? 1 : 0;
        if (any) atomicAdd(&s_nz, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) nz_count[b] = s_nz;
}

// K3b -- reference hbird_eval.py:497-511: multiply the scores of the non-empty patches by the uniform
// noise r (drawn by the HOST from torch's global CPU generator, consumed in image order then patch
// order) and select the K smallest per image, ascending (ties: lower patch index).
// r_off[b] = offset of image b's first noise value.  One block per image, rank by counting.
__global__ __launch_bounds__(256) void patch_select_kernel(const float* __restrict__ scores,
                                                           const int* __restrict__ nonempty,
                                                           const float* __restrict__ r, const int64_t* __restrict__ r_off,
                                                           int SS, int K, int64_t* __restrict__ out_idx,
                                                           float* __restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sv = reinterpret_cast<float*>(smem);
    int* pos = reinterpret_cast<int*>(smem) + SS;
    const int b = blockIdx.x;
    // position of each non-empty patch among the non-empty patches of this image (serial scan by one
    // thread: SS <= a few thousand and this runs once per train batch)
    if (threadIdx.x == 0) {
        int run = 0;
        for (int p = 0; p < SS; ++p) { pos[p] = run; run += nonempty[(int64_t)b * SS + p]; }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < SS; p += 256) {
        float s = scores[(int64_t)b * SS + p];
        if (nonempty[(int64_t)b * SS + p]) s = s * r[r_off[b] + pos[p]];
        sv[p] = s;
        if (out_scores) out_scores[(int64_t)b * SS + p] = s;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < SS; p += 256) {
        const float s = sv[p];
        int rank = 0;
        for (int j = 0; j < SS; ++j) { const float sj = sv[j]; rank += (sj < s) || (sj == s && j < p); }
        if (rank < K) out_idx[(int64_t)b * K + rank] = p;
    }
}

extern "C" int hb_patch_scores(const float* label, int64_t B, int SS, int C, float* scores, int* nonempty, int* nz_count,
                               void* stream) {
    if (B == 0) return 0;
    patch_scores_kernel<<<dim3((unsigned)B), dim3(256), (size_t)C * 4, (hipStream_t)stream>>>(label, SS, C, scores, nonempty, nz_count);
    HB_HIP(hipGetLastError());
    return 0;
}

extern "C" int hb_patch_select(const float* scores, const int* nonempty, const float* r, const int64_t* r_off, int64_t B,
                               int SS, int K, int64_t* out_idx, float* out_scores, void* stream) {
    if (B == 0) return 0;
    if (K > SS) return hb_fail("hb_patch_select: K exceeds the number of patches per image");
    if ((size_t)SS * 8 > 60000) return hb_fail("hb_patch_select: too many patches per image");
    patch_select_kernel<<<dim3((unsigned)B), dim3(256), (size_t)SS * 8, (hipStream_t)stream>>>(scores, nonempty, r, r_off, SS, K, out_idx, out_scores);
    HB_HIP(hipGetLastError());
    return 0;
}
