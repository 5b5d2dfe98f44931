// Bank-build kernels: K2 (patch soft labels) and K3 (bounded-memory patch sampling) of SURVEY.md 2.3.
#include "hbird_internal.h"
#include <map>
#include <mutex>
#include <utility>
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
// nonempty[b*SS + p] = 1/0; nz_count[b] = number of non-empty patches.
// Two kernels (until round 5 one block per image did both: 16 blocks on 256 CUs, 0.28 ms at the cfg-3 batch, with every thread walking
// its patches' classes through strided global reads): class frequencies by coalesced chunks of the image (LDS histogram, one global
// atomic per class and block), then one WAVE per patch with the lanes over the classes.  All sums are integers: any order, same result.
#define K3_CHUNK 8192      // label values per block of the frequency pass
__global__ __launch_bounds__(256) void patch_freq_kernel(const float* __restrict__ label, int SS, int C, int* __restrict__ freq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* fl = reinterpret_cast<int*>(smem);
    const int b = blockIdx.y;
    const int64_t n = (int64_t)SS * C, e0 = (int64_t)blockIdx.x * K3_CHUNK, e1 = e0 + K3_CHUNK < n ? e0 + K3_CHUNK : n;
    const float* lb = label + (int64_t)b * n;
    for (int c = threadIdx.x; c < C; c += 256) fl[c] = 0;
    __syncthreads();
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256)
        if (lb[e] > 0.0f) atomicAdd(&fl[(int)(e % C)], 1);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        if (fl[c]) atomicAdd(&freq[(int64_t)b * C + c], fl[c]);
}
#define K3_PATCHES 64      // patches per block of the score pass (16 per wave): ONE atomic per block on the image's counter -- the counters of
                           // a batch share a cache line, and 21,904 same-line atomics (one per patch) took 250 us at the cfg-3 batch
__global__ __launch_bounds__(256) void patch_scores_kernel(const float* __restrict__ label, int SS, int C, const int* __restrict__ freq,
                                                           float* __restrict__ scores, int* __restrict__ nonempty,
                                                           int* __restrict__ nz_count) {
    __shared__ int s_nz[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.y;
    const int* fb = freq + (int64_t)b * C;
    int mine = 0;
    for (int i = 0; i < K3_PATCHES / 4; ++i) {
        const int p = blockIdx.x * K3_PATCHES + i * 4 + wv;
        if (p >= SS) break;                                // wave-uniform
        const float* lp = label + ((int64_t)b * SS + p) * C;
        int sum = 0, any = 0;
        for (int c = lane; c < C; c += 64)
            if (lp[c] > 0.0f) { sum += fb[c]; any = 1; }
        for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); any |= __shfl_xor(any, o); }
        if (lane == 0) {
            scores[(int64_t)b * SS + p] = any ? (float)sum : 1e6f;
            nonempty[(int64_t)b * SS + p] = any;
        }
        mine += any;
    }
    if (lane == 0) s_nz[wv] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
        if (n) atomicAdd(&nz_count[b], n);
    }
}

// K3b -- reference hbird_eval.py:497-511: multiply the scores of the non-empty patches by the uniform
// noise r (drawn by the HOST from torch's global CPU generator, consumed in image order then patch
// order) and select the K smallest per image, ascending (ties: lower patch index).
// r_off[b] = offset of image b's first noise value.  Rank by counting; blockIdx.x = a run of 256 patches to rank, every block
// rebuilds the image's noisy scores in LDS (SS values) -- until round 5 one block per image ranked all of them after a serial prefix
// scan by one thread (0.23 ms at the cfg-3 batch).
__global__ __launch_bounds__(256) void patch_select_kernel(const float* __restrict__ scores,
                                                           const int* __restrict__ nonempty,
                                                           const float* __restrict__ r, const int64_t* __restrict__ r_off,
                                                           int SS, int K, int64_t* __restrict__ out_idx,
                                                           float* __restrict__ out_scores) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sv = reinterpret_cast<float*>(smem);
    __shared__ int wave_sum[4];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // position of each non-empty patch among the non-empty patches of this image: a block-wide prefix count, 256 patches per round
    int run = 0;
    for (int base = 0; base < SS; base += 256) {
        const int p = base + threadIdx.x;
        const int ne = p < SS ? nonempty[(int64_t)b * SS + p] : 0;
        const unsigned long long m = __ballot(ne != 0);
        if (lane == 0) wave_sum[wv] = __popcll(m);
        __syncthreads();
        int before = run, total = 0;
        for (int w2 = 0; w2 < 4; ++w2) { if (w2 < wv) before += wave_sum[w2]; total += wave_sum[w2]; }
        if (p < SS) {
            float s = scores[(int64_t)b * SS + p];
            if (ne) s = s * r[r_off[b] + before + __popcll(m & ((1ull << lane) - 1ull))];
            sv[p] = s;
            if (out_scores && blockIdx.x == 0) out_scores[(int64_t)b * SS + p] = s;
        }
        run += total;
        __syncthreads();
    }
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= SS) return;
    const float s = sv[p];
    int rank = 0;
    for (int j = 0; j < SS; ++j) { const float sj = sv[j]; rank += (sj < s) || (sj == s && j < p); }
    if (rank < K) out_idx[(int64_t)b * K + rank] = p;
}

extern "C" int hb_patch_scores(const float* label, int64_t B, int SS, int C, float* scores, int* nonempty, int* nz_count,
                               void* stream) {
    if (B == 0) return 0;
    if (B > 65535) return hb_fail("hb_patch_scores: more than 65,535 images per call");
    if ((size_t)C * 4 > 60000) return hb_fail("hb_patch_scores: too many classes");
    hipStream_t s = (hipStream_t)stream;
    // [B, C] class frequencies: a workspace kept per (device, stream) -- work on one stream is ordered, so the buffer is free again when the
    // next call on that stream reaches it (hipMallocAsync / hipFreeAsync per call cost 90 us: 0.016 -> 0.106 ms at the cfg-2 batch)
    int* freq = nullptr;
    {
        static std::mutex mu;
        static std::map<std::pair<int, hipStream_t>, std::pair<int*, size_t>> ws;
        int dev = 0;
        HB_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        auto& e = ws[{dev, s}];
        const size_t need = (size_t)B * C * 4;
        if (e.second < need) {
            if (e.first) { HB_HIP(hipStreamSynchronize(s)); HB_HIP(hipFree(e.first)); e.first = nullptr; e.second = 0; }
            HB_HIP(hipMalloc((void**)&e.first, need));
            e.second = need;
        }
        freq = e.first;
    }
    HB_HIP(hipMemsetAsync(freq, 0, (size_t)B * C * 4, s));
    HB_HIP(hipMemsetAsync(nz_count, 0, (size_t)B * 4, s));
    const int64_t n = (int64_t)SS * C;
    patch_freq_kernel<<<dim3((unsigned)((n + K3_CHUNK - 1) / K3_CHUNK), (unsigned)B), dim3(256), (size_t)C * 4, s>>>(label, SS, C, freq);
    HB_HIP(hipGetLastError());
    patch_scores_kernel<<<dim3((unsigned)((SS + K3_PATCHES - 1) / K3_PATCHES), (unsigned)B), dim3(256), 0, s>>>(label, SS, C, freq, scores, nonempty, nz_count);
    HB_HIP(hipGetLastError());
    return 0;
}

extern "C" int hb_patch_select(const float* scores, const int* nonempty, const float* r, const int64_t* r_off, int64_t B,
                               int SS, int K, int64_t* out_idx, float* out_scores, void* stream) {
    if (B == 0) return 0;
    if (K > SS) return hb_fail("hb_patch_select: K exceeds the number of patches per image");
    if ((size_t)SS * 4 > 60000) return hb_fail("hb_patch_select: too many patches per image");
    if (B > 65535) return hb_fail("hb_patch_select: more than 65,535 images per call");
    patch_select_kernel<<<dim3((unsigned)((SS + 255) / 256), (unsigned)B), dim3(256), (size_t)SS * 4, (hipStream_t)stream>>>(scores, nonempty, r, r_off, SS, K, out_idx, out_scores);
    HB_HIP(hipGetLastError());
    return 0;
}
