// Micro-benchmark: fp32 MFMA rate with interleaved ds_read_b128 fillers, 1 vs 2 waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE __builtin_amdgcn_sched_barrier(0);

// NT = accumulator tiles per wave (8 -> 128 regs, 16 -> 256 regs); NB = query fragments (1 or 2)
// per "stage": NT*NB*4 MFMAs, NT/NB... reads: NT_A + NB fragment reads of 16 B per lane
template <int NA, int NB, bool READS, bool BARRIER, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x4* L = reinterpret_cast<f32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) L[i] = f32x4{1.f, 0.5f, 0.25f, 2.f};
    __syncthreads();
    f32x16 acc[NA * NB];
    for (int t = 0; t < NA * NB; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 fa[NA], fb[NB], ga[NA], gb[NB];
    for (int t = 0; t < NA; ++t) { fa[t] = L[t * 64 + lane]; ga[t] = fa[t]; }
    for (int t = 0; t < NB; ++t) { fb[t] = L[(NA + t) * 64 + lane]; gb[t] = fb[t]; }
    for (int it = 0; it < iters; ++it) {
        const f32x4* P = L + (it & 7) * 128 + lane;
        if (BARRIER) __syncthreads();
        // stage A: MFMAs on fa/fb, prefetch ga/gb
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    FENCE
                    acc[a * NB + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s], fb[b][s], acc[a * NB + b], 0, 0, 0);
                    FENCE
                    if (READS && s == 0 && b == 0) ga[a] = P[a * 64];
                    if (READS && s == 1 && a < NB && b == 0) gb[a] = P[(NA + a) * 64];
                }
        // stage B: MFMAs on ga/gb, prefetch fa/fb
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    FENCE
                    acc[a * NB + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[a][s], gb[b][s], acc[a * NB + b], 0, 0, 0);
                    FENCE
                    if (READS && s == 0 && b == 0) fa[a] = P[a * 64 + 32];
                    if (READS && s == 1 && a < NB && b == 0) fb[a] = P[(NA + a) * 64 + 32];
                }
    }
    float s = 0.f;
    for (int t = 0; t < NA * NB; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NA, int NB, bool READS, bool BARRIER, int THREADS>
double run(int iters) {
    const int threads = THREADS;
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto fn = k<NA, NB, READS, BARRIER, THREADS>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    fn<<<256, threads, 65536>>>(out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fn<<<256, threads, 65536>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * (threads / 64) * (double)iters * 2 * 4 * NA * NB * 4096.0;
    hipFree(out);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        printf("2w/SIMD 8x1 acc  noreads nobar : %.1f TF\n", run<8, 1, false, false, 512>(10000));
        printf("2w/SIMD 8x1 acc  reads   nobar : %.1f TF\n", run<8, 1, true, false, 512>(10000));
        printf("2w/SIMD 8x1 acc  reads   bar   : %.1f TF\n", run<8, 1, true, true, 512>(10000));
        printf("2w/SIMD 4x2 acc  reads   bar   : %.1f TF\n", run<4, 2, true, true, 512>(10000));
        printf("1w/SIMD 8x2 acc  noreads nobar : %.1f TF\n", run<8, 2, false, false, 256>(10000));
        printf("1w/SIMD 8x2 acc  reads   nobar : %.1f TF\n", run<8, 2, true, false, 256>(10000));
        printf("1w/SIMD 8x2 acc  reads   bar   : %.1f TF\n", run<8, 2, true, true, 256>(10000));
    }
    return 0;
}
