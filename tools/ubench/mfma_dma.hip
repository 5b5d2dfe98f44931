// Micro-benchmark: cost of LDS-DMA (global_load_lds_dwordx4) issue next to fp32 MFMA streams on gfx950:
//   mode 0: 4 compute waves (1/SIMD, 64 MFMAs per iteration), no DMA
//   mode 1: the same 4 waves also issue NDMA copies per iteration (interleaved between MFMAs)
//   mode 2: 4 compute waves + 4 producer waves (one per SIMD) that issue the copies; one barrier per iteration
//   mode 3: 4 compute waves + 1 producer wave issuing all copies
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
#define FENCE __builtin_amdgcn_sched_barrier(0);

template <int MODE, int THREADS, int NDMA>   // NDMA = copies per iteration per CU (16 in the real kernel per 256 MFMAs)
__global__ __launch_bounds__(THREADS) void k(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = 1.f + lane * 1e-3f, b = 0.5f;
    const float* g = src + (size_t)blockIdx.x * 65536 + lane * 4;
    if (w < 4) {
        for (int it = 0; it < iters; ++it) {
            if (MODE >= 2) __syncthreads();
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    FENCE
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    FENCE
                    if (MODE == 1 && s == 1 && t < NDMA / 4)
                        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(g + ((it * 16 + w * 4 + t) & 63) * 256), (lds_void*)(smem + (w * 4 + t) * 1024), 16, 0, 0);
                }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    } else {
        const int nprod = THREADS / 64 - 4;
        for (int it = 0; it < iters; ++it) {
            __syncthreads();
            for (int c = w - 4; c < NDMA; c += nprod)
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(g + ((it * 16 + c) & 63) * 256), (lds_void*)(smem + c * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[lane];
}

template <int MODE, int THREADS, int NDMA>
double run(int iters, const float* src) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto fn = k<MODE, THREADS, NDMA>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    fn<<<256, THREADS, 65536>>>(out, src, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fn<<<256, THREADS, 65536>>>(out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * 4 * (double)iters * 64 * 4096.0;
    hipFree(out);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    float* src; hipMalloc(&src, (size_t)256 * 65536 * 4); hipMemset(src, 0, (size_t)256 * 65536 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        printf("mode0 no DMA                         : %.1f TF\n", run<0, 256, 16>(10000, src));
        printf("mode1 compute waves issue 16 copies  : %.1f TF\n", run<1, 256, 16>(10000, src));
        printf("mode2 4 producer waves, 16 copies    : %.1f TF\n", run<2, 512, 16>(10000, src));
        printf("mode3 1 producer wave, 16 copies     : %.1f TF\n", run<3, 320, 16>(10000, src));
        printf("mode2 4 producer waves, no copies    : %.1f TF\n", run<2, 512, 0>(10000, src));
    }
    return 0;
}
