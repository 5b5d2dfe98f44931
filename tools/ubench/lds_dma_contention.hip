// Micro-benchmark: do LDS-DMA writes (global_load_lds_dwordx4) and wave fragment reads (ds_read_b128) share the LDS port
// additively?  One workgroup of 8 waves per CU; per iteration the workgroup
//   mode 1: reads 72 KiB of fragments from LDS (8 waves x 9 x 1 KiB, like a k16 group of the fp16 kNN kernel)
//   mode 2: copies 16 KiB from an L2-resident buffer into an LDS ring (waves 0-3, 4 copies each)
//   mode 3: both in the same iteration
// and reports cycles per iteration per CU (s_memtime) -- if (3) ~ (1) + (2) the two streams serialise on the LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* src, int iters, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // L2-resident source: 1 MiB shared by every workgroup
    const float* base = src + lane * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4* A = reinterpret_cast<const f32x4*>(smem) + lane;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) {
            if (w < 4) {
                const int slot = it & 3;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_global_load_lds((gbl_cvoid*)(base + ((it * 16 + w * 4 + j) & 1023) * 256),
                                                     (lds_void*)(smem + 65536 + slot * 16384 + (w * 4 + j) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
        }
        if (MODE & 1) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const f32x4 v = A[((it + j) & 63) * 64];
                acc += v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

template <int MODE>
static void run(const float* src, float* out, long long* cyc, int iters, const char* what, double bytes_per_iter) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    k<MODE><<<256, 512, 131072>>>(src, iters, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, 512, 131072>>>(src, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_iter = ms * 1e6 / iters;
    printf("%-28s %7.1f ns/iter = %6.0f cycles @2.4 GHz; %5.1f B/clk/CU\n", what, ns_per_iter, ns_per_iter * 2.4,
           bytes_per_iter / (ns_per_iter * 2.4));
}

int main() {
    float* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    float* out; hipMalloc(&out, 64);
    long long* cyc; hipMalloc(&cyc, 256 * 8);
    const int iters = 200000;
    run<1>(src, out, cyc, iters, "fragment reads 72 KiB", 73728.0);
    run<2>(src, out, cyc, iters, "LDS-DMA copies 16 KiB", 16384.0);
    run<3>(src, out, cyc, iters, "both", 73728.0 + 16384.0);
    return 0;
}
