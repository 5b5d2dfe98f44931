// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) streaming rate per chip on gfx950, by footprint:
// every workgroup (8 waves, one per CU) streams its own slice of a buffer into a 64 KiB LDS ring, 3 stages ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__global__ __launch_bounds__(512) void k(const float* src, size_t floats_per_wg, int passes, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* base = src + (size_t)blockIdx.x * floats_per_wg + lane * 4;
    const size_t chunks = floats_per_wg / (8 * 2 * 256);   // per iteration each wave copies 2 KiB
    int slot = 0;
    for (int p = 0; p < passes; ++p)
        for (size_t c = 0; c < chunks; ++c) {
            const float* g = base + (c * 8 + w) * 512;
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)(smem + slot * 16384 + w * 2048), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_cvoid*)(g + 256), (lds_void*)(smem + slot * 16384 + w * 2048 + 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            slot = (slot + 1) & 3;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<float*>(smem)[lane];
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (size_t mb : {64, 128, 512, 4096}) {
        size_t bytes = mb << 20;
        float* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
        size_t per_wg = bytes / 4 / 256;
        int passes = (int)((size_t)16384 / mb) + 1;     // ~16 GB of traffic per measurement
        k<<<256, 512, 65536>>>(src, per_wg, 1, out);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<256, 512, 65536>>>(src, per_wg, passes, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("footprint %5zu MiB x %d passes: %.2f TB/s\n", mb, passes, (double)bytes * passes / (ms * 1e-3) / 1e12);
        hipFree(src);
    }
    return 0;
}
