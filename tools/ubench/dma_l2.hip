// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) ingest rate when the source is SHARED between workgroups, i.e.
// served by the XCD's L2 instead of the fabric: every workgroup (8 waves, one per CU) streams the same `shared_kb`
// window (advancing through a large buffer in lockstep-ish fashion) into a 64 KiB LDS ring.
//   mode 0: all 256 workgroups read the same stream           (each line is fetched once per XCD: 1/32 of the traffic)
//   mode 1: workgroups with the same blockIdx % 8 (= same XCD under round-robin dispatch) share a stream
//   mode 2: every workgroup has its own stream                 (no sharing; the dma_bw.hip case)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

__global__ __launch_bounds__(512) void k(const float* src, size_t floats_per_stream, int mode, int issuers, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stream = mode == 0 ? 0 : (mode == 1 ? blockIdx.x % 8 : blockIdx.x);
    const float* base = src + (size_t)stream * floats_per_stream + lane * 4;
    const size_t chunks = floats_per_stream / (8 * 2 * 256);   // per iteration the workgroup copies 16 KiB
    int slot = 0;
    if (w < issuers) {
        const int per = 16 / issuers;                            // 1 KiB copies per issuing wave per iteration
        for (size_t c = 0; c < chunks; ++c) {
            for (int j = 0; j < per; ++j) {
                const float* g = base + (c * 16 + w * per + j) * 256;
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)g, (lds_void*)(smem + slot * 16384 + (w * per + j) * 1024), 16, 0, 0);
            }
            if (per == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (per == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            slot = (slot + 1) & 3;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<float*>(smem)[lane];
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const size_t stream_bytes = (size_t)64 << 20;              // 64 MiB per stream
    float* src; hipMalloc(&src, stream_bytes * 256); hipMemset(src, 0, stream_bytes * 256);
    for (int issuers : {8, 4, 2})
        for (int mode = 0; mode < 3; ++mode) {
            k<<<256, 512, 65536>>>(src, stream_bytes / 4, mode, issuers, out);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<<<256, 512, 65536>>>(src, stream_bytes / 4, mode, issuers, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tb = (double)stream_bytes * 256 / (ms * 1e-3) / 1e12;
            printf("issuing waves %d, mode %d (%s): %.2f TB/s into LDS = %.1f GB/s per CU = %.1f B/clk/CU @2.4GHz\n", issuers, mode,
                   mode == 0 ? "one stream for all" : mode == 1 ? "one stream per XCD" : "private streams", tb, tb * 1e3 / 256,
                   tb * 1e12 / 256 / 2.4e9);
        }
    return 0;
}
