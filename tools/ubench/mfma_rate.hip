// Micro-benchmark: fp32 MFMA issue rate on gfx950 with 1 or 2 waves per SIMD (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
double run(int threads, int iters) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<256, threads>>>(out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<256, threads>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * (threads / 64) * (double)iters * 4 * NACC * 4096.0;
    hipFree(out);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        printf("NACC=8  1 wave/SIMD (256 thr): %.1f TF\n", run<8>(256, 20000));
        printf("NACC=8  2 waves/SIMD (512 thr): %.1f TF\n", run<8>(512, 20000));
        printf("NACC=4  2 waves/SIMD (512 thr): %.1f TF\n", run<4>(512, 40000));
        printf("NACC=1  2 waves/SIMD (512 thr): %.1f TF\n", run<1>(512, 80000));
        printf("NACC=1  1 wave/SIMD (256 thr): %.1f TF\n", run<1>(256, 80000));
    }
    return 0;
}
