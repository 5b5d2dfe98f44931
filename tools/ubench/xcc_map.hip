// Which XCD (XCC_ID hardware register) and CU does block b of a 256-block, one-block-per-CU launch run on?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
    extern __shared__ char smem[];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)xcc; out[2 * blockIdx.x + 1] = (int)hwid; smem[0] = 1; }
    // stay resident for a while so that all 256 blocks coexist
    long long t0 = clock64();
    while (clock64() - t0 < 2000000) {}
}
int main() {
    int* out; hipMalloc(&out, 256 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    k<<<256, 512, 140000>>>(out);
    hipDeviceSynchronize();
    int h[512]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < 256; ++b) {
        int xcc = h[2 * b] & 0xf, cu = (h[2 * b + 1] >> 8) & 0xf, se = (h[2 * b + 1] >> 13) & 0x7;
        if (b < 24 || b % 37 == 0) printf("block %3d: xcc %d se %d cu %d (raw %08x %08x)\n", b, xcc, se, cu, h[2 * b], h[2 * b + 1]);
        ok += (xcc == b % 8);
    }
    printf("blocks with xcc == b %% 8: %d / 256\n", ok);
    int cnt[16] = {0};
    for (int b = 0; b < 256; ++b) cnt[h[2 * b] & 0xf]++;
    for (int x = 0; x < 8; ++x) printf("xcc %d: %d blocks\n", x, cnt[x]);
    return 0;
}
