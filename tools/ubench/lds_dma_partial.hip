// Which LDS words does a global_load_lds_dword touch when only some lanes are active?  (The cluster soft sync polls `cl`
// progress words with lanes < cl; statistics words placed right behind the landing zone once broke the sync.)
// hipcc --offload-arch=gfx950 -O2 lds_dma_partial.hip -o lds_dma_partial && ./lds_dma_partial
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_cvoid;
__global__ void k(const int* src, int* out, int active, int bytes) {
    __shared__ int lds[128];
    const int lane = threadIdx.x;
    lds[lane] = 0xAAAA0000 + lane; lds[64 + lane] = 0xAAAA0040 + lane;
    __syncthreads();
    if (lane < active) {
        if (bytes == 4) __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + lane * 32), (lds_void*)lds, 4, 0, 16);
        else __builtin_amdgcn_global_load_lds((gbl_cvoid*)(src + lane * 4), (lds_void*)lds, 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = lds[lane]; out[64 + lane] = lds[64 + lane];
}
int main() {
    int *src, *out; hipMalloc(&src, 64 * 32 * 4); hipMalloc(&out, 128 * 4);
    int h[64 * 32]; for (int i = 0; i < 64 * 32; ++i) h[i] = 0x1000 + i; hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int bytes : {4, 16}) for (int active : {4, 8, 32}) {
        k<<<1, 64>>>(src, out, active, bytes); int o[128]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        int changed = 0, last = -1; for (int i = 0; i < 128; ++i) if ((o[i] & 0xFFFF0000) != 0xAAAA0000) { ++changed; last = i; }
        printf("%2d B per lane, %2d active lanes: %3d LDS words changed, last changed word %3d; words 0..11:", bytes, active, changed, last);
        for (int i = 0; i < 12; ++i) printf(" %x", o[i]); printf("\n");
    }
    return 0;
}
