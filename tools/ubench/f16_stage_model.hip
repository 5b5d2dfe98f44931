// Micro-benchmark: the stage loop of the fp16 candidate kernel (csrc/hbird_knn_f16.hip) as a MODEL with every operand RANDOM, so
// that what each kind of traffic costs can be read under the power cap without the confound of the kernel's own ablation builds
// (an ablated stream leaves constant operands behind, which lowers the matrix pipe's power and raises the clock by itself).
//
// Workgroup tile = (NW waves) x (QT x 32 queries per wave) x 256 bank rows, one k16 "group" = QT x 8 MFMAs (32x32x16) per wave:
//   R  bank fragments re-read from LDS for every MFMA row tile (ds_read_b128, ring of 8 x 16 KiB, random contents)
//   Q  query fragments straight from global memory into registers (QT x 1 KiB per wave and group, a per-workgroup 384 KiB "query
//      tile" re-read every 48 groups: L2 / Infinity-Cache resident like the real one)
//   C  LDS-DMA copies of the bank stream (8 KiB per group and CU = 8 / NW pieces per wave; every workgroup reads the SAME stream,
//      so it is L2-resident like a clustered search)
//   B  one s_barrier per k32 stage
// SHAPE 0 = v_mfma_f32_32x32x16_f16, 1 = the same FLOPs as pairs of v_mfma_f32_16x16x32_f16.
// Variants: NW = 8, QT = 1 (the shipped tile: 2 waves per SIMD, 128 accumulator registers), NW = 4, QT = 2 (one wave per SIMD,
// 256 accumulators), NW = 4, QT = 3 (384 accumulators: VERDICT r3 item 1b).
// Prints ms, TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime) and the matrix pipe's busy share.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
#define FENCE __builtin_amdgcn_sched_barrier(0);

template <int NW, int QT, int SHAPE, bool R, bool Q, bool C, bool B, int DEPTH, int STREAM = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void model(const char* __restrict__ bank, const char* __restrict__ qsrc, float* out,
                                                          unsigned long long* clk, int groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane_off = lane * 16u;
    constexpr int CPW = 8 / NW;                       // LDS-DMA pieces per wave and group
    f32x16 acc[QT][8];
    f32x4 acc4[SHAPE ? QT * 32 : 1];
    for (int j = 0; j < QT; ++j) for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
    if (SHAPE) for (int i = 0; i < QT * 32; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    // ring prefill: 128 KiB of random fragments
    for (int i = threadIdx.x; i < 8192; i += NW * 64) reinterpret_cast<f16x8*>(smem)[i] = reinterpret_cast<const f16x8*>(bank)[i];
    __syncthreads();
    f16x8 fa[8], bq[DEPTH][QT];
    const char* qw = qsrc + ((size_t)blockIdx.x * NW + w) * 49152 * QT;   // this wave's share of the workgroup's query tile
    for (int t = 0; t < 8; ++t) fa[t] = reinterpret_cast<const f16x8*>(smem)[t * 128 + lane];
    for (int d = 0; d < DEPTH; ++d) for (int j = 0; j < QT; ++j) bq[d][j] = reinterpret_cast<const f16x8*>(qw)[(d * QT + j) * 64 + lane];
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    int slot = 0;
    size_t boff = 0; unsigned qoff = 0;
    // STREAM 0: every workgroup copies the same 64 MiB (L2-resident, like a clustered search at its best); 1: every workgroup its own
    // 48 MiB region of a 12 GiB buffer (HBM: an unclustered search); 2: eight neighbouring workgroups share a region (8 x 1 clusters)
    constexpr size_t REGION = (size_t)48 << 20;
    // (blocks b, b + 8, ... sit on one XCD: a cluster = eight such neighbours)
    const char* const bsrc = STREAM == 0 ? bank : bank + (size_t)(STREAM == 1 ? blockIdx.x : STREAM == 2 ? (blockIdx.x & 7) * 4 + (blockIdx.x >> 6)
                                                                  : STREAM == 3 ? (blockIdx.x & 7) * 2 + (blockIdx.x >> 7) : (blockIdx.x & 7)) * REGION;   // 3: sixteen share, 4: an XCD's 32
#define MM(J, T, BQ)                                                                                                          \
    if (SHAPE == 0) acc[J][T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[T], BQ, acc[J][T], 0, 0, 0);                          \
    else { acc4[(J) * 32 + 4 * (T)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[T], BQ, acc4[(J) * 32 + 4 * (T)], 0, 0, 0);     \
           acc4[(J) * 32 + 4 * (T) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(T) ^ 1], BQ, acc4[(J) * 32 + 4 * (T) + 1], 0, 0, 0); }
    // one group with static register-ring index U
#define GROUP(U)                                                                                                              \
    {                                                                                                                         \
        const f16x8* A = reinterpret_cast<const f16x8*>(smem + slot * 16384) + lane;                                          \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                                       \
            _Pragma("unroll") for (int j = 0; j < QT; ++j) { FENCE MM(j, t, bq[U][j]) FENCE }                                  \
            if (R) fa[t] = A[((t * 2 + ((U) & 1)) * 64)];                                                                      \
            if (C && t >= 2 && t < 2 + CPW)                                                                                   \
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(bsrc + boff + (size_t)(w * CPW + t - 2) * 1024 + lane_off),     \
                                                 (lds_void*)(smem + ((slot + 4) & 7) * 16384 + (w * CPW + t - 2) * 1024 + ((U) & 1) * 8192), 16, 0, 0); \
            if (Q && t == 0) {                                                                                                \
                _Pragma("unroll") for (int j = 0; j < QT; ++j)                                                                \
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[U][j]) : "v"(lane_off + qoff + j * 1024u), "s"(qw) : "memory"); \
            }                                                                                                                 \
        }                                                                                                                     \
        FENCE                                                                                                                 \
        boff += 8192; if (boff >= (STREAM == 0 ? (size_t)(64 << 20) : REGION)) boff = 0;                                                                      \
        qoff += QT * 1024u; if (qoff >= 49152u * QT) qoff = 0;                                                                \
        if ((U) & 1) {                                                                                                        \
            /* everything but the newest (DEPTH - 1) groups' requests has landed */                                            \
            if (Q || C) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 2) * ((Q ? QT : 0) + (C ? CPW : 0))) : "memory"); } \
            if (B) __builtin_amdgcn_s_barrier();                                                                              \
            slot = (slot + 1) & 7;                                                                                            \
        }                                                                                                                     \
    }
    for (int g = 0; g < groups; g += DEPTH) {
        GROUP(0) GROUP(1)
        if constexpr (DEPTH > 2) { GROUP(2) GROUP(3) }
        if constexpr (DEPTH > 4) { GROUP(4) GROUP(5) GROUP(6) GROUP(7) }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
    for (int j = 0; j < QT; ++j) for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[j][t][r];
    if (SHAPE) for (int i = 0; i < QT * 32; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    for (int d = 0; d < DEPTH; ++d) for (int j = 0; j < QT; ++j) s += (float)bq[d][j][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[lane];
}

static const char* g_bank; static const char* g_q; static float* g_out; static unsigned long long* g_clk;

template <int NW, int QT, int SHAPE, bool R, bool Q, bool C, bool B, int DEPTH, int STREAM = 0>
void run(const char* name, int groups) {
    auto fn = model<NW, QT, SHAPE, R, Q, C, B, DEPTH, STREAM>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    fn<<<256, NW * 64, 131072>>>(g_bank, g_q, g_out, g_clk, groups / 8);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    fn<<<256, NW * 64, 131072>>>(g_bank, g_q, g_out, g_clk, groups);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), g_clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int b = 0; b < 256; ++b) { ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); cyc.push_back((double)h[2 * b]); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double mfma_cycles = (double)groups * QT * 8 * 32 * (NW / 4);      // per SIMD
    const double tf = 256.0 * NW * (double)groups * QT * 8 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-58s %8.2f ms %7.1f TFLOP/s (%.3f)  clock %.3f GHz  pipe busy %.3f%s\n", name, ms, tf, tf / 2516.6, ghz[128], mfma_cycles / cyc[128],
           err == hipSuccess ? "" : "  LAUNCH FAILED");
}

__global__ void fill_kernel(_Float16* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + 12345u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((x & 0xFFFF) / 65535.0f - 0.5f) * 0.2f);
    }
}

int main(int argc, char** argv) {
    const bool streams_only = argc > 1;
    const size_t bank_bytes = (size_t)64 << 20, q_bytes = (size_t)256 * 8 * 49152 * 3;
    std::vector<_Float16> h((bank_bytes + q_bytes) / 2);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    char* buf; hipMalloc(&buf, bank_bytes + q_bytes); hipMemcpy(buf, h.data(), bank_bytes + q_bytes, hipMemcpyHostToDevice);
    g_bank = buf; g_q = buf + bank_bytes;
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_clk, 512 * 8);
    const int G = 48 * 4000;   // groups per workgroup: 4,000 bank tiles of 24 k32 stages
    if (streams_only) {   // `f16_stage_model streams`: the full loop on a bank stream that comes from L2 / from HBM per workgroup / from HBM per cluster of 8
        char* big; const size_t big_bytes = (size_t)256 * ((size_t)48 << 20);
        if (hipMalloc(&big, big_bytes) != hipSuccess) { printf("no memory for the 12 GiB stream buffer\n"); return 1; }
        fill_kernel<<<4096, 256>>>((_Float16*)big, big_bytes / 2); hipDeviceSynchronize();
        const char* shared = g_bank;
        for (int rep = 0; rep < 2; ++rep) {
            g_bank = shared; run<8, 1, 0, true, true, true, true, 8, 0>("full loop, bank stream shared by all workgroups (L2)", G);
            g_bank = big;    run<8, 1, 0, true, true, true, true, 8, 2>("full loop, one bank stream per 8 workgroups (HBM, 8 x 1 clusters)", G);
            g_bank = big;    run<8, 1, 0, true, true, true, true, 8, 3>("full loop, one bank stream per 16 workgroups (HBM)", G);
            g_bank = big;    run<8, 1, 0, true, true, true, true, 8, 4>("full loop, one bank stream per XCD (32 workgroups, HBM)", G);
            g_bank = big;    run<8, 1, 0, true, true, true, true, 8, 1>("full loop, one bank stream per workgroup (HBM, unclustered)", G);
            g_bank = big;    run<8, 1, 0, false, false, true, false, 8, 1>("MFMAs + copies only, one stream per workgroup", G);
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        printf("--- 8 waves x (32 q x 256 rows), 128 accumulators (the shipped tile) ---\n");
        run<8, 1, 0, false, false, false, false, 8>("32x32x16 bare", G);
        run<8, 1, 0, true, false, false, false, 8>("32x32x16 + fragment reads", G);
        run<8, 1, 0, false, true, false, false, 8>("32x32x16 + query loads", G);
        run<8, 1, 0, false, false, true, false, 8>("32x32x16 + copies", G);
        run<8, 1, 0, true, false, true, true, 8>("32x32x16 + reads + copies + barrier (queries stationary)", G);
        run<8, 1, 0, true, true, true, false, 8>("32x32x16 + reads + query loads + copies", G);
        run<8, 1, 0, true, true, true, true, 8>("32x32x16 + reads + query loads + copies + barrier", G);
        run<8, 1, 1, false, false, false, false, 8>("16x16x32 bare", G);
        run<8, 1, 1, true, true, true, true, 8>("16x16x32 + reads + query loads + copies + barrier", G);
        run<8, 1, 1, true, false, true, true, 8>("16x16x32 + reads + copies + barrier (queries stationary)", G);
        printf("--- 4 waves x (64 q x 256 rows), 256 accumulators ---\n");
        run<4, 2, 0, false, false, false, false, 4>("32x32x16 bare", G / 2);
        run<4, 2, 0, true, true, true, true, 4>("32x32x16 + reads + query loads + copies + barrier", G / 2);
        run<4, 2, 0, true, false, true, true, 4>("32x32x16 + reads + copies + barrier (queries stationary)", G / 2);
        printf("--- 4 waves x (96 q x 256 rows), 384 accumulators ---\n");
        run<4, 3, 0, false, false, false, false, 2>("32x32x16 bare", G / 3);
        run<4, 3, 0, true, true, true, true, 2>("32x32x16 + reads + query loads + copies + barrier", G / 3);
        run<4, 3, 0, true, false, true, true, 2>("32x32x16 + reads + copies + barrier (queries stationary)", G / 3);
    }
    return 0;
}
