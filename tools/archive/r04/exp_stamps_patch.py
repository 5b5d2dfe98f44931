# (Applies to the tree of commit ccb971c: pool_compact has changed since -- the bisection compaction came out of these counts.)
# The temporary patch behind lib/abl/libhbird_hip_x_stamps.so (tools/exp_stamps.py): run from csrc/, build with
#   make varu UNIT=hbird_knn_f16 NAME=x_stamps EXTRA="-Wno-unused-variable", then restore hbird_knn_dev.h / hbird_knn_f16.hip with git checkout.
p='hbird_knn_dev.h'
s=open(p).read()
old='''    if (!bulk) {
        float q0v = 0.f, q1v = 0.f, q2v = 0.f, q3v = 0.f;
        int q0c = 0, q1c = 0, q2c = 0, q3c = 0, np = 0;
        HB_SCAN_TILE(0) HB_SCAN_TILE(1) HB_SCAN_TILE(2) HB_SCAN_TILE(3) HB_SCAN_TILE(4) HB_SCAN_TILE(5) HB_SCAN_TILE(6) HB_SCAN_TILE(7)
        if (__ballot(np != 0) == 0ull) return;
        if (__ballot(np > 4) == 0ull) {
            pool_drain<EMAX>(np, q0v, q1v, q2v, q3v, q0c, q1c, q2c, q3c, thr, pool_s, pool_i, qb, lane, k,
                             bt * HB_BT + 4u * (unsigned)(lane >> 5), klw, cnt);
            return;
        }'''
new='''    const unsigned long long xt0 = __builtin_readcyclecounter();
    if (lane == 0) atomicAdd(&x_stamps[0], 1ull);                 // epilogues
    if (bulk && lane == 0) atomicAdd(&x_stamps[1], 1ull);         // ... in bulk mode
    if (!bulk) {
        float q0v = 0.f, q1v = 0.f, q2v = 0.f, q3v = 0.f;
        int q0c = 0, q1c = 0, q2c = 0, q3c = 0, np = 0;
        HB_SCAN_TILE(0) HB_SCAN_TILE(1) HB_SCAN_TILE(2) HB_SCAN_TILE(3) HB_SCAN_TILE(4) HB_SCAN_TILE(5) HB_SCAN_TILE(6) HB_SCAN_TILE(7)
        const unsigned long long xt1 = __builtin_readcyclecounter();
        if (__ballot(np != 0) == 0ull) { if (lane == 0) atomicAdd(&x_stamps[2], xt1 - xt0); return; }      // cycles of scans without a survivor
        if (lane == 0) { atomicAdd(&x_stamps[3], xt1 - xt0); atomicAdd(&x_stamps[4], 1ull); }              // cycles / number of scans with survivors
        { const int tot = np; int sum = tot; for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o); if (lane == 0) atomicAdd(&x_stamps[5], (unsigned long long)sum); }   // survivors
        if (__ballot(np > 4) == 0ull) {
            const unsigned long long xt1b = __builtin_readcyclecounter();
            pool_drain<EMAX>(np, q0v, q1v, q2v, q3v, q0c, q1c, q2c, q3c, thr, pool_s, pool_i, qb, lane, k,
                             bt * HB_BT + 4u * (unsigned)(lane >> 5), klw, cnt);
            const unsigned long long xt2 = __builtin_readcyclecounter();
            if (lane == 0) atomicAdd(&x_stamps[6], xt2 - xt1b);                                            // cycles of drains
            return;
        }
        if (lane == 0) atomicAdd(&x_stamps[7], 1ull);                                                      // overflows'''
assert old in s
s=s.replace(old,new)
old='''    const int flagged = tile_epilogue<true, true, EMAX>(acc, t2, pool_s, pool_i, sc, qb, lane, k, bt, klw, cnt);
    thr = t2;'''
new='''    const unsigned long long xt3 = __builtin_readcyclecounter();
    const int flagged = tile_epilogue<true, true, EMAX>(acc, t2, pool_s, pool_i, sc, qb, lane, k, bt, klw, cnt);
    if (lane == 0) atomicAdd(&x_stamps[8], __builtin_readcyclecounter() - xt3);                             // cycles of quarter loops
    thr = t2;'''
assert old in s
s=s.replace(old,new)
s=s.replace("#pragma once","#pragma once\nstatic __device__ unsigned long long x_stamps[16];",1) if "#pragma once" in s else s
old="    const int E = cap >> 6;\n    float es[EMAX];"
new="    const unsigned long long xc0 = __builtin_readcyclecounter();\n    const int E = cap >> 6;\n    float es[EMAX];"
assert old in s
s=s.replace(old,new)
old="    return kth;\n}\n\n// Append the register queues of a wave"
new="    if (lane == 0) { atomicAdd(&x_stamps[9], __builtin_readcyclecounter() - xc0); atomicAdd(&x_stamps[10], 1ull); }\n    return kth;\n}\n\n// Append the register queues of a wave"
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
p='hbird_knn_f16.hip'
s=open(p).read()
s=s.replace("int hb_knn_f16_launch(const knn16_args& args, int grid, hipStream_t s) {",'''extern "C" int hb_x_read_stamps(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(x_stamps), 16 * 8);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(x_stamps), z, 16 * 8); }
    return 0;
}
int hb_knn_f16_launch(const knn16_args& args, int grid, hipStream_t s) {''')
s=s.replace("    cl_finish(cs, a.cl_stats, w == 0, lane);\n}\n\n// Exact re-rank","    cl_finish(cs, a.cl_stats, w == 0, lane);\n    if (lane == 0) atomicAdd(&x_stamps[11], __builtin_readcyclecounter() - x_k0);\n}\n\n// Exact re-rank")
s=s.replace("    extern __shared__ __attribute__((aligned(16))) char smem[];\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int h = lane >> 5;\n    float* sc = reinterpret_cast<float*>(smem + F2_SCRATCH) + w * 256;","    extern __shared__ __attribute__((aligned(16))) char smem[];\n    const unsigned long long x_k0 = __builtin_readcyclecounter();\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int h = lane >> 5;\n    float* sc = reinterpret_cast<float*>(smem + F2_SCRATCH) + w * 256;",1)
open(p,'w').write(s)
