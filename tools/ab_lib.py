#!/usr/bin/env python3
"""A/B of two builds of libhbird_hip.so through the stable core of the C ABI only (create / reserve / add / search /
timing): usage ab_lib.py rows dim nq k lib1.so lib2.so ...  -- kernel ms (HIP events) per library, interleaved rounds.
AB_FP16=1 in the environment: use_fp16 searches (hb_index_set_fp16); AB_METRIC=1: L2 instead of inner product; AB_WALL=1: also whole-search ms by HIP events."""
import ctypes, os, sys
import torch
M, D, nq, k = (int(x) for x in sys.argv[1:5])
libs = sys.argv[5:]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
g = torch.Generator(device=dev); g.manual_seed(3)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
handles = []
for path in libs:
    L = ctypes.CDLL(path.split("@")[0])
    L.hb_index_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.hb_index_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.hb_index_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.hb_index_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hb_index_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.hb_index_last_knn_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    L.hb_index_free.argtypes = [ctypes.c_void_p]
    L.hb_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    assert L.hb_index_create(D, int(os.environ.get("AB_METRIC", "0")), 0, ctypes.byref(h)) == 0, L.hb_last_error()
    assert L.hb_index_reserve(h, M) == 0
    gg = torch.Generator(device=dev); gg.manual_seed(1)
    for r in range(0, M, 500_000):
        n = min(500_000, M - r)
        rows = torch.randn((n, D), generator=gg, device=dev)
        assert L.hb_index_add(h, ctypes.c_void_p(rows.data_ptr()), n, 1, 1) == 0, L.hb_last_error()
    torch.cuda.synchronize()
    L.hb_index_set_timing(h, 1)
    if os.environ.get("AB_FP16"):
        L.hb_index_set_fp16.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert L.hb_index_set_fp16(h, 1) == 0
    handles.append((path, L, h))
res = {p: [] for p in libs}
ROUND1 = 2 if os.environ.get("AB_MS2") else 1
outs = {}
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for path, L, h in handles:
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        assert L.hb_index_search(h, ctypes.c_void_p(q.data_ptr()), nq, k, 0, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()), 1) == 0, L.hb_last_error()
        torch.cuda.synchronize()
        ms = ctypes.c_double(); L.hb_index_last_knn_ms(h, ctypes.byref(ms))
        if rnd: res[path].append(round(ms.value, ROUND1))
        outs[path] = (idx, dist)
if os.environ.get("AB_WALL"):     # whole searches by HIP events (20 per library), for changes outside the kNN kernel
    for path, L, h in handles:
        idx, dist = outs[path]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.hb_index_set_timing(h, 0)
        e0.record()
        for _ in range(20):
            assert L.hb_index_search(h, ctypes.c_void_p(q.data_ptr()), nq, k, 0, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()), 1) == 0
        e1.record(); torch.cuda.synchronize()
        res[path].append(("search ms", round(e0.elapsed_time(e1) / 20, 3)))
ref = outs[libs[0]]
for p in libs:
    print(p.split("/")[-1], res[p], "same as first:", bool(torch.equal(outs[p][0], ref[0]) and torch.equal(outs[p][1], ref[1])), flush=True)
