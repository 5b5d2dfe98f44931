#!/usr/bin/env python3
"""A/B of two builds of libhbird_hip.so through the stable core of the C ABI only (create / reserve / add / search /
timing): usage ab_lib.py rows dim nq k lib1.so lib2.so ...  -- kernel ms (HIP events) per library, interleaved rounds."""
import ctypes, sys
import torch
M, D, nq, k = (int(x) for x in sys.argv[1:5])
libs = sys.argv[5:]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
g = torch.Generator(device=dev); g.manual_seed(3)
q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
handles = []
for path in libs:
    L = ctypes.CDLL(path)
    L.hb_index_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.hb_index_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.hb_index_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.hb_index_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.hb_index_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.hb_index_last_knn_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    L.hb_index_free.argtypes = [ctypes.c_void_p]
    L.hb_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    assert L.hb_index_create(D, 0, 0, ctypes.byref(h)) == 0, L.hb_last_error()
    assert L.hb_index_reserve(h, M) == 0
    gg = torch.Generator(device=dev); gg.manual_seed(1)
    for r in range(0, M, 500_000):
        n = min(500_000, M - r)
        rows = torch.randn((n, D), generator=gg, device=dev)
        assert L.hb_index_add(h, ctypes.c_void_p(rows.data_ptr()), n, 1, 1) == 0, L.hb_last_error()
    torch.cuda.synchronize()
    L.hb_index_set_timing(h, 1)
    handles.append((path, L, h))
res = {p: [] for p in libs}
outs = {}
for rnd in range(4):
    for path, L, h in handles:
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev); dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        assert L.hb_index_search(h, ctypes.c_void_p(q.data_ptr()), nq, k, 0, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()), 1) == 0, L.hb_last_error()
        torch.cuda.synchronize()
        ms = ctypes.c_double(); L.hb_index_last_knn_ms(h, ctypes.byref(ms))
        if rnd: res[path].append(round(ms.value, 1))
        outs[path] = (idx, dist)
ref = outs[libs[0]]
for p in libs:
    print(p.split("/")[-1], res[p], "same as first:", bool(torch.equal(outs[p][0], ref[0]) and torch.equal(outs[p][1], ref[1])), flush=True)
