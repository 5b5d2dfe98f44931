"""Round-2 boundary work on the GPU: packed cross-shard merge, one-process multi-index plugin (shards / replicas),
error behaviour of the bank-build kernels, the bench launcher over two ranks, and RCCL with a single rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_inputs as gi
import oracle
from hbird_mi import _lib, dist as hdist, ops
from hbird_mi.nn.search_hip import HipFlatIndex, NearestNeighborSearchHIP, merge_topk, merge_topk_packed

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_packed_merge_equals_array_merge(cuda_device, metric):
    """hb_merge_topk_packed reads the all-gather's receive buffer in place: same result as hb_merge_topk on
    [parts, nq, k] arrays, for searches that wrote their lists straight into the packed send buffers."""
    M, D, nq, k, parts = 30_000, 48, 333, 30, 3        # nq * k odd multiples: exercises the 16-byte padding
    bank = gi.unit_bank(M, D, seed=1); bank[25_000] = bank[3]
    q = torch.from_numpy(gi.vit_like_queries(nq, D, seed=2)).cuda()
    m = 0 if metric == "dot_product" else 1
    per = M // parts
    ex = [hdist.PackedTopK(nq, k, q.device, parts) for _ in range(parts)]
    assert ex[0].part_bytes == _lib.lib().hb_packed_list_bytes(nq, k)
    recv = torch.zeros(parts * ex[0].part_bytes, dtype=torch.uint8, device=q.device)
    idxs, dists = [], []
    for p in range(parts):
        sh = HipFlatIndex(D, m, 0)
        sh.add(torch.from_numpy(bank[p * per:(p + 1) * per]).cuda())
        i, s = sh.search_scores(q, k, id_base=p * per, out=(ex[p].idx, ex[p].dist))
        assert i.data_ptr() == ex[p].send.data_ptr()
        recv[p * ex[0].part_bytes:(p + 1) * ex[0].part_bytes] = ex[p].send        # what the all-gather does
        idxs.append(i.clone()); dists.append(s.clone())
    ia, da = merge_topk(torch.stack(dists), torch.stack(idxs), 0)
    ip, dp = merge_topk_packed(recv, ex[0].part_bytes, parts, nq, k, 0)
    assert torch.equal(ia, ip) and torch.equal(da.view(torch.int32), dp.view(torch.int32))
    full = HipFlatIndex(D, m, 0); full.add(torch.from_numpy(bank).cuda())
    ri, rd = full.search(q, k)
    assert torch.equal(ip, ri) and torch.equal(full.distances_from_scores(q, dp).view(torch.int32), rd.view(torch.int32))


@pytest.mark.parametrize("metric,shard,fp16", [("dot_product", True, False), ("l2", True, False), ("dot_product", False, False),
                                               ("l2", False, True), ("dot_product", True, True)])
def test_plugin_drives_several_indices_in_one_process(cuda_device, metric, shard, fp16):
    """gpu_ids with several entries in ONE process (the reference's call shape, search_faiss.py:50-76): one hb_index_t per
    entry, host threads, row shards (+ merge on the first GPU) or replicas (query split).  Listing cuda:0 twice puts two
    indices on the one GPU of this box: the result must be the single-index bits."""
    M, D, nq, k = 50_001, 64, 777, 30
    bank = gi.unit_bank(M, D, seed=7)
    bank[40_000:40_005] = bank[9]; bank[25_000] = bank[9]                 # ties that straddle the shard boundary
    q = gi.vit_like_queries(nq, D, seed=8); q[:5] = 3.0 * bank[9]
    fm = torch.from_numpy(bank)
    one = NearestNeighborSearchHIP(fm, n_neighbors=k, distance_measure=metric, gpu_ids=[0], use_fp16=fp16)
    i1, d1 = one.find_nearest_neighbors(torch.from_numpy(q))
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric)
    assert np.array_equal(i1, ridx) and np.array_equal(d1.view(np.uint32), rdist.view(np.uint32))
    for ids in ([0, 0], [0, 0, 0]):
        nn = NearestNeighborSearchHIP(fm, n_neighbors=k, distance_measure=metric, idx_shard=shard, gpu_ids=ids, use_fp16=fp16)
        assert len(nn.indexes) == len(ids)
        rows = [ix.ntotal for ix in nn.indexes]
        assert (sum(rows) == M) if shard else all(r == M for r in rows)
        i2, d2 = nn.find_nearest_neighbors(torch.from_numpy(q))            # host queries, numpy out
        assert isinstance(i2, np.ndarray) and np.array_equal(i2, i1) and np.array_equal(d2.view(np.uint32), d1.view(np.uint32))
        i3, d3 = nn.find_nearest_neighbors(torch.from_numpy(q).cuda(), k=7)   # device queries, k override
        assert i3.is_cuda and np.array_equal(i3.cpu().numpy(), i1[:, :7]) and np.array_equal(d3.cpu().numpy().view(np.uint32), d1[:, :7].view(np.uint32))


def test_patch_label_hist_rejects_out_of_range_classes(cuda_device):
    """F.one_hot of the reference raises for a class id >= num_classes (hbird_eval.py:319); so does K2 -- as a ValueError
    that is also the library's RuntimeError -- instead of silently dropping the pixel from the histogram."""
    y = torch.randint(0, 5, (2, 1, 32, 32), device="cuda")
    out = ops.patch_label_hist(y, 16, 5)
    assert torch.allclose(out.sum(-1), torch.ones_like(out.sum(-1)))
    y[1, 0, 7, 9] = 5
    with pytest.raises(ValueError, match="out-of-range class"):
        ops.patch_label_hist(y, 16, 5)
    with pytest.raises(RuntimeError):
        ops.patch_label_hist(y, 16, 5)
    y[1, 0, 7, 9] = 255                      # the ignore value: only legal when it is mapped to class 0 (hbird_eval.py:310)
    with pytest.raises(ValueError):
        ops.patch_label_hist(y, 16, 5, map255=False)
    assert float(ops.patch_label_hist(y, 16, 5, map255=True).sum()) == pytest.approx(2 * 2 * 2)
    y[1, 0, 7, 9] = -1
    with pytest.raises(ValueError):
        ops.patch_label_hist(y, 16, 5, map255=True)


def _run_bench(args, env_extra, timeout=900):
    env = dict(os.environ); env.update(env_extra)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


_SMALL = ["--rows", "600000", "--dim", "64", "--classes", "21", "--nq", "3000", "--steps", "3", "--warmup", "1",
          "--no-cpu-baseline", "--no-traffic"]


def test_bench_gpus_2_runs_two_ranks(cuda_device):
    """`python bench.py --gpus 2` (no torchrun environment) starts two ranks itself and reports n_gpus = 2.  On this
    1-GPU box the ranks share cuda:0 and talk over gloo (HBIRD_BENCH_ONE_GPU=1; RCCL refuses two ranks on one device)."""
    one = _run_bench(["--gpus", "1"] + _SMALL, {})
    two = _run_bench(["--gpus", "2"] + _SMALL, {"HBIRD_BENCH_ONE_GPU": "1"})
    assert one["n_gpus"] == 1 and one["config"]["parallelism"] == "single-gpu" and "multi_gpu" not in one
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "bank-shard2" and two["scaling"] == "strong"
    mg = two["multi_gpu"]
    assert mg["world_size"] == 2 and mg["rows_per_rank"] == [300000, 300000] and len(mg["knn_ms_per_rank"]) == 2
    assert mg["packed_list_bytes_per_rank"] == (3000 * 30 * 12 + 15) // 16 * 16
    assert all(x > 0 for x in mg["knn_ms_per_rank"]) and all(x > 0 for x in mg["exchange_ms_per_rank"])
    assert two["value"] > 0 and two["roofline"]["algorithmic_flops_per_launch"] == 2.0 * 3000 * 300000 * 64
    # what the first real N-GPU run needs on its line: the exposed exchange split three ways, the N=1-equivalent efficiency, and the
    # use_fp16 leg as whole N-rank steps
    sp = mg["exchange_split_ms_per_rank"]
    assert set(sp) == {"all_gather", "merge", "aggregate"} and all(len(v) == 2 and min(v) >= 0 for v in sp.values())
    assert 0 < mg["efficiency"] <= 1.05 and mg["n1_equivalent_ms"] == pytest.approx(sum(mg["knn_ms_per_rank"]), rel=1e-3)
    f16 = two["use_fp16_mode"]
    assert f16["value"] > 0 and len(f16["knn_ms_per_rank"]) == 2 and f16["fallback_queries_rank0"] == 0


def test_bench_rccl_path_with_one_rank(cuda_device):
    """The N-rank code path on the real backend: init_process_group("nccl") = RCCL, the packed all-gather, the in-place
    merge and the aggregation against the replicated label table -- with world_size 1 (all this box can host)."""
    r = _run_bench(["--gpus", "1"] + _SMALL, {"HBIRD_BENCH_FORCE_DIST": "1"})
    assert r["n_gpus"] == 1 and r["multi_gpu"]["backend"].startswith("nccl") and r["multi_gpu"]["world_size"] == 1
    assert r["multi_gpu"]["rows_per_rank"] == [600000]


def test_bench_gpus_8_ragged_shards_equal_the_single_rank_result(cuda_device):
    """The 8-rank path end to end before the first real 8-GPU run: `bench.py --gpus 8` on a bank whose row count is NOT
    divisible by 8 (ragged shards, the last rank short), eight replicated label tables, a `parts = 8` packed merge fed by a
    real all-gather (gloo here: the ranks share cuda:0) -- and the merged + aggregated result carries the same bits as the
    1-rank run (`label_hat_checksum`: int64 sum of the fp32 bit patterns over all queries)."""
    small = ["--rows", "600001", "--dim", "64", "--classes", "21", "--nq", "3001", "--steps", "2", "--warmup", "1",
             "--no-cpu-baseline", "--no-traffic", "--checksum"]
    one = _run_bench(["--gpus", "1"] + small, {})
    eight = _run_bench(["--gpus", "8"] + small, {"HBIRD_BENCH_ONE_GPU": "1"}, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["config"]["parallelism"] == "bank-shard8"
    mg = eight["multi_gpu"]
    assert mg["world_size"] == 8 and len(mg["rows_per_rank"]) == 8 and sum(mg["rows_per_rank"]) == 600001
    assert mg["rows_per_rank"] == [75001] * 7 + [74994] and len(set(mg["rows_per_rank"])) == 2
    assert len(mg["knn_ms_per_rank"]) == 8 and all(x > 0 for x in mg["knn_ms_per_rank"])
    assert mg["device_per_rank"] == [0] * 8                       # test mode; one rank per GPU reports 0..7
    c1, c8 = one["label_hat_checksum"], eight["label_hat_checksum"]
    assert sum(c8["rows_per_rank"]) == 3001 and len(c8["rows_per_rank"]) == 8
    assert c1["bits"] == c8["bits"], (c1, c8)
    assert abs(c1["sum"] - c8["sum"]) <= 1e-9 * abs(c1["sum"])
    assert abs(c1["sum"] - 3001) < 1e-2                           # soft labels: every label_hat row sums to 1


@pytest.mark.parametrize("metric,shard,fp16", [("dot_product", 1, False), ("l2", 1, False), ("l2", 0, False), ("dot_product", 1, True)])
def test_c_abi_multi_gpu_handle(cuda_device, metric, shard, fp16):
    """hb_multi_*: shards / replicas over several GPUs behind ONE handle of the C ABI (what a non-Python host binds instead of
    redoing the composition; search_faiss.py:50-76), host buffers in and out like the reference's numpy arrays.  cuda:0 listed
    three times: the single-index bits, ties across the shard boundaries included, rows appended in ragged pieces."""
    import ctypes
    L = _lib.lib()
    M, D, nq, k = 50_001, 64, 777, 30
    bank = gi.unit_bank(M, D, seed=7)
    bank[40_000:40_005] = bank[9]; bank[25_000] = bank[9]
    q = gi.vit_like_queries(nq, D, seed=8); q[:5] = 3.0 * bank[9]
    m = 0 if metric == "dot_product" else 1
    h = ctypes.c_void_p()
    ids = (ctypes.c_int * 3)(0, 0, 0)
    bad = (ctypes.c_int * 2)(0, 99)
    assert L.hb_multi_create(D, m, bad, 2, shard, ctypes.byref(h)) != 0 and b"invalid GPU id" in L.hb_last_error()
    _lib.check(L.hb_multi_create(D, m, ids, 3, shard, ctypes.byref(h)))
    try:
        _lib.check(L.hb_multi_set_fp16(h, 1 if fp16 else 0))
        _lib.check(L.hb_multi_reserve(h, M))
        for a, b in ((0, 10_000), (10_000, 33_333), (33_333, M)):            # pieces that straddle the shard boundaries
            piece = np.ascontiguousarray(bank[a:b])
            _lib.check(L.hb_multi_add(h, piece.ctypes.data_as(ctypes.c_void_p), b - a, 0))
        rows = (ctypes.c_int64 * 3)()
        _lib.check(L.hb_multi_shard_rows(h, rows, 3))
        assert L.hb_multi_ntotal(h) == M
        assert list(rows) == ([16667, 16667, 16667] if shard else [M, M, M])
        idx = np.empty((nq, k), np.int64); dist = np.empty((nq, k), np.float32)
        _lib.check(L.hb_multi_search(h, q.ctypes.data_as(ctypes.c_void_p), nq, k, idx.ctypes.data_as(ctypes.c_void_p),
                                     dist.ctypes.data_as(ctypes.c_void_p)))
        ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric)
        assert np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
        assert L.hb_multi_search(h, q.ctypes.data_as(ctypes.c_void_p), nq, 0, idx.ctypes.data_as(ctypes.c_void_p),
                                 dist.ctypes.data_as(ctypes.c_void_p)) != 0
    finally:
        L.hb_multi_free(h)


@pytest.mark.parametrize("metric,shard", [("dot_product", 1), ("l2", 1), ("dot_product", 0)])
def test_c_abi_multi_gpu_handle_on_distinct_devices(cuda_device, metric, shard):
    """The same through TWO DISTINCT GPUs (peer copies and per-device streams that `gpu_ids=[0, 0, 0]` never exercises), and the Python
    composition (HipMultiIndex) beside it: skipped where only one GPU is visible (the build pool), live on a multi-GPU node."""
    import ctypes
    from hbird_mi.nn.search_hip import NearestNeighborSearchHIP
    L = _lib.lib()
    if _lib.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    M, D, nq, k = 40_003, 96, 513, 30
    bank = gi.unit_bank(M, D, seed=17)
    bank[30_000:30_004] = bank[5]
    q = gi.vit_like_queries(nq, D, seed=18); q[:4] = 2.0 * bank[5]
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric)
    h = ctypes.c_void_p()
    ids = (ctypes.c_int * 2)(0, 1)
    _lib.check(L.hb_multi_create(D, 0 if metric == "dot_product" else 1, ids, 2, shard, ctypes.byref(h)))
    try:
        _lib.check(L.hb_multi_reserve(h, M))
        for a, b in ((0, 15_000), (15_000, 25_001), (25_001, M)):
            piece = np.ascontiguousarray(bank[a:b])
            _lib.check(L.hb_multi_add(h, piece.ctypes.data_as(ctypes.c_void_p), b - a, 0))
        idx = np.empty((nq, k), np.int64); dist = np.empty((nq, k), np.float32)
        _lib.check(L.hb_multi_search(h, q.ctypes.data_as(ctypes.c_void_p), nq, k, idx.ctypes.data_as(ctypes.c_void_p),
                                     dist.ctypes.data_as(ctypes.c_void_p)))
        assert np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    finally:
        L.hb_multi_free(h)
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, idx_shard=bool(shard), gpu_ids=[0, 1])
    pi, pd = nn.find_nearest_neighbors(torch.from_numpy(q).cuda())
    pi = pi.cpu().numpy() if isinstance(pi, torch.Tensor) else pi; pd = pd.cpu().numpy() if isinstance(pd, torch.Tensor) else pd
    assert np.array_equal(pi, ridx) and np.array_equal(np.asarray(pd).view(np.uint32), rdist.view(np.uint32))
