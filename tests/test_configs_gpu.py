"""Parity at the shapes BASELINE.json's configs name: every config's D / k / bank size goes
through the HIP path and is checked against the CPU oracle (oracle.knn_chain_f32: bit-exact indices and distance bits)
where the oracle finishes in seconds, and through size-independent properties at the full sizes.

  cfg-1  50,176 x 384, 12,544 queries, k = 30      full oracle (the cold-start instantiation of the kernel)
  cfg-2  2,074,072 x 384, 12,544 queries, k = 30   full-size bank, oracle on a 512-query sample
  cfg-3  10 M x 768, k = 30                        tests/test_knn_gpu.py::test_headline_size_10m_x_768
  cfg-4  20,345,364 x 1024 (83 GB), k = 30         D = 1024 vs the oracle at 30 k rows + full-size property test
  cfg-5  768-d bank, k = 90                        D = 768 / k = 90 vs the oracle + 10 M-row property test (pool path)
"""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import oracle
from helpers import chain_oracle_topk_chunked
from hbird_mi.nn.search_hip import HipFlatIndex, NearestNeighborSearchHIP, merge_topk

pytestmark = pytest.mark.gpu


def _check_exact(idx, dist, q, bank, k, metric, id_base=0):
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric, id_base)
    idx = idx.cpu().numpy() if isinstance(idx, torch.Tensor) else idx
    dist = dist.cpu().numpy() if isinstance(dist, torch.Tensor) else dist
    bad = np.argwhere(idx != ridx)
    assert bad.size == 0, f"{len(bad)} index mismatches, first {bad[:5].tolist()}"
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32)), "distance bits differ"


# ---- cfg-4's width: D = 1024 ------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,fp16,k", [("dot_product", False, 30), ("l2", False, 30), ("dot_product", True, 30),
                                           ("l2", True, 30), ("dot_product", False, 90)])
def test_cfg4_width_1024_vs_oracle(cuda_device, metric, fp16, k):
    M, D, nq = 30_000, 1024, 700
    bank = gi.unit_bank(M, D, seed=41)
    bank[20_000] = bank[17]                                  # an exact tie
    q = gi.vit_like_queries(nq, D, seed=42)
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, use_fp16=fp16, gpu_ids=[0])
    idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))
    _check_exact(idx, dist, q, bank, k, metric)
    nn.index.set_tuning(7, 3)                                # several slots per query tile
    idx, dist = nn.index.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)


# ---- cfg-5's k = 90 at the real width -----------------------------------------------------------------------------
@pytest.mark.parametrize("metric,fp16", [("dot_product", False), ("l2", False), ("dot_product", True)])
def test_cfg5_k90_width_768_vs_oracle(cuda_device, metric, fp16):
    M, D, nq, k = 40_000, 768, 520, 90
    bank = gi.unit_bank(M, D, seed=51)
    bank[39_000:39_003] = bank[5]
    q = gi.vit_like_queries(nq, D, seed=52)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(fp16)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    ix.set_tuning(64, 4)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)


# ---- cfg-1: the exact shape, full oracle ------------------------------------------------------------------------------
def test_cfg1_exact_shape_vs_full_oracle(cuda_device):
    """256 images x 196 patches = 50,176 bank rows of 384 dims against a 64 x 196 = 12,544-query batch: the search with few
    bank tiles per workgroup (radix-select cold start of every slot)."""
    M, D, nq, k = 50_176, 384, 12_544, 30
    bank = gi.unit_bank(M, D, seed=11)
    q = gi.vit_like_queries(nq, D, seed=12)
    ix = HipFlatIndex(D, 0, 0)
    ix.add(torch.from_numpy(bank).cuda())
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    info = ix.schedule_info()
    assert info["query_tiles"] == 49 and info["bank_tiles"] == 196
    _check_exact(idx, dist, q, bank, k, "dot_product")
    ix.set_fp16(True)
    idx16, dist16 = ix.search(torch.from_numpy(q).cuda(), k)
    assert torch.equal(idx16, idx) and torch.equal(dist16, dist)


def _device_bank(M, D, seed, dev, targets, chunk=500_000):
    """Rows N(0,1)/|.| generated on the device in chunks and appended to every index in `targets` whose [lo, hi) range
    they fall into: (index, lo, hi) triples."""
    g = torch.Generator(device=dev).manual_seed(seed)
    for r in range(0, M, chunk):
        n = min(chunk, M - r)
        rows = torch.randn((n, D), generator=g, device=dev)
        rows = rows / rows.norm(dim=1, keepdim=True)
        for ix, lo, hi in targets:
            a, b = max(lo, r), min(hi, r + n)
            if a < b:
                ix.add(rows[a - r:b - r])
    return g


def _float64_topk(ix, q_sel, M, k, dev, chunk=1_000_000):
    best_s = torch.full((q_sel.shape[0], k), -float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.full((q_sel.shape[0], k), -1, dtype=torch.int64, device=dev)
    qs = q_sel.double()
    for r in range(0, M, chunk):
        ids = torch.arange(r, min(M, r + chunk), device=dev)
        sc = qs @ ix.reconstruct(ids).double().T
        cs = torch.cat([best_s, sc], dim=1); ci = torch.cat([best_i, ids[None].expand(q_sel.shape[0], -1)], dim=1)
        top = cs.topk(k, dim=1)
        best_s, best_i = top.values, ci.gather(1, top.indices)
    return best_i, best_s


def _full_size_properties(M, D, nq, ks, dev, seed, n_plant=128):
    """Planted neighbours, sortedness, distinct ids, determinism, 2-shard merge == single index, use_fp16 == fp32 bits, the fp32
    chain oracle bit for bit on 512 queries against ALL rows (chunked; whatever kernel the size selects automatically -- clusters,
    pools -- and the use_fp16 path) and a float64 check of 16 queries -- for every k in `ks` on one bank."""
    ix = HipFlatIndex(D, 0, 0); ix.reserve(M)
    half = M // 2
    a, b = HipFlatIndex(D, 0, 0), HipFlatIndex(D, 0, 0)
    a.reserve(half); b.reserve(M - half)
    g = _device_bank(M, D, seed, dev, [(ix, 0, M), (a, 0, half), (b, half, M)])
    assert ix.ntotal == M and a.ntotal == half and b.ntotal == M - half
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    planted = torch.arange(n_plant, device=dev) * (M // n_plant) + 11
    q[:n_plant] = 2.5 * ix.reconstruct(planted)
    sel = torch.linspace(0, nq - 1, 16, device=dev).long()
    out = {}
    for k in ks:
        idx, dist = ix.search(q, k)
        assert (idx[:n_plant, 0] == planted).all()
        assert (dist[:, :-1] >= dist[:, 1:]).all() and (idx >= 0).all() and (idx < M).all()
        srt = idx.sort(dim=1).values
        assert (srt[:, 1:] != srt[:, :-1]).all()
        idx2, dist2 = ix.search(q, k)
        assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
        ia, da = a.search(q, k, id_base=0)
        ib, db = b.search(q, k, id_base=half)
        im, dm = merge_topk(torch.stack([da, db]), torch.stack([ia, ib]), 0)
        assert torch.equal(im, idx) and torch.equal(dm, dist)
        out[k] = (idx, dist)
    del a, b
    torch.cuda.empty_cache()
    ix.set_fp16(True)
    for k in ks:
        idx16, dist16 = ix.search(q, k)
        assert torch.equal(idx16, out[k][0]) and torch.equal(dist16, out[k][1])
        assert ix.last_fp16_fallbacks() < max(1, nq // 100)
    ix.set_fp16(False)
    kmax = max(ks)
    sel32 = torch.linspace(0, nq - 1, 512, device=dev).long()      # (512 queries, 2.3 % of the batch: the bank chunks' trip to the host is most of what the check costs)
    ci, cd = chain_oracle_topk_chunked(ix, q[sel32], M, kmax)
    for k in ks:      # idx16 == idx was asserted above: the same bits hold for the use_fp16 path
        assert np.array_equal(out[k][0][sel32].cpu().numpy(), ci[:, :k]), f"k={k}: indices differ from the chain oracle at full size"
        assert np.array_equal(out[k][1][sel32].cpu().numpy().view(np.uint32), cd[:, :k].view(np.uint32)), f"k={k}: distance bits differ"
    best_i, best_s = _float64_topk(ix, q[sel], M, kmax, dev)
    for k in ks:
        got = out[k][0][sel].cpu().numpy(); ref = best_i[:, :k].cpu().numpy()
        # a D-term fp32 chain at |q| ~ 3 sqrt(D) is off by a few 1e-5: positions may swap where the float64 scores are
        # closer than that (64 fp32 ulps of the score), nowhere else
        rep = oracle.near_tie_report(got, ref, best_s[:, :k].cpu().numpy(), ulps=64.0)
        assert rep["excused_rate"] == 1.0 and rep["set_rate"] >= 0.8, (k, rep)
        assert np.abs(out[k][1][sel].cpu().numpy() - best_s[:, :k].cpu().numpy()).max() < 2e-4


def test_cfg4_full_size_20m_x_1024(cuda_device):
    """BASELINE cfg-4: 20,345,364 x 1024 (an 83 GB bank; with the two half-size shards 166 GB of the 288 GB)."""
    _full_size_properties(20_345_364, 1024, 16 * 1369, [30], torch.device("cuda:0"), seed=44)


def test_cfg5_k90_full_size_10m_x_768(cuda_device):
    """BASELINE cfg-5's retrieval shape: k = 90 over a 10 M x 768 bank (candidate pools in HBM, radix-select compaction)."""
    _full_size_properties(10_000_000, 768, 16 * 1369, [90], torch.device("cuda:0"), seed=55)


def test_cfg2_full_size_bank_oracle_on_a_query_sample(cuda_device):
    """BASELINE cfg-2: the full 2,074,072 x 384 bank and the 12,544-query batch; the oracle checks a 512-query sample bit
    for bit (the full oracle takes minutes), every query's list is checked for order and distinct ids."""
    M, D, nq, k = 2_074_072, 384, 12_544, 30
    dev = torch.device("cuda:0")
    ix = HipFlatIndex(D, 0, 0); ix.reserve(M)
    g = _device_bank(M, D, 22, dev, [(ix, 0, M)])
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    idx, dist = ix.search(q, k)
    assert (dist[:, :-1] >= dist[:, 1:]).all() and (idx >= 0).all() and (idx < M).all()
    sel = torch.randperm(nq, generator=torch.Generator().manual_seed(3))[:512].to(dev)
    bank = ix.reconstruct(torch.arange(M, device=dev)).cpu().numpy()
    _check_exact(idx[sel], dist[sel], q[sel].cpu().numpy(), bank, k, "dot_product")
    ix.set_fp16(True)
    idx16, dist16 = ix.search(q, k)
    assert torch.equal(idx16, idx) and torch.equal(dist16, dist)


# ---- the widest published operating point: DINOv2 ViT-G/14, D = 1536 (reference README.md:327-334) ----------------------------------
@pytest.mark.parametrize("metric,fp16,k", [("dot_product", False, 30), ("l2", False, 30), ("dot_product", True, 30), ("l2", True, 30),
                                           ("dot_product", False, 90), ("dot_product", True, 90)])
def test_vit_g14_width_1536_vs_oracle(cuda_device, metric, fp16, k):
    M, D, nq = 30_000, 1536, 600
    bank = gi.unit_bank(M, D, seed=61)
    bank[25_000] = bank[9]                                   # an exact tie
    q = gi.vit_like_queries(nq, D, seed=62)
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, use_fp16=fp16, gpu_ids=[0])
    idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))
    _check_exact(idx, dist, q, bank, k, metric)
    nn.index.set_fp16(1 if fp16 else 0)                      # the candidate pass itself (the plugin's mode 2 may decline a bank this small)
    nn.index.set_tuning(11, 5)                               # several slots per query tile
    idx, dist = nn.index.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    if fp16:
        assert nn.index.last_fp16_fallbacks() < nq // 20


def test_vit_g14_width_1536_two_million_rows(cuda_device):
    """D = 1536 at a bank size the reference publishes for it (1024 x 10^3 x ... rows, README.md:327-334): planted neighbours, order, distinct
    ids, determinism, 2-shard merge, use_fp16 == fp32 bits, the chain oracle on 512 queries against all rows, float64 on 16."""
    _full_size_properties(2_000_000, 1536, 8 * 1369, [30], torch.device("cuda:0"), seed=66)


# ---- k beyond the candidate pass (k' = 2k <= 256): use_fp16 is served by the fp32 kernel, as documented ------------------------------
@pytest.mark.parametrize("k", [129, 200, 256])
def test_use_fp16_beyond_k_128_is_the_fp32_search(cuda_device, k):
    """include/hbird_hip.h: the fp16 candidate pass keeps k' = 2k <= 256 candidates, so a use_fp16 search
    with k > 128 runs on the exact fp32 kernel: no fallback queries are counted, ids and distance bits are the chain oracle's."""
    M, D, nq = 50_000, 256, 400
    bank = gi.unit_bank(M, D, seed=71); q = gi.vit_like_queries(nq, D, seed=72)
    ix = HipFlatIndex(D, 0, 0); ix.add(torch.from_numpy(bank).cuda())
    i32, d32 = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(i32, d32, q, bank, k, "dot_product")
    for mode in (1, 2):
        ix.set_fp16(mode)
        i16, d16 = ix.search(torch.from_numpy(q).cuda(), k)
        assert ix.last_fp16_fallbacks() == 0
        assert torch.equal(i16, i32) and torch.equal(d16.view(torch.int32), d32.view(torch.int32))
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, use_fp16=True, gpu_ids=[0])
    i2, d2 = nn.find_nearest_neighbors(torch.from_numpy(q))
    _check_exact(i2, d2, q, bank, k, "dot_product")
    with pytest.raises(ValueError):
        nn.find_nearest_neighbors(torch.from_numpy(q), k=2049)


@pytest.mark.parametrize("k,metric", [(257, "dot_product"), (512, "l2"), (1024, "dot_product"), (2048, "dot_product")])
def test_k_beyond_256_in_passes_behind_a_ceiling(cuda_device, k, metric):
    """The reference forwards any k to Faiss (search_faiss.py:84-85; faiss-gpu: up to 2048).  Beyond the pools' 256 a search runs
    ceil(k / 256) passes, each restricted to the rows BEHIND the last neighbour delivered so far in the ordering (score, id): ids and
    distance bits of the chain oracle for one list of k -- 50,000 x 256 as VERDICT r05 asked, exact duplicates planted across the pass
    boundaries (ties by id), and through the plugin's host path."""
    M, D, nq = 50_000, 256, 300
    bank = gi.unit_bank(M, D, seed=81); q = gi.vit_like_queries(nq, D, seed=82)
    q[0] = 5.0 * bank[123]
    bank[[9_000, 20_000, 31_000, 42_000]] = bank[123]               # exact ties at the very top ...
    order = np.argsort(-(q[1] @ bank.T))
    bank[order[258]] = bank[order[254]]; bank[order[513]] = bank[order[510]]      # ... and across the first two pass boundaries of query 1
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0); ix.add(torch.from_numpy(bank).cuda())
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    srt = idx.sort(dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all()                           # no row twice
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=30, distance_measure=metric, gpu_ids=[0])
    i2, d2 = nn.find_nearest_neighbors(torch.from_numpy(q[:64]), k=k)   # host buffers, k override (search_faiss.py:84-85)
    _check_exact(i2, d2, q[:64], bank, k, metric)


@pytest.mark.parametrize("k", [300, 384, 600])
def test_k_beyond_256_with_use_fp16_set(cuda_device, k):
    """use_fp16 applies up to k = 128; a search with k > 256 whose LAST pass is that narrow (k = 300: 256 + 44) must still run it behind the
    ceiling on the fp32 kernel -- the flag is accepted, the bits are the fp32 search's."""
    M, D, nq = 30_000, 128, 200
    bank = gi.unit_bank(M, D, seed=101); q = gi.vit_like_queries(nq, D, seed=102)
    ix = HipFlatIndex(D, 0, 0); ix.add(torch.from_numpy(bank).cuda())
    for mode in (1, 2):
        ix.set_fp16(mode)
        idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
        _check_exact(idx, dist, q, bank, k, "dot_product")
        assert ix.last_fp16_fallbacks() == 0 and ix.last_fp16_escalated() == 0


def test_k_beyond_256_on_a_bank_with_fewer_rows(cuda_device):
    """k = 700 over 600 rows, three of them NaN: 597 neighbours, then id -1 / -inf -- the list closes in the pass where the rows run out
    and later passes deliver nothing."""
    M, D, nq, k = 600, 64, 70, 700
    bank = gi.unit_bank(M, D, seed=91); bank[[5, 300, 599]] = np.nan
    q = gi.vit_like_queries(nq, D, seed=92)
    ix = HipFlatIndex(D, 0, 0); ix.add(torch.from_numpy(bank).cuda())
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, "dot_product")
    assert (idx[:, 597:] == -1).all() and (idx[:, :597] >= 0).all()
