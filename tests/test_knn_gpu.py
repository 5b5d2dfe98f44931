"""Parity of the fused HIP kNN kernel (through the C ABI) against the CPU oracle.

Bit-exact target: oracle.knn_chain_f32 (every score one k-ascending fp32 fmaf chain, ties by lower
id) -- indices AND distances must be identical.  The float64 definition is checked modulo near-ties.
"""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import oracle
from hbird_mi.nn.search_hip import HipFlatIndex, NearestNeighborSearchHIP

pytestmark = pytest.mark.gpu


def _check_exact(idx, dist, q, bank, k, metric, id_base=0):
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric, id_base)
    idx = idx.cpu().numpy() if isinstance(idx, torch.Tensor) else idx
    dist = dist.cpu().numpy() if isinstance(dist, torch.Tensor) else dist
    assert idx.dtype == np.int64 and dist.dtype == np.float32
    bad = np.argwhere(idx != ridx)
    assert bad.size == 0, f"{len(bad)} index mismatches, first {bad[:5].tolist()}: got {idx[tuple(bad[0])]} want {ridx[tuple(bad[0])]}"
    # bit-exact distances (compare the raw bits; -inf/+inf sentinels included)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32)), \
        f"distance bits differ, max abs diff {np.nanmax(np.abs(dist - rdist))}"


@pytest.mark.parametrize("M,D,nq,k,metric", [
    (1000, 32, 100, 30, "dot_product"),
    (1000, 32, 100, 30, "l2"),
    (5000, 64, 257, 1, "dot_product"),
    (777, 20, 33, 32, "dot_product"),        # D not a multiple of 16, M and nq ragged, k = HB_MAX_K
    (20000, 384, 300, 30, "dot_product"),    # ViT-S width
    (20000, 384, 300, 30, "euclidean"),
    (9000, 768, 520, 30, "dot_product"),     # ViT-B width, 3 query tiles
])
def test_search_bit_exact_host_path(cuda_device, M, D, nq, k, metric):
    bank = gi.unit_bank(M, D, seed=M + D)
    q = gi.vit_like_queries(nq, D, seed=nq + D)
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, gpu_ids=[0])
    idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))
    assert isinstance(idx, np.ndarray) and idx.shape == (nq, k)
    _check_exact(idx, dist, q, bank, k, metric)


def test_search_device_path_and_k_override(cuda_device):
    M, D, nq = 6000, 128, 300
    bank = gi.unit_bank(M, D, seed=1)
    q = gi.vit_like_queries(nq, D, seed=2)
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=30, gpu_ids=[0])
    idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q).cuda(), k=7)   # k override, search_faiss.py:84-85
    assert idx.is_cuda and idx.shape == (nq, 7)
    _check_exact(idx, dist, q, bank, 7, "dot_product")


@pytest.mark.parametrize("G,panel", [(1, 1), (3, 2), (7, 1), (16, 5), (64, 3)])
def test_search_any_work_partition(cuda_device, G, panel):
    """The result must not depend on how the (query tile, bank tile) pairs are cut into workgroup segments."""
    M, D, nq, k = 4000, 48, 700, 30
    bank = gi.unit_bank(M, D, seed=5)
    q = gi.vit_like_queries(nq, D, seed=6)
    ix = HipFlatIndex(D, 0, 0)
    ix.add(bank)
    ix.set_tuning(G, panel)
    idx, dist = ix.search(q, k)
    info = ix.schedule_info()
    assert info["workgroups"] == min(G, info["query_tiles"] * info["bank_tiles"])
    _check_exact(idx, dist, q, bank, k, "dot_product")


@pytest.mark.parametrize("D", [48, 64])     # 6 k8 stages per tile: the LDS-staged kernels; 8: the kernel with register-resident query fragments
@pytest.mark.parametrize("cq,cb,lag,G,panel,k,fp16,metric", [
    (2, 2, 6, 32, 0, 30, False, "dot_product"), (2, 2, 0, 64, 6, 30, False, "l2"), (4, 2, 3, 64, 0, 30, False, "dot_product"),
    (2, 4, 1, 64, 5, 90, False, "dot_product"), (1, 4, 6, 32, 3, 30, True, "dot_product"), (8, 1, 2, 64, 0, 30, True, "l2"),
    (2, 2, 6, 256, 0, 30, False, "dot_product"), (2, 2, 4, 256, 0, 64, True, "dot_product"), (2, 4, 16, 256, 0, 30, False, "l2"),
    (4, 2, 3, 128, 0, 90, True, "dot_product"),      # 16 slots per query tile x pools of 512: the floor kernel with 128 KiB of LDS
])
def test_clustered_schedules_bit_exact(cuda_device, cq, cb, lag, G, panel, k, fp16, metric, D):
    """L2-sharing clusters (strided segments, common cluster clock, soft sync on progress words) are a speed feature: the
    result must be the oracle's bits for any cluster shape, sync lag (0 = never wait), ragged query / bank groups."""
    M, nq = 70_001, 1300                   # 6 query tiles (ragged), 274 bank tiles (odd: partial bank groups)
    bank = gi.unit_bank(M, D, seed=31)
    bank[60_000:60_004] = bank[11]; bank[257] = bank[11]
    q = gi.vit_like_queries(nq, D, seed=32); q[:4] = 3.0 * bank[11]
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(fp16)
    ix.set_tuning(G, panel)
    ix.set_cluster(cq, cb, lag)
    for _ in range(2):
        idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
        assert ix.schedule_info()["cluster"] == [cq, cb]
        _check_exact(idx, dist, q, bank, k, metric)
    ix.set_cluster(1, 1, 0)
    i1, d1 = ix.search(torch.from_numpy(q).cuda(), k)
    assert ix.schedule_info()["cluster"] == [1, 1] and torch.equal(i1, idx) and torch.equal(d1, dist)


def test_cluster_soft_sync_holds_the_members_together(cuda_device):
    """Speed only, so the parity tests cannot see it break: in every kernel that syncs (B-direct fp32, LDS-staged fp32, fp16
    candidate kernel) and for 4- and 8-member shapes, the members must keep meeting -- progress checks happen all along the
    search and (almost) nobody gives up waiting.  (A change that moved the sync's counters to LDS once made every member of
    every 4-member cluster time out at its first wait: same results, 8 ms slower per workgroup.)"""
    M, D, nq, k = 600_000, 64, 2560, 30
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    bank = torch.nn.functional.normalize(torch.randn((M, D), generator=g, device="cuda"), dim=1)
    q = 3.0 * torch.randn((nq, D), generator=g, device="cuda")
    ix = HipFlatIndex(D, 0, 0)
    ix.add(bank)
    ref = ix.search(q, k)
    for variant, fp16 in ((0, False), (4, False), (0, True)):
        ix.set_variant(variant); ix.set_fp16(fp16)
        for shape in ((2, 2, 16), (4, 1, 16), (2, 4, 16), (8, 1, 16)):
            ix.set_cluster(*shape)
            idx, dist = ix.search(q, k)
            st = ix.cluster_stats()
            assert ix.schedule_info()["cluster"] == list(shape[:2])
            assert torch.equal(idx, ref[0]) and torch.equal(dist, ref[1])
            assert st["checks"] >= 256 and st["timeouts"] <= 8, (variant, fp16, shape, st)
    ix.set_cluster(0, 0, -1); ix.set_variant(0); ix.set_fp16(False)


def test_incremental_add_and_id_base(cuda_device):
    """Appending in ragged chunks (device and host rows mixed) == adding once; id_base offsets the ids."""
    M, D, nq, k = 3001, 64, 64, 30
    bank = gi.unit_bank(M, D, seed=9)
    q = gi.vit_like_queries(nq, D, seed=10)
    ix = HipFlatIndex(D, 0, 0)
    cuts = [0, 17, 300, 301, 1500, 3001]
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        chunk = torch.from_numpy(bank[a:b])
        ix.add(chunk.cuda() if i % 2 else chunk)
    assert ix.ntotal == M
    idx, dist = ix.search(q, k, id_base=1_000_000_000_000)
    _check_exact(idx, dist, q, bank, k, "dot_product", id_base=1_000_000_000_000)


def test_fewer_rows_than_k_and_empty(cuda_device):
    D, k = 16, 30
    bank = gi.unit_bank(10, D, seed=3)
    q = gi.vit_like_queries(5, D, seed=4)
    ix = HipFlatIndex(D, 0, 0)
    idx, dist = ix.search(q, k)                      # empty index: all -1 (faiss convention)
    assert (idx == -1).all() and np.isneginf(dist).all()
    ix.add(bank)
    idx, dist = ix.search(q, k)
    _check_exact(idx, dist, q, bank, k, "dot_product")
    assert (idx[:, 10:] == -1).all()
    idx0, _ = ix.search(q[:0], k)
    assert idx0.shape == (0, k)


def test_exact_ties_lower_id_first(cuda_device):
    """Duplicate bank rows score identically; the lower row id must come first and none may be dropped."""
    M, D, k = 2048, 32, 30
    bank = gi.unit_bank(M, D, seed=11)
    for dup in (5, 300, 301, 1029, 2047):
        bank[dup] = bank[77]
    q = gi.vit_like_queries(40, D, seed=12)
    q[0] = 5.0 * bank[77]
    ix = HipFlatIndex(D, 0, 0)
    ix.add(bank)
    idx, dist = ix.search(q, k)
    _check_exact(idx, dist, q, bank, k, "dot_product")
    assert idx[0, :6].tolist() == [5, 77, 300, 301, 1029, 2047]
    # a bank made of ONE repeated row: every score ties, ids must be 0..k-1
    ix2 = HipFlatIndex(D, 0, 0)
    ix2.add(np.repeat(bank[:1], 1000, axis=0))
    idx2, _ = ix2.search(q[:3], k)
    assert (idx2 == np.arange(k)[None, :]).all()


def test_golden_g4_replay(cuda_device, golden_dir):
    """Fixture produced by the reference's _find_nearest_key_to_query over the float64 exact backend."""
    g = np.load(f"{golden_dir}/g4_knn.npz")
    for name in ("ip32", "l2_32", "ip384"):
        M, D, B, N, k, C = g[f"shape_{name}"].tolist()
        metric = str(g[f"metric_{name}"])
        bank = gi.unit_bank(M, D, seed=41)
        q = gi.vit_like_queries(B * N, D, seed=43)
        if name == "ip32":
            bank[1234] = bank[77]; bank[4000] = bank[77]; q[0] = 5.0 * bank[77]
        nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, gpu_ids=[0])
        idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))
        _check_exact(idx, dist, q, bank, k, metric)
        rep = oracle.near_tie_report(idx, g[f"idx_{name}"], (-1 if metric != "dot_product" else 1) * g[f"dist_{name}"].astype(np.float64))
        assert rep["set_rate"] >= 0.99 and rep["excused_rate"] == 1.0, rep
        if name == "ip32":
            assert idx[0, :3].tolist() == [77, 1234, 4000]


def test_plugin_errors(cuda_device):
    fm = torch.from_numpy(gi.unit_bank(64, 16, seed=0))
    with pytest.raises(ValueError):
        NearestNeighborSearchHIP(fm, distance_measure="cosine")          # search_faiss.py:48
    with pytest.raises(ValueError):
        NearestNeighborSearchHIP(fm, gpu_ids=[99])                       # search_faiss.py:25
    nn = NearestNeighborSearchHIP(fm, n_neighbors=5, some_unknown_kwarg=1)   # **kwargs swallowed
    with pytest.raises(ValueError):
        nn.find_nearest_neighbors(fm[:2], k=2049)                        # beyond faiss-gpu's own limit of 2048


@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_mid_size_vs_float64_definition(cuda_device, metric):
    """300k x 128 bank, both metrics: fp32 chain result equals the float64 definition except at near-ties."""
    M, D, nq, k = 300_000, 128, 512, 30
    bank = gi.unit_bank(M, D, seed=21)
    if metric == "l2":
        bank = bank * (1.0 + 0.5 * np.random.default_rng(23).random((M, 1), dtype=np.float32))      # rows of unequal norm: L2 is not a re-ordered IP
    q = gi.vit_like_queries(nq, D, seed=22)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    _check_exact(idx, dist, q, bank, k, metric)
    i64, d64 = oracle.knn_f64(q, bank, k, metric)
    rep = oracle.near_tie_report(idx, i64, -d64 if metric == "l2" else d64)
    assert rep["excused_rate"] == 1.0 and rep["set_rate"] > 0.98, rep


@pytest.mark.parametrize("metric,fp16", [("dot_product", False), ("l2", False), ("dot_product", True)])
def test_exact_ties_against_an_independent_float64_definition(cuda_device, metric, fp16):
    """Small-integer rows with ~170 exact copies each: every score is exact in fp32 and float64, so the kernel's neighbours must equal
    torch-float64 scores ordered by (score, id ascending) -- computed here without oracle/ -- id for id, also across tile, slot and pass
    boundaries (the copies are scattered over 10,000 rows; use_fp16: such ties defeat the certificate and walk the whole escalation)."""
    rng = np.random.default_rng(19)
    M, D, nq, k = 10_000, 64, 300, 30
    base = rng.integers(-3, 4, size=(60, D)).astype(np.float32)
    bank = base[rng.integers(0, 60, size=M)]
    q = rng.integers(-2, 3, size=(nq, D)).astype(np.float32)
    b64, q64 = torch.from_numpy(bank).double(), torch.from_numpy(q).double()
    if metric == "dot_product":
        sc = (q64 @ b64.T).numpy(); key = -sc
    else:
        sc = torch.stack([((qq[None, :] - b64) ** 2).sum(-1) for qq in q64]).numpy(); key = sc
    ids = np.arange(M)
    want = np.stack([np.lexsort((ids, key[r]))[:k] for r in range(nq)])
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda()); ix.set_fp16(fp16)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(dist.cpu().numpy().astype(np.float64), np.take_along_axis(sc, want, axis=1))


@pytest.mark.parametrize("M,D,nq", [(2_074_072, 384, 12_544), (1_000_000, 768, 21_904)])
def test_full_batch_properties_at_scale(cuda_device, M, D, nq):
    """cfg-2 bank (full size) and a cfg-3-shaped batch: size-independent properties + an oracle spot check."""
    k = 30
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M)
    ix = HipFlatIndex(D, 0, 0)
    ix.reserve(M)
    for r in range(0, M, 500_000):
        n = min(500_000, M - r)
        ix.add(torch.randn((n, D), generator=g, device=dev), normalize=True)
    assert ix.ntotal == M
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    # plant: query j (j < 256) is a scaled copy of bank row 7919*j -> that row must be its best neighbour
    planted = torch.arange(256, device=dev) * 7919 % M
    q[:256] = 4.0 * ix.reconstruct(planted)
    idx, dist = ix.search(q, k)
    assert (idx[:256, 0] == planted).all()
    assert torch.allclose(dist[:256, 0], torch.full((256,), 4.0, device=dev), atol=1e-4)
    assert (dist[:, :-1] >= dist[:, 1:]).all()                                  # sortedness
    assert (idx >= 0).all() and (idx < M).all()
    srt = idx.sort(dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all()                                    # no duplicates in a row
    # idempotence / determinism: same call, same bits
    idx2, dist2 = ix.search(q, k)
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
    # sharding invariance: two half banks + merge == one bank
    half = M // 2
    ids = torch.arange(M, device=dev)
    a, b = HipFlatIndex(D, 0, 0), HipFlatIndex(D, 0, 0)
    for lo in range(0, M, 500_000):
        hi = min(M, lo + 500_000)
        rows = ix.reconstruct(ids[lo:hi])
        if lo < half:
            a.add(rows[: max(0, min(hi, half) - lo)])
        if hi > half:
            b.add(rows[max(0, half - lo):])
    ia, da = a.search(q, k, id_base=0)
    ib, db = b.search(q, k, id_base=half)
    from hbird_mi.nn.search_hip import merge_topk
    im, dm = merge_topk(torch.stack([da, db]), torch.stack([ia, ib]), 0)
    assert torch.equal(im, idx) and torch.equal(dm, dist)
    # oracle spot check on 48 random queries against the whole bank
    sel = torch.randperm(nq, generator=torch.Generator().manual_seed(1))[:48]
    bank = ix.reconstruct(ids).cpu().numpy()
    _check_exact(idx[sel.to(dev)], dist[sel.to(dev)], q[sel.to(dev)].cpu().numpy(), bank, k, "dot_product")


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        M = int(rng.choice([1, 31, 255, 256, 257, 1000, 4097, 20000]))
        D = int(rng.choice([1, 7, 8, 17, 64, 100, 384, 768, 1024]))
        nq = int(rng.choice([1, 32, 255, 256, 257, 700]))
        k = int(rng.choice([1, 2, 29, 32, 33, 64, 65, 130, 256]))
        metric = str(rng.choice(["dot_product", "l2"]))
        fp16 = bool(rng.integers(0, 2)) and k <= 128
        G = int(rng.choice([0, 1, 7, 64, 300]))
        panel = int(rng.choice([0, 1, 3, 16]))
        cases.append((M, D, nq, k, metric, fp16, G, panel, 1000 + i))
    return cases


@pytest.mark.parametrize("M,D,nq,k,metric,fp16,G,panel,seed", _random_cases(24, seed=2026))
def test_seeded_random_shapes_bit_exact(cuda_device, M, D, nq, k, metric, fp16, G, panel, seed):
    """A seeded sweep over ragged sizes (rows / dims / queries around the 256 and 32 tile edges), k on both sides of the
    LDS-list and pool limits, both metrics, fp16 mode, and arbitrary work partitions: always the oracle's bits."""
    rng = np.random.default_rng(seed)
    bank = gi.unit_bank(M, D, seed=seed)
    if M > 40:
        bank[rng.integers(0, M, size=5)] = bank[0]          # a few exact ties
    q = gi.vit_like_queries(nq, D, seed=seed + 1)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    half = M // 2
    ix.add(bank[:half]); ix.add(torch.from_numpy(bank[half:]).cuda())     # host + device appends
    ix.set_fp16(fp16)
    ix.set_tuning(G, panel)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k, id_base=7)
    _check_exact(idx, dist, q, bank, k, metric, id_base=7)


@pytest.mark.parametrize("M,D,nq,k,metric", [
    (3000, 32, 70, 33, "dot_product"),
    (5000, 64, 300, 64, "dot_product"),
    (20000, 128, 260, 90, "dot_product"),     # cfg-5's k
    (20000, 128, 260, 90, "l2"),
    (4000, 48, 100, 200, "dot_product"),
    (300, 16, 20, 256, "dot_product"),        # k = HB_MAX_K, fewer rows than... no: 300 rows > 256
    (100, 16, 20, 128, "dot_product"),        # fewer rows than k
])
def test_wide_k_bit_exact(cuda_device, M, D, nq, k, metric):
    """k > 32: the per-query lists live in global memory instead of LDS; same results."""
    bank = gi.unit_bank(M, D, seed=M + k)
    bank[M // 2] = bank[3]                    # a tie
    q = gi.vit_like_queries(nq, D, seed=nq + k)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(bank)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    ix.set_tuning(5, 2)                       # several segments / slots per query tile
    idx, dist = ix.search(q, k)
    _check_exact(idx, dist, q, bank, k, metric)


@pytest.mark.parametrize("k,metric,fp16", [(40, "dot_product", False), (100, "dot_product", False), (64, "l2", False),
                                           (30, "dot_product", True)])
def test_candidate_pool_exact_ties_across_compactions(cuda_device, k, metric, fp16):
    """k > 32 keeps unsorted candidate pools that are compacted whenever they fill up (by bisection; by an exact radix select when the scores cannot be separated): hundreds of
    bit-identical scores (duplicate rows, zero rows under L2: scores of -0/+0) must still leave by ascending id."""
    M, D, nq = 9000, 32, 70
    bank = gi.unit_bank(M, D, seed=21)
    rng = np.random.default_rng(22)
    dup = rng.choice(M, size=700, replace=False)
    bank[dup[:400]] = bank[dup[0]]             # 400 copies of one row
    bank[dup[400:]] = 0.0                      # 300 zero rows
    q = gi.vit_like_queries(nq, D, seed=23)
    q[:10] = 3.0 * bank[dup[0]]                # these queries' top-400 are all ties
    q[10:14] = 0.0                             # every score is (+-)0 for IP
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(bank)
    ix.set_fp16(fp16)
    for G, panel in ((0, 0), (3, 2)):
        ix.set_tuning(G, panel)
        idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
        _check_exact(idx, dist, q, bank, k, metric)


def test_wide_k_aggregate(cuda_device):
    M, D, C, nq, k = 20000, 128, 19, 300, 90
    bank = gi.unit_bank(M, D, seed=1); lab = gi.labels_from_masks(M, C, 196, seed=2)
    q = gi.vit_like_queries(nq, D, seed=3)
    ix = HipFlatIndex(D, 0, 0); ix.add(bank); ix.add_labels(lab); ix.set_num_classes(C)
    out = ix.search_aggregate(torch.from_numpy(q).cuda(), k).cpu().numpy()
    ridx, _ = oracle.knn_chain_f32(q, bank, k)
    kf, kl = oracle.gather_neighbours(ridx, bank, lab, 1, nq)
    assert np.abs(out - oracle.cross_attention(q[None], kf, kl)[0]).max() < 2e-5


def test_removed_kernel_variants_are_rejected(cuda_device):
    """Variants 1 (4-wave fp32 kernel), 2 (first fp16 design) and 5 (16x16x32 fp16 kernel) left the library in round 4 (same bits, not faster)."""
    ix = HipFlatIndex(32, 0, 0)
    for v in (1, 2, 5, 7, -1):
        with pytest.raises(RuntimeError):
            ix.set_variant(v)
    ix.set_variant(3); ix.set_variant(0)


@pytest.mark.parametrize("variant", [3, 4, 6])   # 6: small searches on sorted LDS lists (the default runs them on pools since round 3)
@pytest.mark.parametrize("M,D,nq,k,metric", [(5000, 64, 300, 30, "dot_product"), (20000, 384, 520, 30, "l2"), (7000, 96, 257, 5, "dot_product"),
                                              (3000, 32, 100, 90, "dot_product"), (9000, 128, 300, 200, "l2"), (4000, 40, 64, 30, "dot_product")])
def test_register_resident_query_fragment_kernel_bit_exact(cuda_device, variant, M, D, nq, k, metric):
    """hb_index_set_variant(3) forces the big-search fp32 kernel (query fragments straight into registers, hbird_knn_bd.hip) on
    any search it can run -- D padded to a multiple of 32; D = 40 pads to 40 and must fall back by itself --, 4 forbids it: either
    way the oracle's bits, on the lists (k <= 32) and on the candidate pools (k > 32), with odd work-list splits."""
    bank = gi.unit_bank(M, D, seed=1)
    q = gi.vit_like_queries(nq, D, seed=2)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(bank)
    ix.set_variant(variant)
    for G, panel in ((0, 0), (5, 2), (256, 1)):
        ix.set_tuning(G, panel)
        idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
        _check_exact(idx, dist, q, bank, k, metric)


def test_reference_named_backends(cuda_device):
    """hbird_mi.nn.search_faiss / search_scann carry the reference's class names and keyword surfaces."""
    from hbird_mi.nn.search_faiss import NearestNeighborSearchFaiss
    from hbird_mi.nn.search_scann import NearestNeighborSearchScaNN
    bank = gi.unit_bank(3000, 32, seed=1); q = gi.vit_like_queries(50, 32, seed=2)
    fm = torch.from_numpy(bank)
    nn = NearestNeighborSearchFaiss(fm, n_neighbors=10, distance_measure="l2", idx_shard=False, use_fp16=True, gpu_ids=[0])
    idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))
    _check_exact(idx, dist, q, bank, 10, "l2")
    sc = NearestNeighborSearchScaNN(fm, n_neighbors=10, num_leaves=100, num_leaves_to_search=10)
    idx, dist = sc.find_nearest_neighbors(torch.from_numpy(q), k=3)      # k is ignored, like the reference
    assert idx.shape == (50, 10)
    _check_exact(idx, dist, q, bank, 10, "dot_product")
    with pytest.raises(ValueError):
        NearestNeighborSearchScaNN(fm, distance_measure="l2")            # search_scann.py:19-20


def test_headline_size_10m_x_768(cuda_device):
    """BASELINE.json's headline shape (10 M x 768 bank, 21,904-query batch, k = 30): planted neighbours, sortedness, determinism,
    sharding invariance, the fp32 chain oracle BIT FOR BIT on 512 queries against all rows (the automatic clustered kernel that the
    bench times, and the use_fp16 path) and an independent float64 check of 16 queries."""
    M, D, nq, k = 10_000_000, 768, 21_904, 30
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    ix = HipFlatIndex(D, 0, 0)
    ix.reserve(M)
    half = M // 2
    a, b = HipFlatIndex(D, 0, 0), HipFlatIndex(D, 0, 0)
    a.reserve(half); b.reserve(M - half)
    for r in range(0, M, 500_000):
        rows = torch.randn((500_000, D), generator=g, device=dev)
        rows = rows / rows.norm(dim=1, keepdim=True)
        ix.add(rows)
        (a if r < half else b).add(rows)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    planted = torch.arange(128, device=dev) * 78_125 + 11
    q[:128] = 2.5 * ix.reconstruct(planted)
    idx, dist = ix.search(q, k)
    assert (idx[:128, 0] == planted).all()
    assert (dist[:, :-1] >= dist[:, 1:]).all() and (idx >= 0).all() and (idx < M).all()
    srt = idx.sort(dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all()
    idx2, dist2 = ix.search(q, k)
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
    from hbird_mi.nn.search_hip import merge_topk
    ia, da = a.search(q, k, id_base=0)
    ib, db = b.search(q, k, id_base=half)
    im, dm = merge_topk(torch.stack([da, db]), torch.stack([ia, ib]), 0)
    assert torch.equal(im, idx) and torch.equal(dm, dist)
    del a, b
    # use_fp16 mode (fp16 candidate pass + certified exact re-rank) must return the very same bits
    ix.set_fp16(True)
    idx16, dist16 = ix.search(q, k)
    assert torch.equal(idx16, idx) and torch.equal(dist16, dist)
    assert ix.last_fp16_fallbacks() < nq // 100
    ix.set_fp16(False)
    # the chain oracle at full size: the bank comes back in 1 M-row chunks, each searched by the oracle with its id base, merged on the host
    from helpers import chain_oracle_topk_chunked
    # both forms of the big fp32 search -- in 2 x 4 L2-sharing clusters and without (the index keeps whichever measures faster on the box)
    for shape in ((2, 4, -1), (1, 1, 0)):
        ix.set_cluster(*shape)
        ic, dc = ix.search(q, k)
        assert tuple(ix.schedule_info()["cluster"]) == shape[:2]
        assert torch.equal(ic, idx) and torch.equal(dc, dist), shape
    ix.set_cluster(0, 0, -1)
    sel32 = torch.linspace(0, nq - 1, 512, device=dev).long()      # (512 queries, 2.3 % of the batch: the bank chunks' trip to the host is most of what the check costs)
    ci, cd = chain_oracle_topk_chunked(ix, q[sel32], M, k)
    for name, (gi_, gd_) in (("fp32", (idx, dist)), ("use_fp16", (idx16, dist16))):
        assert np.array_equal(gi_[sel32].cpu().numpy(), ci), f"{name}: indices differ from the chain oracle at 10 M x 768"
        assert np.array_equal(gd_[sel32].cpu().numpy().view(np.uint32), cd.view(np.uint32)), f"{name}: distance bits differ"
    # float64 scores of 16 queries against every bank row (chunked reconstruct), exact top-k by (score, id)
    sel = torch.tensor([0, 5, 127, 128, 1000, 5000, 9999, 12345, 15000, 17000, 19000, 20000, 21000, 21500, 21900, 21903],
                       device=dev)
    qs = q[sel].double()
    best_s = torch.full((16, k), -float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.full((16, k), -1, dtype=torch.int64, device=dev)
    for r in range(0, M, 1_000_000):
        ids = torch.arange(r, min(M, r + 1_000_000), device=dev)
        sc = qs @ ix.reconstruct(ids).double().T                      # [16, chunk]
        cs = torch.cat([best_s, sc], dim=1); ci = torch.cat([best_i, ids[None].expand(16, -1)], dim=1)
        top = cs.topk(k, dim=1)
        best_s, best_i = top.values, ci.gather(1, top.indices)
    got = idx[sel].cpu().numpy(); ref = best_i.cpu().numpy()
    rep = oracle.near_tie_report(got, ref, best_s.cpu().numpy())
    assert rep["excused_rate"] == 1.0 and rep["set_rate"] >= 0.9, rep
    # fp32 chain rounding: ~1e-7 * sum|q_k b_k| (a few 1e-5 at |q| ~ 83, D = 768)
    assert np.abs(dist[sel].cpu().numpy() - best_s.cpu().numpy()).max() < 1e-4


@pytest.mark.parametrize("M,D,nq,k,metric", [
    (5000, 64, 300, 30, "dot_product"),
    (20000, 384, 520, 30, "dot_product"),
    (20000, 384, 520, 30, "l2"),
    (9000, 768, 300, 10, "dot_product"),
    (3000, 40, 100, 90, "dot_product"),        # D padded to 64, k' = 184
    (100, 16, 20, 30, "dot_product"),          # fewer rows than k'
])
def test_use_fp16_candidate_pass_with_exact_rerank(cuda_device, M, D, nq, k, metric):
    """use_fp16 (search_faiss.py:40): fp16 MFMA candidate pass (k' >= 2k) + fp32 chain re-rank -> the fp32 answer."""
    bank = gi.unit_bank(M, D, seed=M + 1)
    q = gi.vit_like_queries(nq, D, seed=nq + 1)
    nn = NearestNeighborSearchHIP(torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, use_fp16=True, gpu_ids=[0])
    idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))      # the plugin's mode: the candidate pass only where it pays
    _check_exact(idx, dist, q, bank, k, metric)                     # (rows x queries x D >= 1.5e10 (k' / 64)^2; the fp32 kernel for most of these)
    ix = nn.index
    ix.set_fp16(True)                                               # always: these small banks go through the fp16 kernels
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    ix.set_tuning(6, 2)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    ix.add(bank[:100] * 0.5)                    # appending after a search re-converts the touched tiles
    idx, dist = ix.search(q, k)
    _check_exact(idx, dist, q, np.concatenate([bank, bank[:100] * 0.5]), k, metric)
    ix.set_tuning(0, 0)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, np.concatenate([bank, bank[:100] * 0.5]), k, metric)


@pytest.mark.parametrize("M,D,nq,k,metric", [
    (6000, 50, 200, 30, "dot_product"),        # D = 50: query rows 8 B aligned, the last 32-k chunk holds 18 values
    (6000, 33, 130, 90, "l2"),                 # one value in the last chunk; k' = 184: three batches of candidates
    (9000, 96, 150, 128, "dot_product"),       # k' = 256
    (20000, 384, 300, 30, "l2"),
    (40, 24, 70, 30, "dot_product"),           # fewer rows than k
])
def test_use_fp16_rerank_on_the_row_major_copy(cuda_device, M, D, nq, k, metric):
    """The exact re-rank of use_fp16 searches reads a second, row-major fp32 copy of the bank when there is one (hb_index_set_rerank_copy:
    automatic / always / never) and the fragment tiles otherwise: the oracle's bits either way, also after rows were appended behind a
    partly filled row tile and after a reset."""
    bank = gi.unit_bank(M, D, seed=M + 7)
    q = gi.vit_like_queries(nq, D, seed=nq + 7)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(True)
    qd = torch.from_numpy(q).cuda()
    ix.set_rerank_copy(2)
    i0, d0 = ix.search(qd, k)
    assert ix.rerank_copy_bytes() == 0
    _check_exact(i0, d0, q, bank, k, metric)
    ix.set_rerank_copy(1)
    i1, d1 = ix.search(qd, k)
    assert ix.rerank_copy_bytes() >= M * D * 4
    assert torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
    more = gi.unit_bank(77, D, seed=5) * 1.5
    ix.add(torch.from_numpy(more).cuda())       # 77 rows behind a partly filled row tile: the copy follows
    both = np.concatenate([bank, more])
    i2, d2 = ix.search(qd, k)
    _check_exact(i2, d2, q, both, k, metric)
    ix.set_rerank_copy(2)                       # released
    assert ix.rerank_copy_bytes() == 0
    i3, d3 = ix.search(qd, k)
    assert torch.equal(i2, i3) and torch.equal(d2.view(torch.int32), d3.view(torch.int32))
    ix.set_rerank_copy(0)                       # automatic: a small bank on an empty device gets its copy
    ix.reset()
    ix.add(torch.from_numpy(more).cuda())
    i4, d4 = ix.search(qd, k)
    assert ix.rerank_copy_bytes() > 0
    _check_exact(i4, d4, q, more, k, metric)


def test_use_fp16_certificate_and_exact_fallback(cuda_device):
    """Near-duplicate bank rows cannot be ranked by fp16 scores: the per-query certificate must fail for them and the
    exact fp32 re-search must deliver the fp32 answer anyway; well-separated queries stay on the fast path."""
    M, D, k = 6000, 64, 30
    rng = np.random.default_rng(5)
    bank = gi.unit_bank(M, D, seed=3)
    centre = bank[10].copy()
    for r in range(200, 500):                      # 300 rows within 1e-4 of one direction
        v = centre + 1e-4 * rng.standard_normal(D).astype(np.float32)
        bank[r] = v / np.linalg.norm(v)
    q = gi.vit_like_queries(300, D, seed=4)
    q[:40] = 4.0 * centre + 1e-3 * rng.standard_normal((40, D)).astype(np.float32)   # queries aimed at the cluster
    ix = HipFlatIndex(D, 0, 0)
    ix.add(bank)
    ix.set_fp16(True)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, "dot_product")
    nfb = ix.last_fp16_fallbacks()
    assert 40 <= nfb < 150, nfb                    # the 40 cluster queries (at least) were re-searched exactly
    idx, dist = ix.search(q[40:], k)               # host path, only ordinary queries
    _check_exact(idx, dist, q[40:], bank, k, "dot_product")
    assert ix.last_fp16_fallbacks() <= nfb - 40
    ix.set_fp16(False)
    i32, d32 = ix.search(q, k)
    _check_exact(i32, d32, q, bank, k, "dot_product")


@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_use_fp16_escalation_second_pass_then_fp32(cuda_device, metric):
    """What happens to a query whose first certificate fails (hb_index_set_fp16_escalation): a second fp16 pass with k' = 256 candidates,
    seeded with the floor (exact k-th best - 1.001 E), and only then the fp32 kernel, seeded with the exact k-th best found so far.
    Planted near-duplicate clusters force both: 150 rows within 1e-4 of one direction are wider than the first pass's k' = 64 but fit the
    second pass's list (certified there: the list does not fill up); 400 such rows are wider than k' = 256 too and reach the fp32 kernel.
    Always the fp32 search's bits -- with the escalation on, off, and against the oracle; ordinary queries never leave the first pass."""
    M, D, k = 60_000, 128, 30
    rng = np.random.default_rng(11)
    bank = gi.unit_bank(M, D, seed=13)
    c1, c2 = bank[7].copy(), bank[8].copy()
    for r in range(1000, 1150):
        v = c1 + 1e-4 * rng.standard_normal(D).astype(np.float32); bank[r] = v / np.linalg.norm(v)
    for r in range(30_000, 30_400):
        v = c2 + 1e-4 * rng.standard_normal(D).astype(np.float32); bank[r] = v / np.linalg.norm(v)
    bank[40_000:40_003] = bank[1000]                     # exact duplicates: ties by id across the passes
    q = gi.vit_like_queries(700, D, seed=14)
    q[:60] = 4.0 * c1 + 1e-3 * rng.standard_normal((60, D)).astype(np.float32)
    q[60:100] = 4.0 * c2 + 1e-3 * rng.standard_normal((40, D)).astype(np.float32)
    qd = torch.from_numpy(q).cuda()
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ref_i, ref_d = ix.search(qd, k)                      # the fp32 kernel
    _check_exact(ref_i, ref_d, q, bank, k, metric)
    ix.set_fp16(True)
    for rerank_copy in (1, 2):
        ix.set_rerank_copy(rerank_copy)
        ix.set_fp16_escalation(True)
        i1, d1 = ix.search(qd, k)
        esc, fb = ix.last_fp16_escalated(), ix.last_fp16_fallbacks()
        assert torch.equal(i1, ref_i) and torch.equal(d1.view(torch.int32), ref_d.view(torch.int32))
        assert 100 <= esc < 200 and 40 <= fb <= esc - 55, (esc, fb)      # both clusters escalate; the narrow one is settled by the second pass
        ix.set_fp16_escalation(False)
        i2, d2 = ix.search(qd, k)
        assert torch.equal(i2, ref_i) and torch.equal(d2.view(torch.int32), ref_d.view(torch.int32))
        assert ix.last_fp16_escalated() == ix.last_fp16_fallbacks() == esc, (ix.last_fp16_escalated(), ix.last_fp16_fallbacks(), esc)
    ix.set_fp16_escalation(True)
    i3, d3 = ix.search(qd[100:], k)                      # ordinary queries only
    assert torch.equal(i3, ref_i[100:]) and torch.equal(d3.view(torch.int32), ref_d[100:].view(torch.int32))
    assert ix.last_fp16_escalated() <= 6 and ix.last_fp16_fallbacks() <= ix.last_fp16_escalated()
    i4, d4 = ix.search(qd[:100], 90)                     # k' = 184 first, then 256
    ix.set_fp16(False)
    r4i, r4d = ix.search(qd[:100], 90)
    assert torch.equal(i4, r4i) and torch.equal(d4.view(torch.int32), r4d.view(torch.int32))



@pytest.mark.parametrize("k,fp16", [(30, False), (32, False), (90, False), (256, False), (30, True), (100, True)])
def test_few_queries_against_a_big_bank_two_level_merge(cuda_device, k, fp16):
    """One query tile against many bank tiles: every workgroup holds a partial list of the same queries (up to 256
    lists per query), merged in two levels."""
    M, D, nq = 300_000, 64, 100
    bank = gi.unit_bank(M, D, seed=8)
    q = gi.vit_like_queries(nq, D, seed=9)
    ix = HipFlatIndex(D, 0, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(fp16)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    assert ix.schedule_info()["max_slots_per_qtile"] == 256
    _check_exact(idx, dist, q, bank, k, "dot_product")


@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_repeatable_and_invariant_under_1_2_4_8_way_sharding(cuda_device, metric):
    """The slots exchange threshold floors while they run, so which scores are filtered early depends on timing -- the
    answer must not: ten searches return the same bits, and so does the bank cut into 2, 4 or 8 row shards (each its
    own index with successive ids; per-shard ORDERING scores merged by hb_merge_topk, then converted -- the multi-GPU
    path on one device.  Merging squared L2 distances instead would reorder rows whose distances round equal)."""
    from hbird_mi.nn.search_hip import merge_topk
    M, D, nq, k = 120_000, 64, 600, 30
    bank = gi.unit_bank(M, D, seed=77)
    bank[50_000:50_040] = bank[7]                        # ties that straddle shard boundaries
    bank[119_990:] = bank[7]
    q = gi.vit_like_queries(nq, D, seed=78)
    q[:20] = 3.0 * bank[7]
    m = 0 if metric == "dot_product" else 1
    qd = torch.from_numpy(q).cuda()
    ix = HipFlatIndex(D, m, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_tuning(64, 8)                                 # many segments and slots per query tile
    ref_i, ref_d = ix.search(qd, k)
    _check_exact(ref_i, ref_d, q, bank, k, metric)
    for _ in range(9):
        i2, d2 = ix.search(qd, k)
        assert torch.equal(i2, ref_i) and torch.equal(d2.view(torch.int32), ref_d.view(torch.int32))
    for parts in (2, 4, 8):
        per = (M + parts - 1) // parts
        idxs, dists = [], []
        for p in range(parts):
            lo, hi = p * per, min(M, (p + 1) * per)
            sh = HipFlatIndex(D, m, 0)
            sh.add(torch.from_numpy(bank[lo:hi]).cuda())
            i, sc = sh.search_scores(qd, k, id_base=lo)
            idxs.append(i); dists.append(sc)
        im, dm = merge_topk(torch.stack(dists), torch.stack(idxs), 0)
        dm = ix.distances_from_scores(qd, dm.contiguous())
        assert torch.equal(im, ref_i), parts
        assert torch.equal(dm.view(torch.int32), ref_d.view(torch.int32)), parts
    # score output: inner product = the distance itself; L2 = q.b - |b|^2 / 2, ordered like the distances
    i3, s3 = ix.search_scores(qd, k)
    assert torch.equal(i3, ref_i)
    if m == 0:
        assert torch.equal(s3, ref_d)
    else:
        assert (s3[:, :-1] >= s3[:, 1:]).all()
        assert torch.equal(ix.distances_from_scores(qd, s3.clone()).view(torch.int32), ref_d.view(torch.int32))
    i4, d4 = ix.search(qd, k)                              # the mode does not stick
    assert torch.equal(d4.view(torch.int32), ref_d.view(torch.int32))


def test_phased_and_unphased_pool_searches_return_the_same_bits(cuda_device):
    """Pool searches (k > 32, use_fp16, small fp32 searches) are launched in phases whose boundaries hand every pool the union's k-th best
    as a floor (hb_launch_knn); hb_index_set_search_options(ix, 0, ...) runs them in one launch, variant 6 keeps small fp32 searches on
    the sorted LDS lists, a small-search limit of 1 stage sends everything to the big-search instantiations.  Same ids and distances,
    bit for bit, in all four."""
    bank = gi.unit_bank(70001, 64, seed=5); q = torch.from_numpy(gi.vit_like_queries(1301, 64, seed=6)).cuda()
    results = []
    for setup in ("default", "unphased", "lists", "never_small"):
        out = []
        for k, fp16 in ((30, False), (90, False), (30, True)):
            ix = HipFlatIndex(64, 0, 0); ix.add(torch.from_numpy(bank).cuda()); ix.set_fp16(fp16)
            if setup == "unphased":
                ix.set_search_options(phases=False)
            elif setup == "lists":
                ix.set_variant(6)
            elif setup == "never_small":
                ix.set_search_options(small_limit_stages=1)
            out.append(ix.search(q, k))
        results.append(out)
    for other in results[1:]:
        for (i0, d0), (i1, d1) in zip(results[0], other):
            assert torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
    with pytest.raises(RuntimeError):
        HipFlatIndex(8, 0, 0).set_search_options(small_limit_stages=-1)


def test_per_xcd_work_shares_never_change_the_result(cuda_device):
    """hb_index_set_xcd_weights: big fp32 searches give each XCD group (blocks equal mod 8) a share of the work list that follows its measured
    speed (calibrated between searches from the workgroups' own time stamps, mode 0), equal shares (1) or given ones (2).  Speed only: ids and
    distance bits are the same in every mode, also with wildly uneven shares and under the 2 x 4 clusters."""
    M, D, nq, k = 2_000_000, 768, 21_904, 30
    dev = torch.device("cuda:0")
    ix = HipFlatIndex(D, 0, 0); ix.reserve(M)
    g = torch.Generator(device=dev).manual_seed(77)
    for r in range(0, M, 500_000):
        rows = torch.randn((500_000, D), generator=g, device=dev)
        ix.add(rows, normalize=True)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ix.set_xcd_weights(1)
    ref_i, ref_d = ix.search(q, k)
    assert ix.xcd_weights()[0] == [1.0] * 8
    for w8, cluster in (([1.2, 0.85, 1.1, 0.9, 1.0, 1.05, 0.95, 1.15], (0, 0, -1)), ([0.9, 1.1, 0.9, 1.1, 0.9, 1.1, 0.9, 1.1], (2, 4, 16))):
        ix.set_xcd_weights(2, w8); ix.set_cluster(*cluster)
        i1, d1 = ix.search(q, k)
        assert torch.equal(i1, ref_i) and torch.equal(d1.view(torch.int32), ref_d.view(torch.int32)), (w8, cluster)
    ix.set_cluster(0, 0, -1)
    ix.set_xcd_weights(0)
    for _ in range(3):
        i1, d1 = ix.search(q, k)
        torch.cuda.synchronize()                       # (the stamps of this search have landed when the next one looks for them)
        assert torch.equal(i1, ref_i) and torch.equal(d1.view(torch.int32), ref_d.view(torch.int32))
    w, rounds = ix.xcd_weights()
    assert rounds >= 1 and abs(sum(w) / 8 - 1.0) < 1e-6 and all(0.8 <= v <= 1.25 for v in w), (w, rounds, ix.xcd_stats())
    print("calibrated shares", [round(v, 4) for v in w], "after", rounds, "rounds")
    # a PHASED search on a long list (k = 90: pools; 2,600 pairs per workgroup): its phase cuts follow the shares (hb_finish_schedule)
    ix.set_xcd_weights(1)
    ref_i, ref_d = ix.search(q[:12_544], 90)
    for w8 in ([1.2, 0.85, 1.1, 0.9, 1.0, 1.05, 0.95, 1.15], [0.97, 1.03, 0.98, 1.02, 0.99, 1.01, 1.0, 1.0]):
        ix.set_xcd_weights(2, w8)
        i1, d1 = ix.search(q[:12_544], 90)
        assert torch.equal(i1, ref_i) and torch.equal(d1.view(torch.int32), ref_d.view(torch.int32)), w8
    # the fp16 candidate kernel has shares of its own (use_fp16: the fp32 search's bits, whatever the shares)
    ix.set_xcd_weights(1)
    f32_i, f32_d = ix.search(q, k)
    ix.set_fp16(True)
    ix.set_xcd_weights(2, [1.2, 0.85, 1.1, 0.9, 1.0, 1.05, 0.95, 1.15])
    i1, d1 = ix.search(q, k)
    assert torch.equal(i1, f32_i) and torch.equal(d1.view(torch.int32), f32_d.view(torch.int32))
    ix.set_xcd_weights(0)
    for _ in range(4):
        i1, d1 = ix.search(q, k)
        torch.cuda.synchronize()
        assert torch.equal(i1, f32_i) and torch.equal(d1.view(torch.int32), f32_d.view(torch.int32))
        assert ix.last_fp16_fallbacks() < nq // 100
    w16, rounds16 = ix.xcd_weights(True)
    assert rounds16 >= 1 and abs(sum(w16) / 8 - 1.0) < 1e-6 and all(0.8 <= v <= 1.25 for v in w16), (w16, rounds16, ix.xcd_stats(True))
    print("fp16 candidate kernel: calibrated shares", [round(v, 4) for v in w16], "after", rounds16, "rounds")
    ix.set_fp16(False)
    with pytest.raises(RuntimeError):
        ix.set_xcd_weights(2, [1.0] * 7 + [9.0])


@pytest.mark.parametrize("D,k,metric,variant", [(768, 30, "dot_product", 0), (384, 8, "l2", 0), (768, 32, "dot_product", 4)])
def test_small_search_list_kernel_against_the_oracle(cuda_device, D, k, metric, variant):
    """Searches of 120 k - 400 k stages per workgroup run on the small-search LIST kernel (cold start, register-queue insertions, quota floors
    exchanged every tile at first and every 16th later: hbird_knn_bd.hip; variant 4: the LDS-staged form).  With 21,904 queries that is a
    1.25 - 3 M-row bank; here eight workgroups reach the same stage count on 500 k rows, so that the oracle can check EVERY query, twice
    (other floors arrive at other times: the bits must not care)."""
    dev = torch.device("cuda:0")
    M, nq = 500_000 if D == 768 else 800_000, 2048
    rng = np.random.default_rng(77 + D + k)
    bank = rng.standard_normal((M, D), dtype=np.float32); bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    bank[rng.integers(0, M, size=64)] = bank[5]                   # ties by id
    q = (3.0 * rng.standard_normal((nq, D))).astype(np.float32)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).to(dev)); ix.set_tuning(8, 0); ix.set_variant(variant)
    qd = torch.from_numpy(q).to(dev)
    idx, dist = ix.search(qd, k)
    info = ix.schedule_info()
    stages = info["query_tiles"] * info["bank_tiles"] // info["workgroups"] * (D // 8)
    assert 120_000 <= stages < 400_000 and tuple(info["cluster"]) == (1, 1), (stages, info)
    _check_exact(idx, dist, q, bank, k, metric)
    for _ in range(3):
        idx2, dist2 = ix.search(qd, k)
        assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
