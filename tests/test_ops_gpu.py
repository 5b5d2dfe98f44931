"""HIP kernels K1, K2, K3, K5, K6, K7 (through the C ABI) against the oracle and the golden fixtures."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import oracle
from hbird_mi import ops
from hbird_mi.nn.search_hip import HipFlatIndex
from hbird_mi.utils.eval_metrics import PredsmIoU

pytestmark = pytest.mark.gpu


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai); bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def test_k2_patch_label_hist_golden_bit_exact(cuda_device, golden_dir):
    g = np.load(f"{golden_dir}/g12_patchify_softlabels.npz")
    for name in "abc":
        for C in (21, 151):
            y, ps = g[f"y_{name}_{C}"], int(g[f"ps_{name}_{C}"])
            out = ops.patch_label_hist(torch.from_numpy(y).cuda(), ps, C).cpu().numpy()
            ref = g[f"label_{name}_{C}"]
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_k2_map255_and_big_shapes(cuda_device):
    y = gi.random_masks(3, 518, 518, 151, seed=4, with_255=True)
    y0 = y.copy(); y0[y0 == 255] = 0                                   # hbird_eval.py:310
    ref = oracle.patch_label_hist(y0, 14, 151)
    out = ops.patch_label_hist(torch.from_numpy(y).cuda(), 14, 151, map255=True).cpu().numpy()
    assert out.shape == (3, 37, 37, 151)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert np.allclose(out.sum(-1), 1.0, atol=1e-6)
    with pytest.raises(Exception):
        ops.patch_label_hist(torch.from_numpy(y).cuda(), 15, 151)      # 518 % 15 != 0


@pytest.mark.parametrize("n,D", [(1000, 384), (333, 768), (70, 20), (4096, 1024)])
def test_k1_normalize_append(cuda_device, n, D):
    x = gi.vit_like_queries(n, D, seed=n)
    ref = oracle.normalize_rows(x)
    ix = HipFlatIndex(D, 0, 0)
    ix.add(torch.from_numpy(x[: n // 3]).cuda(), normalize=True)       # ragged appends
    ix.add(torch.from_numpy(x[n // 3:]).cuda(), normalize=True)
    got = ix.reconstruct(np.arange(n))
    assert _ulp_diff(got, ref).max() <= 1      # double-accumulated norm: differs from the oracle by rounding order only
    assert (got == ref).mean() > 0.99
    got2 = ops.normalize_rows(torch.from_numpy(x).cuda()).cpu().numpy()
    assert _ulp_diff(got2, ref).max() <= 1
    # un-normalised append is an exact copy
    ix2 = HipFlatIndex(D, 0, 0)
    ix2.add(x)
    assert np.array_equal(ix2.reconstruct(np.arange(n)), x)
    assert np.array_equal(ix2.reconstruct(torch.arange(n).cuda()).cpu().numpy(), x)


def test_k3_sampling_golden(cuda_device, golden_dir):
    g = np.load(f"{golden_dir}/g5_sample.npz")
    for name in "ab":
        ps, C, K, seed = g[f"cfg_{name}"].tolist()
        y, r = g[f"y_{name}"], g[f"r_{name}"]
        B = y.shape[0]
        lab = ops.patch_label_hist(torch.from_numpy(y).cuda(), ps, C)
        SS = lab.shape[1] * lab.shape[2]
        scores, nonempty, nz = ops.patch_scores(lab.view(B, SS, C))
        assert nz.cpu().tolist() == [SS] * B
        pt = oracle.patchify_gt(y, ps)
        _, ref_scores = oracle.sample_patches(pt, C, K, np.ones_like(r))
        assert np.array_equal(scores.cpu().numpy(), ref_scores)
        r_off = torch.arange(B, dtype=torch.int64) * SS
        sidx, noisy = ops.patch_select(scores, nonempty, torch.from_numpy(r).cuda(), r_off.cuda(), K, want_scores=True)
        assert np.array_equal(sidx.cpu().numpy(), g[f"sidx_{name}"])
        _, ref_noisy = oracle.sample_patches(pt, C, K, r)
        assert np.array_equal(noisy.cpu().numpy(), ref_noisy)
        feats = torch.from_numpy(g[f"feats_{name}"]).cuda()
        rows = (sidx + torch.arange(B, device="cuda")[:, None] * SS).reshape(-1)
        sf = ops.gather_rows(feats.reshape(B * SS, -1), rows).view(B, K, -1)
        assert np.array_equal(sf.cpu().numpy(), g[f"sfeat_{name}"])


def test_k5_aggregate_golden_g3(cuda_device, golden_dir):
    """_cross_attention fixture: neighbours are given, so feed them as a bank + explicit (idx, ip)."""
    g = np.load(f"{golden_dir}/g3_cross_attention.npz")
    for name in ("small", "vitS"):
        q, k, v, ref = g[f"q_{name}"], g[f"k_{name}"], g[f"v_{name}"], g[f"out_{name}"]
        B, N, K, D = k.shape
        C = v.shape[-1]
        ix = HipFlatIndex(D, 0, 0)
        ix.add(k.reshape(-1, D))
        ix.add_labels(v.reshape(-1, C))
        ix.set_num_classes(C)
        idx = torch.arange(B * N * K, dtype=torch.int64).view(B * N, K).cuda()
        ip = (q[:, :, None, :].astype(np.float64) * k.astype(np.float64)).sum(-1).astype(np.float32)
        out = ix.aggregate(torch.from_numpy(q.reshape(B * N, D)).cuda(), idx,
                           torch.from_numpy(ip.reshape(B * N, K)).cuda()).cpu().numpy().reshape(B, N, C)
        assert np.abs(out - ref).max() < 2e-5, np.abs(out - ref).max()
        assert np.abs(out - oracle.cross_attention(q, k, v)).max() < 2e-5


@pytest.mark.parametrize("P,C", [(196, 21), (256, 151), (4096, 19)])
def test_label_table_as_counts_returns_the_fp32_bits(cuda_device, golden_dir, P, C):
    """hb_index_set_label_denominator: label rows j / P (hbird_eval.py:319-320) stored as uint16 counts.  Everything that reads the table
    -- the fused search + aggregation, the aggregation alone, the label-sharded partial sums, gather_labels, a borrowed count table --
    returns the bits of the fp32 table (P = 4096 is beyond the kernel's LDS quotient table: in-place division); the G2 fixture's values
    (the reference's own one_hot means) convert without loss; a value that is no multiple of 1 / P fails the next read."""
    M, D, nq, k = 6000, 64, 500, 30
    bank = gi.unit_bank(M, D, seed=3)
    lab = gi.labels_from_masks(M, C, P, seed=4)
    q = torch.from_numpy(gi.vit_like_queries(nq, D, seed=5)).cuda()
    a, b = HipFlatIndex(D, 0, 0), HipFlatIndex(D, 0, 0)
    b.set_label_denominator(P)
    assert a.label_denominator == 0 and b.label_denominator == P
    for ix in (a, b):
        ix.add(torch.from_numpy(bank).cuda())
        ix.add_labels(torch.from_numpy(lab[:2500]).cuda()); ix.add_labels(lab[2500:])      # device and host rows, a growth in between
        ix.set_num_classes(C)
    la, ia, da = a.search_aggregate(q, k, want_neighbours=True)
    lb, ib, db = b.search_aggregate(q, k, want_neighbours=True)
    assert torch.equal(ia, ib) and torch.equal(la.view(torch.int32), lb.view(torch.int32))
    assert torch.equal(a.aggregate(q, ia, da).view(torch.int32), b.aggregate(q, ib, db).view(torch.int32))
    ids = torch.tensor([0, 5, M - 1, 17, 17, -1, M + 3], device="cuda")
    assert torch.equal(a.gather_labels(ids).view(torch.int32), b.gather_labels(ids).view(torch.int32))
    assert np.array_equal(b.gather_labels(np.arange(M)).view(np.uint32), lab.view(np.uint32))
    nrm = a.copy_norms()
    pa = a.aggregate_partial(q, ia, da, nrm); pb = b.aggregate_partial(q, ib, db, nrm)
    assert torch.equal(pa.view(torch.int32), pb.view(torch.int32))
    cnt = b.copy_label_counts()
    assert cnt.dtype == torch.int16 and tuple(cnt.shape) == (M, C)
    assert np.array_equal(cnt.cpu().numpy().view(np.uint16).astype(np.float32) / np.float32(P), lab)
    c = HipFlatIndex(D, 0, 0); c.set_label_count_table(cnt, nrm, P, 0)               # the replicated table of a sharded run
    assert torch.equal(c.aggregate(q, ia, da).view(torch.int32), la.view(torch.int32))
    with pytest.raises(RuntimeError, match="already holds label rows"):
        b.set_label_denominator(P + 1)
    bad = lab[:10].copy(); bad[3, 2] += 1e-4
    d = HipFlatIndex(D, 0, 0); d.set_label_denominator(P)
    d.add(torch.from_numpy(bank[:10]).cuda()); d.add_labels(torch.from_numpy(bad).cuda()); d.set_num_classes(C)
    with pytest.raises(RuntimeError, match="not multiples of 1 /"):
        d.search_aggregate(q[:4], 5)
    g = np.load(f"{golden_dir}/g12_patchify_softlabels.npz")                         # the reference's own values
    for name in "abc":
        ref = g[f"label_{name}_{C if C in (21, 151) else 21}"]
        ps = int(g[f"ps_{name}_{C if C in (21, 151) else 21}"])
        e = HipFlatIndex(8, 0, 0); e.set_label_denominator(ps * ps)
        rows = ref.reshape(-1, ref.shape[-1])
        e.add(torch.zeros((rows.shape[0], 8)).cuda()); e.add_labels(rows); e.set_num_classes(rows.shape[1])
        assert np.array_equal(e.gather_labels(np.arange(rows.shape[0])).view(np.uint32), rows.view(np.uint32))


def test_count_quotients_are_exact(cuda_device):
    """K5 turns a stored count j back into (float)j / (float)P with r = RN(1 / P), q' = RN(j r), q = fma(fma(-q', P, j), r, q') -- three
    instructions instead of a division.  Exhaustive over j for a spread of denominators (every P up to 64, the patch sizes in use, primes,
    powers of two, the last one the shortcut covers and the first beyond it): with ONE neighbour the softmax weight is exactly 1, so
    label_hat is the label value itself, bit for bit."""
    Ps = sorted(set(list(range(1, 65)) + [196, 256, 49, 64, 81, 100, 121, 144, 169, 225, 1024, 2039, 2047, 2048, 2049, 4096] + list(range(97, 2048, 131))))
    for P in Ps:
        j = np.arange(P + 1, dtype=np.float32)
        vals = (j / np.float32(P)).astype(np.float32)              # numpy divides in fp32: the values K2 stores
        C = 4 if P % 2 else 40                                     # few classes: the (neighbour group, class) form; many: the 16-byte gather
        lab = np.zeros((P + 1, C), dtype=np.float32); lab[:, 1] = vals; lab[:, 3] = vals[::-1]
        ix = HipFlatIndex(8, 0, 0)
        ix.set_label_denominator(P)
        bank = np.zeros((P + 1, 8), dtype=np.float32); bank[:, 0] = 1.0
        ix.add(torch.from_numpy(bank).cuda()); ix.add_labels(torch.from_numpy(lab).cuda()); ix.set_num_classes(C)
        q = torch.from_numpy(bank).cuda()
        idx = torch.arange(P + 1, device="cuda", dtype=torch.int64)[:, None].contiguous()
        dist = torch.ones((P + 1, 1), device="cuda")
        got = ix.aggregate(q, idx, dist).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), lab.view(np.uint32)), P


@pytest.mark.parametrize("multi", [False, True])
def test_label_storage_form_can_change_after_reset(cuda_device, multi):
    """reset() keeps allocations; a following set_label_denominator may switch between fp32 rows and uint16 counts -- in both
    directions the table of the other form must not be written through (a null / smaller buffer in the library, an int16 tensor
    silently truncating fp32 rows in HipMultiIndex).  With fewer rows than the old capacity, so that nothing regrows by itself."""
    from hbird_mi.nn.search_hip import HipMultiIndex
    D, C, P = 32, 21, 196
    bank = gi.unit_bank(3000, D, seed=3)
    lab = gi.labels_from_masks(3000, C, P, seed=4)
    q = torch.from_numpy(gi.vit_like_queries(200, D, seed=5)).cuda()
    ref = HipFlatIndex(D, 0, 0); ref.add(torch.from_numpy(bank[:2000]).cuda()); ref.add_labels(torch.from_numpy(lab[:2000]).cuda()); ref.set_num_classes(C)
    want = ref.search_aggregate(q, 30)
    mk = (lambda: HipMultiIndex(D, 0, [0, 0], shard=True)) if multi else (lambda: HipFlatIndex(D, 0, 0))
    for first, second in ((0, P), (P, 0)):
        ix = mk()
        ix.set_label_denominator(first)
        ix.reserve(3000)
        ix.add(torch.from_numpy(bank).cuda()); ix.add_labels(torch.from_numpy(lab).cuda()); ix.set_num_classes(C)
        ix.use_current_stream()
        assert torch.equal(ix.search_aggregate(q, 30).view(torch.int32), ix.search_aggregate(q, 30).view(torch.int32))
        ix.reset()
        ix.set_label_denominator(second)
        assert ix.label_denominator == second
        ix.add(torch.from_numpy(bank[:2000]).cuda()); ix.add_labels(torch.from_numpy(lab[:2000]).cuda())
        got = ix.search_aggregate(q, 30)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (multi, first, second)
        got_rows = ix.gather_labels(torch.arange(2000, device="cuda"))
        assert np.array_equal(got_rows.cpu().numpy().view(np.uint32), lab[:2000].view(np.uint32))


@pytest.mark.parametrize("P", [0, 196])
def test_class_count_can_change_after_reset(cuda_device, P):
    """reset() keeps allocations and `lab_cap` counts ROWS: label rows of another width (more classes) after a reset must not be written
    into the buffer sized for the old width.  Also the padded count rows (16-byte granules) against a dense fp32 table, C = 151 -> 152."""
    D = 32
    bank = gi.unit_bank(4000, D, seed=3)
    q = torch.from_numpy(gi.vit_like_queries(300, D, seed=5)).cuda()
    ix = HipFlatIndex(D, 0, 0)
    ix.set_label_denominator(P)
    ix.reserve(4000)
    ix.add(torch.from_numpy(bank).cuda()); ix.add_labels(torch.from_numpy(gi.labels_from_masks(4000, 21, 196, seed=4)).cuda()); ix.set_num_classes(21)
    ix.search_aggregate(q, 30)
    ix.reset()
    lab = gi.labels_from_masks(4000, 151, 196, seed=6)
    ix.add(torch.from_numpy(bank).cuda()); ix.add_labels(torch.from_numpy(lab).cuda()); ix.set_num_classes(151)
    ref = HipFlatIndex(D, 0, 0); ref.add(torch.from_numpy(bank).cuda()); ref.add_labels(torch.from_numpy(lab).cuda()); ref.set_num_classes(151)
    assert torch.equal(ix.search_aggregate(q, 30).view(torch.int32), ref.search_aggregate(q, 30).view(torch.int32))
    assert np.array_equal(ix.gather_labels(np.arange(4000)).view(np.uint32), lab.view(np.uint32))
    if P:
        cnt = ix.copy_label_counts()
        assert tuple(cnt.shape) == (4000, 151)
        assert np.array_equal(cnt.cpu().numpy().view(np.uint16).astype(np.float32) / np.float32(P), lab)


def test_k4_k5_fused_vs_oracle(cuda_device):
    M, D, C, nq, k = 30000, 384, 21, 500, 30
    bank = gi.unit_bank(M, D, seed=1)
    lab = gi.labels_from_masks(M, C, 196, seed=2)
    q = gi.vit_like_queries(nq, D, seed=3)
    ix = HipFlatIndex(D, 0, 0)
    ix.add(bank); ix.add_labels(lab); ix.set_num_classes(C)
    out, idx, dist = ix.search_aggregate(torch.from_numpy(q).cuda(), k, want_neighbours=True)
    ridx, _ = oracle.knn_chain_f32(q, bank, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    kf, kl = oracle.gather_neighbours(ridx, bank, lab, 1, nq)
    ref = oracle.cross_attention(q[None], kf, kl)[0]
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-5
    out_host = ix.search_aggregate(q, k)                    # host-pointer path
    assert np.array_equal(out_host, out.cpu().numpy())
    # L2 index: the cosine logits are recovered from squared distances
    ix2 = HipFlatIndex(D, 1, 0)
    ix2.add(bank); ix2.add_labels(lab); ix2.set_num_classes(C)
    out2 = ix2.search_aggregate(torch.from_numpy(q).cuda(), k).cpu().numpy()
    r2, _ = oracle.knn_chain_f32(q, bank, k, "l2")
    kf2, kl2 = oracle.gather_neighbours(r2, bank, lab, 1, nq)
    assert np.abs(out2 - oracle.cross_attention(q[None], kf2, kl2)[0]).max() < 2e-4


@pytest.mark.parametrize("B,S,C,h,w", [(2, 4, 5, 32, 32), (3, 14, 21, 224, 224), (2, 37, 151, 518, 518), (1, 7, 19, 100, 60)])
def test_k6_upsample_argmax(cuda_device, B, S, C, h, w):
    rng = np.random.default_rng(S)
    lh = rng.random((B, S * S, C)).astype(np.float32)
    lh /= lh.sum(-1, keepdims=True)
    got = ops.upsample_argmax(torch.from_numpy(lh).cuda(), S, h, w).cpu().numpy()
    ref = oracle.upsample_argmax(lh, S, h, w)
    assert got.shape == (B, 1, h, w)
    assert (got == ref).mean() > 0.9999


def test_k6_golden_cluster_map(cuda_device, golden_dir):
    g = np.load(f"{golden_dir}/g67_memory_evaluate.npz")
    for name in ("unb", "bnd", "trim", "ade"):
        C, D, H, ps = g[f"cfg_{name}"][:4].tolist()
        lh = g[f"label_hat_{name}"]
        got = ops.upsample_argmax(torch.from_numpy(lh).cuda(), H // ps, H, H).cpu().numpy()
        assert (got == g[f"cluster_map_{name}"]).mean() > 0.9995


def test_k7_predsmiou_golden(cuda_device, golden_dir):
    g = np.load(f"{golden_dir}/g8_predsmiou.npz")
    for name in ("c5", "c21", "ade"):
        C, n, ign = g[f"cfg_{name}"].tolist()
        gt, pred = g[f"gt_{name}"], g[f"pred_{name}"]
        for mode, kw in {"hung": {}, "m2o": {"many_to_one": True},
                         "m2o_prec": {"many_to_one": True, "precision_based": True}, "lin": {"linear_probe": True}}.items():
            m = PredsmIoU(C, C, ignore_index=ign)
            m.update(torch.from_numpy(gt.reshape(2, -1)), torch.from_numpy(pred.reshape(2, -1)))   # CPU in, GPU kernel
            assert np.array_equal(m._conf_mat.cpu().numpy(), g[f"conf_{name}"])
            miou, tp, fp, fn, reordered, bg = m.compute(is_global_zero=True, **kw)
            assert abs(miou - float(g[f"miou_{name}_{mode}"])) < 1e-12
            assert tp == g[f"tp_{name}_{mode}"].tolist() and fp == g[f"fp_{name}_{mode}"].tolist()
            assert fn == g[f"fn_{name}_{mode}"].tolist()
            assert abs(bg - float(g[f"bg_{name}_{mode}"])) < 1e-12
            assert len(reordered) == int(g[f"nreordered_{name}_{mode}"])
            assert reordered[:64] == g[f"reordered_head_{name}_{mode}"].tolist()
        with pytest.raises(ValueError):
            m.update(torch.zeros(3), torch.zeros(4))
        assert m.compute(is_global_zero=False) == (0.0, [], [], [], [], 0.0)


@pytest.mark.parametrize("B,S,C,h,w,ign", [(2, 14, 21, 224, 224, 255), (3, 37, 151, 518, 518, 0), (1, 7, 5, 100, 61, None), (2, 16, 300, 130, 70, 255)])
def test_k6_k7_fused_equals_the_two_kernels(cuda_device, B, S, C, h, w, ign):
    """hb_upsample_argmax_confusion (row f1: upsample + argmax + confusion matrix in one kernel): the confusion counts of K6 followed
    by K7 -- piecewise-constant masks (wave-aggregated atomics: one per group of equal pairs), ignore values, out-of-range gt,
    ragged right borders, more classes than an LDS histogram holds -- and, on request, K6's class map bit for bit."""
    rng = np.random.default_rng(S * C + h)
    lh = torch.from_numpy(rng.random((B, S * S, C), dtype=np.float32)).cuda()
    gt = gi.random_masks(B, h, w, C, seed=h + w, with_255=True)
    gt[0, 0, :3, :5] = C + 3                                   # out of range: dropped
    gt = torch.from_numpy(gt).cuda()
    pred = ops.upsample_argmax(lh, S, h, w)
    conf2 = torch.zeros((C, C), dtype=torch.int64, device="cuda"); ops.confusion_update(conf2, gt, pred, ign)
    conf1 = torch.zeros((C, C), dtype=torch.int64, device="cuda")
    out = ops.upsample_argmax_confusion(lh, S, gt, conf1, ign, want_map=True)
    assert torch.equal(out, pred) and torch.equal(conf1, conf2) and int(conf1.sum()) > 0
    conf3 = torch.zeros((C, C), dtype=torch.int64, device="cuda")
    assert ops.upsample_argmax_confusion(lh, S, gt, conf3, ign) is None and torch.equal(conf3, conf2)
    ref = oracle.confusion_matrix(gt.cpu().numpy().reshape(-1), pred.cpu().numpy().reshape(-1), C, C, ign) if hasattr(oracle, "confusion_matrix") else None
    if ref is not None:
        assert np.array_equal(conf1.cpu().numpy(), ref)
    m1 = PredsmIoU(C, C, ignore_index=ign, store_reordered_preds=True); m2 = PredsmIoU(C, C, ignore_index=ign, store_reordered_preds=True)
    m1.update(gt, pred); m2.update_from_label_hat(gt, lh, S)
    assert m1.compute(True, many_to_one=True) == m2.compute(True, many_to_one=True)


def test_k7_large_class_count_global_path(cuda_device):
    rng = np.random.default_rng(0)
    C = 200                                        # 200*200*4 B > LDS budget -> global-atomic path
    gt = rng.integers(0, C, 200_000); pred = rng.integers(0, C, 200_000)
    conf = torch.zeros((C, C), dtype=torch.int64, device="cuda")
    ops.confusion_update(conf, torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda(), None)
    assert np.array_equal(conf.cpu().numpy(), oracle.confusion_matrix(gt, pred, C, C, None))


@pytest.mark.parametrize("S,C,win,H,W,stride", [(4, 5, 32, 48, 80, 24), (16, 21, 224, 224, 320, 96), (37, 19, 518, 600, 900, 300)])
def test_sliding_window_accumulate_and_argmax_bit_exact(cuda_device, S, C, win, H, W, stride):
    """hb_upsample_accumulate + hb_argmax_channels against the oracle's stitching: same fp32 sums, same class map."""
    rng = np.random.default_rng(S + C)
    B = 2
    origins = oracle.window_origins(H, W, win, stride)
    lhs = [rng.random((B, S * S, C), dtype=np.float32) for _ in origins]
    acc = torch.zeros((B, H, W, C), device="cuda")
    for lh, (y0, x0) in zip(lhs, origins):
        ops.upsample_accumulate(torch.from_numpy(lh).cuda(), S, acc, y0, x0, win, win)
    cm = ops.argmax_channels(acc)
    rcm, racc = oracle.sliding_window_argmax(lhs, origins, S, win, H, W)
    got = acc.permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), racc.view(np.uint32))
    assert np.array_equal(cm.cpu().numpy(), rcm)


def test_sliding_window_ops_errors(cuda_device):
    acc = torch.zeros((1, 16, 16, 3), device="cuda")
    lh = torch.rand((1, 16, 3), device="cuda")
    with pytest.raises(RuntimeError):
        ops.upsample_accumulate(lh, 4, acc, 8, 0, 16, 16)        # window sticks out of the frame
    with pytest.raises(ValueError):
        ops.upsample_accumulate(lh, 5, acc, 0, 0, 16, 16)        # 16 tokens are not 5 x 5
    with pytest.raises(ValueError):
        ops.upsample_accumulate(lh, 4, acc[..., :2], 0, 0, 16, 16)
    ties = torch.zeros((1, 2, 2, 4), device="cuda")
    assert int(ops.argmax_channels(ties).sum()) == 0             # first maximum wins


@pytest.mark.parametrize("D,metric", [(384, 0), (768, 1), (1024, 0), (400, 1), (1536, 0), (16, 1)])
def test_k1_lds_form_and_first_form_write_the_same_bits(cuda_device, D, metric):
    """K1 in its two forms (hb_set_layout_form): rows staged through LDS once (D a multiple of 16 up to 1152) against the first form, which
    D = 400 and D = 1536 keep anyway.  Same tiles (reconstructed rows), same stored norms, same L2 row constants (through the distances of
    a search) and the same query tiles -- bit for bit, with ragged appends behind a partly filled row tile, unnormalised rows, a NaN row
    (a zero token through the eps-free normalisation) and host buffers."""
    from hbird_mi import _lib
    rng = np.random.default_rng(D)
    pieces = [rng.standard_normal((n, D)).astype(np.float32) * s for n, s in ((1000, 1.0), (37, 3.0), (2048, 0.2), (5, 1.0), (331, 1.0))]
    pieces[2][7] = 0.0                                           # normalised: NaN row
    q = (3.0 * rng.standard_normal((300, D))).astype(np.float32)
    out = []
    try:
        for form in (1, 0, 8, 16, 32):                             # the first form, automatic, and every rows-per-workgroup choice of the LDS form
            _lib.check(_lib.lib().hb_set_layout_form(form))
            ix = HipFlatIndex(D, metric, 0)
            for j, pc in enumerate(pieces):
                ix.add(torch.from_numpy(pc).cuda() if j % 2 == 0 else pc, normalize=(j != 3))      # device and host paths; piece 3 as it is
            n = ix.ntotal
            rows = ix.reconstruct(torch.arange(n, device="cuda")).cpu().numpy()
            norms = ix.copy_norms().cpu().numpy()
            idx, dist = ix.search(torch.from_numpy(q).cuda(), 20)
            out.append((rows, norms, idx.cpu().numpy(), dist.cpu().numpy()))
    finally:
        _lib.check(_lib.lib().hb_set_layout_form(0))
    r1, n1, i1, d1 = out[0]
    for r0, n0, i0, d0 in out[1:]:
        assert np.array_equal(r1.view(np.uint32), r0.view(np.uint32))
        assert np.array_equal(n1.view(np.uint32), n0.view(np.uint32))
        assert np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint32), d0.view(np.uint32))
