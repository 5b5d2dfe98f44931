"""Edge cases of the hot path that the reference leaves undefined or untested, pinned here:

  * NaN bank rows -- a zero token through the reference's eps-free normalisation (hbird_eval.py:324) -- and queries with
    NaN / +-inf components: a NaN (or -inf) score never enters a neighbour list, in any kernel path (cold start, scan,
    dump-and-walk, pools, fp16 candidates, clusters); the oracle states the rule (hbird_oracle.c: topk_push_f32);
  * fp32 values outside the fp16 range under use_fp16 (search_faiss.py:40): the candidate pass cannot certify them, the
    fp32 kernel answers, and the call reports it;
  * FeatureExtractor (auto-detect + fp16 autocast, the reference's default API path: hbird_eval.py:674-681,
    models.py:188-192) on the GPU, feeding K1;
  * the full 1024 x 2048 Cityscapes frame (BASELINE cfg-5) through the sliding-window stitcher.
"""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import oracle
from hbird_mi import ops
from hbird_mi.nn.search_hip import HipFlatIndex

pytestmark = pytest.mark.gpu


def _check_exact(idx, dist, q, bank, k, metric):
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric)
    idx = idx.cpu().numpy() if isinstance(idx, torch.Tensor) else idx
    dist = dist.cpu().numpy() if isinstance(dist, torch.Tensor) else dist
    bad = np.argwhere(idx != ridx)
    assert bad.size == 0, f"{len(bad)} index mismatches, first {bad[:5].tolist()}: got {idx[tuple(bad[0])]} want {ridx[tuple(bad[0])]}"
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    return ridx, rdist


def _nan_bank(M, D, seed):
    """Unit rows with NaN rows planted where they hurt: most of the FIRST tile of the bank (what a slot's cold start
    selects its threshold from), a whole 256-row tile, and strays."""
    bank = gi.unit_bank(M, D, seed=seed)
    nan_rows = np.r_[3:200, 1024:1280, [M // 2, M - 1]]
    nan_rows = nan_rows[nan_rows < M]
    bank[nan_rows] = np.nan
    return bank, nan_rows


# (M, D, nq, k, metric, fp16, workgroups, cluster): one case per kernel path
_PATHS = [
    (20_000, 64, 700, 30, "dot_product", False, 0, None),    # small search: phased pools, B-direct <WIDE, COLD> (bisection cold start + scan)
    (20_000, 64, 700, 30, "dot_product", -6, 0, None),       # small search on the sorted LDS lists (variant 6): B-direct COLD instantiation
    (20_000, 64, 700, 5, "dot_product", False, 0, None),     # ... which k < 8 takes by itself
    (20_000, 48, 700, 30, "dot_product", False, 0, None),    # small search: LDS-staged kernel (6 stages per tile)
    (20_000, 64, 700, 30, "l2", False, 0, None),             # L2: NaN row init (-0.5 |b|^2)
    (20_000, 64, 300, 90, "dot_product", False, 0, None),    # candidate pools (k > 32)
    (20_000, 64, 700, 30, "dot_product", True, 0, None),     # fp16 candidate kernel (second design) + re-rank
    (20_000, 64, 300, 100, "dot_product", True, 0, None),    # fp16 candidate kernel, pools beyond 256 entries (its <8> instantiation)
    (70_001, 64, 1300, 30, "dot_product", False, 64, (2, 4, 4)),   # clustered schedule
    (1_600_000, 64, 2048, 30, "dot_product", False, 1, None),      # 400 k stages per workgroup: the plain instantiation
    (150, 64, 70, 30, "dot_product", True, 0, None),         # fewer finite rows than candidates: nothing to certify
]


@pytest.mark.parametrize("M,D,nq,k,metric,fp16,G,cluster", _PATHS)
def test_nan_bank_rows_never_enter_a_list(cuda_device, M, D, nq, k, metric, fp16, G, cluster):
    bank, nan_rows = _nan_bank(M, D, seed=M + D + k)
    q = gi.vit_like_queries(nq, D, seed=9)
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    if fp16 == -6:
        ix.set_variant(6); fp16 = False
    ix.set_fp16(bool(fp16))
    if G:
        ix.set_tuning(G, 0)
    if cluster:
        ix.set_cluster(*cluster)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    ridx, _ = _check_exact(idx, dist, q, bank, k, metric)
    assert not np.isin(ridx, nan_rows).any() and ((ridx >= 0).all() or M < 1000)
    if fp16 and M > 1000:
        assert ix.last_fp16_fallbacks() == 0             # NaN rows are excluded by both precisions alike: still certified


def test_all_nan_bank_and_short_lists(cuda_device):
    """A bank of NaN rows only has no neighbours at all; 10 finite rows among NaNs give 10 neighbours and k - 10 gaps
    (id -1, distance -inf / +inf), as for a bank with fewer than k rows."""
    D, k = 64, 30
    bank = np.full((3000, D), np.nan, dtype=np.float32)
    q = gi.vit_like_queries(300, D, seed=1)
    for metric in ("dot_product", "l2"):
        ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
        ix.add(torch.from_numpy(bank).cuda())
        idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
        assert (idx == -1).all() and torch.isinf(dist).all()
        some = bank.copy()
        some[100:2100:200] = gi.unit_bank(10, D, seed=2)
        ix2 = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
        ix2.add(torch.from_numpy(some).cuda())
        for fp16 in (False, True):
            ix2.set_fp16(fp16)
            idx, dist = ix2.search(torch.from_numpy(q).cuda(), k)
            ridx, _ = _check_exact(idx, dist, q, some, k, metric)
            assert ((ridx >= 0).sum(1) == 10).all()


@pytest.mark.parametrize("fp16", [False, True])
def test_non_finite_queries(cuda_device, fp16):
    """Queries with NaN / +-inf components: their scores are NaN (or +-inf where every product keeps one sign) and follow
    the same rule; the other queries of the batch are untouched.  K5 gives a query without neighbours a zero row."""
    M, D, nq, k, C = 30_000, 64, 600, 30, 7
    bank = gi.unit_bank(M, D, seed=4)
    bank[:, 5] = np.abs(bank[:, 5]) + 1e-3                   # component 5 is positive in every row
    q = gi.vit_like_queries(nq, D, seed=5)
    q[3, 17] = np.nan
    q[40, 5] = np.inf                                        # +inf * positive: every score is +inf (ties -> lowest ids)
    q[41, 5] = -np.inf                                       # every score -inf: nothing listed
    q[300, 0] = np.inf                                       # mixed signs: +inf, -inf rows
    q[301, :] = np.nan
    q[599, 1] = np.inf; q[599, 2] = -np.inf                  # inf - inf = NaN (almost) everywhere
    lab = gi.labels_from_masks(M, C, 196, seed=6)
    ix = HipFlatIndex(D, 0, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.add_labels(torch.from_numpy(lab).cuda())
    ix.set_num_classes(C)
    ix.set_fp16(fp16)
    qd = torch.from_numpy(q).cuda()
    idx, dist = ix.search(qd, k)
    ridx, rdist = _check_exact(idx, dist, q, bank, k, "dot_product")
    assert (ridx[[3, 41, 301]] == -1).all() and (ridx[40] == np.arange(k)).all() and np.isposinf(rdist[40]).all()
    if fp16:
        assert ix.last_fp16_fallbacks() >= 5                 # the non-finite queries cannot be certified
    clean = np.ones(nq, bool); clean[[3, 40, 41, 300, 301, 599]] = False
    ix.set_fp16(False)
    i2, d2 = ix.search(torch.from_numpy(q[clean]).cuda(), k)
    assert torch.equal(i2, idx[torch.from_numpy(clean).cuda()]) and torch.equal(d2, dist[torch.from_numpy(clean).cuda()])
    lh = ix.search_aggregate(qd, k)
    assert torch.equal(lh[[3, 41, 301]], torch.zeros((3, C), device="cuda"))
    kf, kl = oracle.gather_neighbours(ridx[clean], bank, lab, 1, int(clean.sum()))
    want = oracle.cross_attention(q[clean][None], kf, kl)[0]
    assert np.abs(lh.cpu().numpy()[clean] - want).max() < 2e-5


@pytest.mark.parametrize("k,fp16,variant", [(30, False, 0), (30, False, 6), (90, False, 0), (30, True, 0), (5, False, 0)])
@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_zero_and_denormal_scale_queries(cuda_device, k, fp16, variant, metric):
    """A zero query scores exactly 0 against every row, a 1e-42-scale query has denormal scores: every row ties (or nearly), the
    threshold searches of the cold start / the phase floors (bisection between a tile's smallest and largest score) must not lean on how
    the hardware treats denormals.  Ties resolve to the lowest ids, as everywhere."""
    M, D, nq = 40_000, 64, 300
    bank = gi.unit_bank(M, D, seed=21)
    q = gi.vit_like_queries(nq, D, seed=22)
    q[0] = 0.0
    q[1] = 1e-42                      # denormal components
    q[2] = -1e-42
    q[150] = 0.0
    q[299] = 1e-30 * q[298]
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(fp16)
    ix.set_variant(variant)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    ridx, _ = _check_exact(idx, dist, q, bank, k, metric)
    if metric == "dot_product":
        assert (ridx[0] == np.arange(k)).all() and (ridx[150] == np.arange(k)).all()


@pytest.mark.parametrize("metric", ["dot_product", "l2"])
def test_fp16_overflow_falls_back_to_fp32(cuda_device, metric):
    """use_fp16 with fp32 values beyond the fp16 range (|x| > 65,504 -> inf in fp16): a query that overflows is reported
    as a fallback and answered by the fp32 kernel; a bank that overflows stays on the fp32 kernel altogether.  Same bits
    as the oracle either way."""
    M, D, nq, k = 40_000, 64, 500, 30
    bank = gi.unit_bank(M, D, seed=11)
    q = gi.vit_like_queries(nq, D, seed=12)
    q[7, 3] = 1.0e5; q[8, :] *= 4.0e4; q[100, 63] = -7.0e4
    ix = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(True)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, bank, k, metric)
    assert 3 <= ix.last_fp16_fallbacks() <= 12, ix.last_fp16_fallbacks()
    idx, dist = ix.search(torch.from_numpy(q[200:]).cuda(), k)           # no overflow: certified, nothing falls back
    assert ix.last_fp16_fallbacks() == 0
    big = bank.copy()
    big[123] *= 3.0e5; big[39_999, 0] = 1.0e5                            # un-normalised rows (the plugin adds what it is given)
    ix2 = HipFlatIndex(D, 0 if metric == "dot_product" else 1, 0)
    ix2.add(torch.from_numpy(big).cuda())
    ix2.set_fp16(True)
    idx, dist = ix2.search(torch.from_numpy(q).cuda(), k)
    _check_exact(idx, dist, q, big, k, metric)
    assert ix2.last_fp16_fallbacks() == nq
    ix2.reset()
    ix2.add(torch.from_numpy(bank).cuda())                               # the flag belongs to the rows, not to the handle
    idx, dist = ix2.search(torch.from_numpy(q[200:]).cuda(), k)
    assert ix2.last_fp16_fallbacks() == 0
    _check_exact(idx, dist, q[200:], bank, k, metric)


def test_zero_token_through_the_bank_build(cuda_device):
    """hbird_eval.py:324 divides by the norm with no eps: a zero token becomes a NaN bank row.  K1 does the same, the row
    keeps its place (ids unchanged) and is never returned."""
    D, k = 32, 10
    feats = torch.from_numpy(gi.vit_like_queries(600, D, seed=3)).cuda()
    feats[17] = 0.0; feats[599] = 0.0
    ix = HipFlatIndex(D, 0, 0)
    ix.add(feats, normalize=True)
    rec = ix.reconstruct(torch.arange(600, device="cuda"))
    assert torch.isnan(rec[17]).all() and torch.isnan(rec[599]).all() and not torch.isnan(rec[18]).any()
    want = oracle.normalize_rows(feats.cpu().numpy())
    assert np.isnan(want[17]).all()
    ok = ~np.isnan(want).any(1)
    assert np.abs(rec.cpu().numpy()[ok] - want[ok]).max() <= 2.5e-7
    q = feats[:50].clone()
    idx, dist = ix.search(q, k)
    bank = rec.cpu().numpy()
    _check_exact(idx, dist, q.cpu().numpy(), bank, k, "dot_product")
    assert not np.isin(idx.cpu().numpy(), [17, 599]).any()


# ---- FeatureExtractor on the GPU ---------------------------------------------------------------------------------------
class _Attn(torch.nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.h, self.qkv, self.proj = heads, torch.nn.Linear(D, 3 * D), torch.nn.Linear(D, D)
        self.last = None

    def forward(self, x):
        B, N, D = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.h, D // self.h).permute(2, 0, 3, 1, 4)
        a = (q @ k.transpose(-1, -2) * (D // self.h) ** -0.5).softmax(-1)
        self.last = a
        return self.proj((a @ v).transpose(1, 2).reshape(B, N, D))


class _Block(torch.nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.n1, self.attn, self.n2 = torch.nn.LayerNorm(D), _Attn(D, heads), torch.nn.LayerNorm(D)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(D, 4 * D), torch.nn.GELU(), torch.nn.Linear(4 * D, D))

    def forward(self, x):
        x = x + self.attn(self.n1(x))
        return x + self.mlp(self.n2(x))


class _TinyViT(torch.nn.Module):
    def __init__(self, D=64, ps=8, heads=4):
        super().__init__()
        self.embed = torch.nn.Conv2d(3, D, ps, ps)
        self.cls = torch.nn.Parameter(torch.randn(1, 1, D) * 0.02)
        self.blocks = torch.nn.ModuleList([_Block(D, heads), _Block(D, heads)])
        self.norm = torch.nn.LayerNorm(D)

    def _run(self, x):
        t = self.embed(x).flatten(2).transpose(1, 2)
        t = torch.cat([self.cls.expand(t.shape[0], -1, -1).to(t.dtype), t], 1)
        for b in self.blocks:
            t = b(t)
        return self.norm(t)


class _Dino(_TinyViT):                                     # DINO: get_intermediate_layers + get_last_selfattention
    def get_intermediate_layers(self, x):
        return [self._run(x)]

    def get_last_selfattention(self, x):
        self._run(x)
        return self.blocks[-1].attn.last


class _DinoV2Tiny(_TinyViT):                               # DINOv2: class name + forward_features dict
    def forward_features(self, x):
        t = self._run(x)
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:]}


class _Timm(_TinyViT):                                     # timm: forward_features + blocks[0].attn
    def forward_features(self, x):
        return self._run(x)


@pytest.mark.parametrize("cls,backend", [(_Dino, "dino"), (_DinoV2Tiny, "dinov2"), (_Timm, "timm")])
def test_feature_extractor_autocast_on_the_gpu(cuda_device, cls, backend):
    """The reference's default API path (ftr_extr_fn=None -> FeatureExtractor: fp16 autocast + inference_mode,
    hbird/models.py:188-192) on ROCm: fp32 tokens out, within fp16 tolerance of the fp32 run, and the bank the evaluator
    builds from them is bit for bit what K1 makes of those very tokens."""
    from hbird_mi.hbird_eval import HbirdEvaluation
    from hbird_mi.models import FeatureExtractor
    torch.manual_seed(0)
    D, ps, H, C, B = 64, 8, 64, 5, 4
    S = H // ps
    model = cls(D, ps).cuda().eval()
    fe = FeatureExtractor(model, eval_spatial_resolution=S, d_model=D)
    assert fe.backend == backend and fe.use_autocast
    x = torch.randn(B, 3, H, H, device="cuda")
    tok, attn = fe.forward_features(x)
    assert tok.dtype == torch.float32 and tok.shape == (B, S * S, D) and tok.is_cuda
    fe32 = FeatureExtractor(model, eval_spatial_resolution=S, d_model=D, use_autocast=False)
    tok32, _ = fe32.forward_features(x)
    err = (tok - tok32).abs().max().item()
    assert 0.0 < err < 3e-2 * tok32.abs().max().item(), err          # fp16 GEMMs ran, and only fp16 rounding separates the two
    if backend == "dino":
        assert attn.shape == (B, S * S) and float(attn.min()) == 0.0
    train = [(torch.randn(B, 3, H, H), torch.from_numpy(gi.random_masks(B, H, H, C, seed=i)).float() / 255) for i in range(3)]
    ev = HbirdEvaluation(fe, train, num_classes=C, n_neighbours=10, device="cuda", nn_method="faiss")
    ix = HipFlatIndex(D, 0, 0)
    for xb, _ in train:
        t, _ = fe.forward_features(xb.cuda())
        ix.add(t.reshape(-1, D), normalize=True)                                        # K1 on the same tokens
    ids = torch.arange(ix.ntotal, device="cuda")
    assert ev.index.ntotal == ix.ntotal == 3 * B * S * S
    assert torch.equal(ev.index.reconstruct(ids), ix.reconstruct(ids))
    val = [(torch.randn(B, 3, H, H), torch.from_numpy(gi.random_masks(B, H, H, C, seed=9)).float() / 255)]
    jac = ev.evaluate(val, S)
    assert isinstance(jac, float) and 0.0 <= jac <= 1.0


# ---- BASELINE cfg-5: the full Cityscapes frame -----------------------------------------------------------------------------
def test_full_cityscapes_frame_stitch(cuda_device):
    """1024 x 2048 frame, C = 19, 518-pixel windows of 37 x 37 tokens (stride 510: 2 x 4 windows, the last row / column
    flush with the border, overlaps everywhere): the accumulator (a 159 MB fp32 [1024, 2048, 19] frame) and the class map
    are bit-equal to the oracle's stitching."""
    from hbird_mi import tiling
    H, W, C, win, S, stride = 1024, 2048, 19, 518, 37, 510
    rng = np.random.default_rng(5)
    origins = tiling.window_origins(H, W, win, stride)
    assert origins == oracle.window_origins(H, W, win, stride) and len(origins) == 8
    lhs = [rng.random((1, S * S, C), dtype=np.float32) for _ in origins]
    acc = torch.zeros((1, H, W, C), dtype=torch.float32, device="cuda")
    for (y0, x0), lh in zip(origins, lhs):
        ops.upsample_accumulate(torch.from_numpy(lh).cuda(), S, acc, y0, x0, win, win)
    pred = ops.argmax_channels(acc)
    want_pred, want_acc = oracle.sliding_window_argmax(lhs, origins, S, win, H, W)            # acc as [B, C, H, W]
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), np.ascontiguousarray(want_acc.transpose(0, 2, 3, 1)).view(np.uint32))
    assert np.array_equal(pred.cpu().numpy(), want_pred)


@pytest.mark.parametrize("k,fp16", [(30, False), (30, True), (90, False)])
def test_astronomically_large_scores_do_not_break_the_cold_start(cuda_device, k, fp16):
    """one +inf / 1e38-sized score per query stretches the cold start's bisection interval beyond what twelve linear
    halvings can cross; such queries bisect on the monotone keys instead.  Results are the oracle's either way (speed only)."""
    M, D, nq = 30_000, 64, 300
    bank = gi.unit_bank(M, D, seed=41)
    bank[777] *= 1e30                                          # an unnormalised row of huge norm: scores of ~1e31 .. inf
    bank[20_001] *= 3e37
    q = gi.vit_like_queries(nq, D, seed=42)
    q[5, 3] = np.inf                                           # and a query with an inf component: +inf / -inf / NaN scores
    ix = HipFlatIndex(D, 0, 0)
    ix.add(torch.from_numpy(bank).cuda())
    ix.set_fp16(fp16)
    idx, dist = ix.search(torch.from_numpy(q).cuda(), k)
    ridx, rdist = oracle.knn_chain_f32(q, bank, k)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy().view(np.uint32), rdist.view(np.uint32))
