"""Host-side pieces that need no GPU: folder datasets (on a generated miniature VOC tree), transforms, CLI parsing,
and the C-ABI surface (the library must load and export every symbol declared in include/hbird_hip.h)."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_voc(root, n=6):
    rng = np.random.default_rng(0)
    for d in ("images", "SegmentationClassAug", "SegmentationClass", "sets"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    names = [f"img{i:03d}" for i in range(n)]
    for nm in names:
        h, w = int(rng.integers(40, 80)), int(rng.integers(40, 80))
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "images", nm + ".jpg"))
        m = rng.integers(0, 21, (h, w), dtype=np.uint8); m[:2] = 255
        for d in ("SegmentationClassAug", "SegmentationClass"):
            Image.fromarray(m).save(os.path.join(root, d, nm + ".png"))
    open(os.path.join(root, "sets", "trainaug.txt"), "w").write("\n".join(names[:4]))
    open(os.path.join(root, "sets", "val.txt"), "w").write("\n".join(names[4:]))


def test_voc_folder_datamodule(tmp_path):
    from hbird_mi.data import get_dataset
    _make_voc(str(tmp_path))
    dm, ignore = get_dataset("voc", str(tmp_path), batch_size=2, num_workers=0, input_size=32)
    assert ignore == 255 and dm.get_num_classes() == 21 and dm.get_train_dataset_size() == 4
    for loader, n in ((dm.train_dataloader(), 4), (dm.val_dataloader(), 2)):
        seen = 0
        for x, y in loader:
            assert x.shape[1:] == (3, 32, 32) and y.shape[1:] == (1, 32, 32) and x.dtype == torch.float32
            cls = (y * 255).long()                      # hbird_eval.py:219 round trip
            assert ((cls >= 0) & ((cls <= 20) | (cls == 255))).all()
            seen += x.shape[0]
        assert seen == n
    dm2, _ = get_dataset("voc*0.5", str(tmp_path), 2, 0, 32)
    assert dm2.get_train_dataset_size() == 2
    with pytest.raises(ValueError):
        get_dataset("imagenet", str(tmp_path), 2, 0, 32)
    with pytest.raises(RuntimeError):
        get_dataset("voc", str(tmp_path / "nope"), 2, 0, 32)


def test_dataset_table_matches_reference_constants():
    from hbird_mi.data import DATASET_INFO
    assert DATASET_INFO == {"voc": (21, 255), "ade20k": (151, 0), "cityscapes": (19, 255), "coco-thing": (12, 255),
                            "coco-stuff": (15, 255)}


def test_synthetic_datamodule_is_deterministic():
    from hbird_mi.data import get_dataset
    a, _ = get_dataset("synthetic", "", 4, 0, 32)
    b, _ = get_dataset("synthetic", "", 4, 0, 32)
    xa, ya = next(iter(a.train_dataloader())); xb, yb = next(iter(b.train_dataloader()))
    assert torch.equal(xa, xb) and torch.equal(ya, yb) and xa.shape == (4, 3, 32, 32)
    assert torch.equal((ya * 255).long().unique(), (ya * 255).round().long().unique())


def test_cli_parsing():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hb_cli", os.path.join(ROOT, "eval.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    assert cli.parse_nn_params(["idx_shard=true", "gpu_ids=3", "beta=0.5", "distance_measure=l2"]) == \
        {"idx_shard": True, "gpu_ids": [3], "beta": 0.5, "distance_measure": "l2"}
    assert cli.parse_nn_params(["gpu_ids=0,1, 2", "label_shard=True"]) == {"gpu_ids": [0, 1, 2], "label_shard": True}
    assert cli.parse_nn_params(["gpu_ids=[4,5]"]) == {"gpu_ids": [4, 5]}
    a = cli.build_parser().parse_args(["--dataset-name", "voc*0.2", "--data-dir", "/x", "--d-model", "384",
                                       "--patch-size", "16", "--nn-method", "faiss", "--nn-param", "use_fp16=false"])
    assert a.n_neighbours == 30 and a.batch_size == 64 and a.input_size == 224 and a.seed == 123
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["--dataset-name", "voc", "--data-dir", "/x", "--d-model", "0", "--patch-size", "16"])


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    """No compute calls (no GPU here): the .so must load and export exactly what include/hbird_hip.h declares."""
    from hbird_mi import _lib
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "hbird_hip.h")).read()
    declared = set(re.findall(r"\b(hb_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in hbird_hip.h but not exported by libhbird_hip.so"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # ... and INTEGRATION.md's entry-point map (C ABI <-> reference) names every one of them
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert not [n for n in sorted(declared) if n not in doc], "entries missing from INTEGRATION.md's map"
    assert _lib.device_count() == 0 or _lib.device_count() > 0


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hbird_mi.nn.search_hip import NearestNeighborSearchHIP
    from hbird_mi.utils.eval_metrics import PredsmIoU
    with pytest.raises(RuntimeError):
        NearestNeighborSearchHIP(torch.zeros(4, 8))
    with pytest.raises(RuntimeError):
        PredsmIoU(3, 3)


def test_feature_extractor_backends_cpu():
    """FeatureExtractor auto-detection (counterpart of hbird/models.py:164-235): DINO-style
    get_intermediate_layers (CLS first), DINOv2-style forward_features dict, plain module output with CLS."""
    from hbird_mi.models import FeatureExtractor, FeatureExtractorSimple
    B, S, D = 2, 4, 8
    tok = torch.arange(B * (S * S + 1) * D, dtype=torch.float32).view(B, S * S + 1, D)

    class Dino(torch.nn.Module):
        def get_intermediate_layers(self, x):
            return [tok]

        def get_last_selfattention(self, x):
            return None

    class DinoV2(torch.nn.Module):
        def forward_features(self, x):
            return {"x_norm_clstoken": tok[:, 0], "x_norm_patchtokens": tok[:, 1:]}

    class Plain(torch.nn.Module):
        def forward(self, x):
            return tok

    class TimmLike(torch.nn.Module):
        """timm >= 0.9 VisionTransformer: has get_intermediate_layers too, but it skips the final norm (norm=False);
        the reference takes forward_features(imgs)[:, 1:] for this family (hbird/models.py:208-216, 344-345)."""

        def __init__(self):
            super().__init__()
            blk = torch.nn.Module(); blk.attn = torch.nn.Identity()
            self.blocks = torch.nn.ModuleList([blk])

        def get_intermediate_layers(self, x):
            return [tok]                              # un-normed tokens: must NOT be used

        def forward_features(self, x):
            return 2.0 * tok + 1.0                    # "after the final norm"

    class HfLike(torch.nn.Module):
        class config:
            model_type = "vit"

        def forward(self, x, output_attentions=False, return_dict=True):
            att = torch.rand(B, 3, S * S + 1, S * S + 1)
            return type("Out", (), {"last_hidden_state": tok, "attentions": (att,)})()

    x = torch.zeros(B, 3, 8, 8)
    Dino.get_last_selfattention = lambda self, x: torch.rand(B, 2, S * S + 1, S * S + 1)
    # a module with none of the family APIs and no blocks[i].attn.qkv: the reference's generic QKV-hook fallback raises RuntimeError
    # (hbird/models.py:280-285) -- so does this one, instead of guessing a token stream from the module's output
    fe = FeatureExtractor(Plain(), eval_spatial_resolution=S, d_model=D)
    assert fe.backend == "generic"
    with pytest.raises(RuntimeError, match="qkv module not found"):
        fe.forward_features(x)
    for m, backend, want in ((Dino(), "dino", tok[:, 1:]), (DinoV2(), "dinov2", tok[:, 1:]),
                             (TimmLike(), "timm", 2.0 * tok[:, 1:] + 1.0), (HfLike(), "hf", tok[:, 1:])):
        fe = FeatureExtractor(m, eval_spatial_resolution=S, d_model=D)
        assert fe.backend == backend
        out, attn = fe.forward_features(x)
        assert out.shape == (B, S * S, D) and torch.equal(out, want)
        if backend in ("dino", "hf"):
            assert attn.shape == (B, S * S) and float(attn.min()) == 0.0 and float(attn.max()) == 1.0
        else:
            assert attn is None
        assert fe.eval_spatial_resolution == S and fe.d_model == D
    fs = FeatureExtractorSimple(Plain(), lambda model, imgs: model(imgs)[:, 1:], eval_spatial_resolution=S, d_model=D)
    out, attn = fs.forward_features(x)
    assert attn is None and torch.equal(out, tok[:, 1:])
    fs2 = FeatureExtractorSimple(Plain(), lambda model, imgs: (model(imgs)[:, 1:], "a"), S, D)
    assert fs2(x)[1] == "a"


def test_feature_extractor_generic_qkv_hook_golden_g9(golden_dir):
    """The generic fallback (hook on blocks[i].attn.qkv, hbird/models.py:257-321) against the REFERENCE's FeatureExtractor on a tiny
    random ViT (tests/golden/g9_feature_extractor.npz, written by gen_golden.py): Q / K / V of both blocks bit for bit; the default
    layer (-1: the reference's dotted lookup never resolves it and raises, recorded in the fixture) = the last block here; a plain
    nn.Linear qkv ([B, N, 3 D]) gives the same features; freeze_feature_extractor (237-255) with substrings and regexes."""
    from tiny_vit import TinyQKVViT
    from hbird_mi.models import FeatureExtractor
    g = np.load(os.path.join(golden_dir, "g9_feature_extractor.npz"))
    assert str(g["backend"]) == "generic" and str(g["default_layer_raises"]) == "RuntimeError"
    imgs = torch.from_numpy(g["imgs"])
    vit = TinyQKVViT(seed=5).eval()
    fe = FeatureExtractor(vit, eval_spatial_resolution=4, d_model=16)
    assert fe.backend == "generic" and fe.device == torch.device("cpu")
    for feat in "qkv":
        for layer in (0, 1):
            f, att = fe.get_intermediate_layer_feats(imgs, feat=feat, layer_num=layer)
            assert att is None and np.array_equal(f.numpy(), g[f"feats_{feat}_{layer}"]), (feat, layer)
        f, att = fe.forward_features(imgs, feat=feat)                     # layer -1 = the last block
        assert att is None and f.dtype == torch.float32 and np.array_equal(f.numpy(), g[f"feats_{feat}_1"])
        f2, _ = FeatureExtractor(TinyQKVViT(seed=5, flat=True).eval(), 4, 16).forward_features(imgs, feat=feat)
        assert np.array_equal(f2.numpy(), g[f"feats_{feat}_1"])
    assert np.array_equal(fe(imgs)[0].numpy(), g["feats_k_1"])          # "k" is the default (models.py:164)
    with pytest.raises(RuntimeError, match="qkv module not found"):
        fe.get_intermediate_layer_feats(imgs, layer_num=7)
    fe.freeze_feature_extractor(["blocks.1.attn"])
    assert sorted(n for n, p in vit.named_parameters() if p.requires_grad) == list(g["trainable_after_freeze"])
    fe.freeze_feature_extractor([r"blocks\.0\..*bias$"], regex=True)
    assert sorted(n for n, p in vit.named_parameters() if p.requires_grad) == list(g["trainable_after_regex_freeze"])
    fe.freeze_feature_extractor()
    assert not any(p.requires_grad for p in vit.parameters())


def test_coco_stuff_and_cityscapes_folder(tmp_path):
    """Category -> coarse-class tables (coco_data.py:104-124) and the Cityscapes labelId -> trainId map (28-48)."""
    import json
    from hbird_mi.data import get_dataset
    rng = np.random.default_rng(1)
    root = tmp_path / "coco"
    for sp in ("train", "val"):
        os.makedirs(root / "images" / f"{sp}2017"); os.makedirs(root / "annotations" / "stuff_annotations" / f"stuff_{sp}2017_pixelmaps")
        for i in range(3):
            Image.fromarray(rng.integers(0, 255, (40, 50, 3), dtype=np.uint8)).save(root / "images" / f"{sp}2017" / f"{i:04d}.jpg")
            m = rng.choice(np.array([0, 92, 93, 100, 183], dtype=np.uint8), size=(40, 50))
            Image.fromarray(m).save(root / "annotations" / "stuff_annotations" / f"stuff_{sp}2017_pixelmaps" / f"{i:04d}.png")
    cats = [{"id": 92, "supercategory": "textile"}, {"id": 93, "supercategory": "building"},
            {"id": 100, "supercategory": "textile"}, {"id": 183, "supercategory": "other"}]
    json.dump({"categories": cats}, open(root / "annotations" / "stuff_annotations" / "stuff_val2017.json", "w"))
    dm, ign = get_dataset("coco-stuff", str(root), 3, 0, 32)
    assert ign == 255 and dm.get_num_classes() == 15
    x, y = next(iter(dm.val_dataloader()))
    vals = set((y * 255).long().unique().tolist())
    assert vals <= {0, 1, 255} and {0, 1} <= vals          # building -> 0, textile -> 1, things/other -> 255
    # cityscapes
    croot = tmp_path / "city"
    for sp in ("train", "val"):
        os.makedirs(croot / "leftImg8bit" / sp / "ulm"); os.makedirs(croot / "gtFine" / sp / "ulm")
        Image.fromarray(rng.integers(0, 255, (32, 64, 3), dtype=np.uint8)).save(croot / "leftImg8bit" / sp / "ulm" / "ulm_000_leftImg8bit.png")
        m = rng.choice(np.array([0, 7, 8, 26, 33], dtype=np.uint8), size=(32, 64))
        Image.fromarray(m).save(croot / "gtFine" / sp / "ulm" / "ulm_000_gtFine_labelIds.png")
    dm, ign = get_dataset("cityscapes", str(croot), 1, 0, 32)
    x, y = next(iter(dm.val_dataloader()))
    assert set((y * 255).long().unique().tolist()) <= {0, 1, 13, 18, 255}    # 7->0 road, 8->1, 26->13 car, 33->18 bicycle


def test_voc_from_a_tar_archive_equals_the_folder(tmp_path):
    """The dataset root may be `/x/voc.tar` or `/x/all.tar!/inner` (hbird/utils/io.py:10-15, voc_tar_data.py): same
    samples as the unpacked tree, also through DataLoader workers; frames may be rectangular (sliding windows)."""
    import tarfile
    from hbird_mi.data import get_dataset
    from hbird_mi.data.folder import TarStore, read_file_set
    tree = tmp_path / "VOC"
    _make_voc(str(tree))
    plain, nested = tmp_path / "voc.tar", tmp_path / "all.tar.gz"
    with tarfile.open(plain, "w") as t:
        for d in sorted(os.listdir(tree)):
            t.add(tree / d, arcname=d)
    with tarfile.open(nested, "w:gz") as t:
        t.add(tree, arcname="data/VOC")
    ref, _ = get_dataset("voc", str(tree), 2, 0, (32, 48))
    want = [(x.clone(), y.clone()) for x, y in ref.val_dataloader()]
    for root, workers in ((str(plain), 0), (str(nested) + "!/data/VOC/", 0), (str(plain), 2)):
        dm, ign = get_dataset("voc", root, 2, workers, (32, 48))
        assert ign == 255 and dm.get_train_dataset_size() == 4
        got = list(dm.val_dataloader())
        assert len(got) == len(want)
        for (x, y), (rx, ry) in zip(got, want):
            assert x.shape[-2:] == (32, 48) and torch.equal(x, rx) and torch.equal(y, ry)
    st = TarStore(str(nested) + "!/data/VOC")
    assert st.isdir("images") and not st.isdir("nope") and st.listdir("sets") == ["trainaug.txt", "val.txt"]
    assert st.listdir("") == ["SegmentationClass", "SegmentationClassAug", "images", "sets"]
    assert read_file_set(str(nested) + "!/data/VOC/sets/val.txt") == ["img004", "img005"]
    assert read_file_set(str(tree / "sets" / "val.txt")) == ["img004", "img005"]
    with pytest.raises(FileNotFoundError):
        TarStore(str(tmp_path / "missing.tar"))
    with pytest.raises(FileNotFoundError):
        st.open("images/none.jpg")
    with pytest.raises(RuntimeError):
        get_dataset("voc", str(nested) + "!/data/other", 2, 0, 32)


def test_c_abi_rejects_a_null_handle_without_a_gpu():
    """Entry points validate the handle before touching the device (error code + hb_last_error, never a crash)."""
    from hbird_mi import _lib
    L = _lib.lib()
    assert L.hb_index_add(None, None, 10, 0, 0) != 0
    assert b"NULL index handle" in L.hb_last_error()
    assert L.hb_index_search(None, None, 1, 1, 0, None, None, 0) != 0
    assert L.hb_index_set_fp16(None, 1) != 0
    # EVERY entry that takes an index handle: a NULL handle is an error (or -1 rows), never a dereference
    import ctypes
    ms, n64, info = ctypes.c_double(0), ctypes.c_int64(0), (ctypes.c_int64 * 8)()
    calls = {
        "hb_index_set_stream": (None, None), "hb_index_reserve": (None, 10), "hb_index_add_labels": (None, None, 1, 3, 0),
        "hb_index_reset": (None,), "hb_index_search_aggregate": (None, None, 1, 1, 0, 0.02, None, None, None, 0),
        "hb_index_aggregate": (None, None, 1, None, None, 1, 0, 0.02, None, 1),
        "hb_index_aggregate_partial": (None, None, 1, None, None, 1, 0, 0.02, None, 0, None),
        "hb_index_reconstruct": (None, None, 1, 0, None, 0), "hb_index_gather_labels": (None, None, 1, 0, None, 0),
        "hb_index_set_label_table": (None, None, None, 0, 0, 0), "hb_index_copy_norms": (None, None, 0),
        "hb_index_set_score_output": (None, 1), "hb_index_distances_from_scores": (None, None, 1, 1, None),
        "hb_index_set_timing": (None, 1), "hb_index_last_knn_ms": (None, ctypes.byref(ms)),
        "hb_index_set_tuning": (None, 0, 0), "hb_index_last_fp16_fallbacks": (None, ctypes.byref(n64)), "hb_index_set_fp16_escalation": (None, 0), "hb_index_last_fp16_escalated": (None, ctypes.byref(n64)),
        "hb_index_set_variant": (None, 0), "hb_index_set_search_options": (None, 1, 0), "hb_index_set_rerank_copy": (None, 0), "hb_index_kernel_clock": (None, (ctypes.c_double * 4)()), "hb_index_xcd_stats": (None, 0, (ctypes.c_double * 12)()), "hb_index_wg_stamps": (None, info, 8, ctypes.byref(ctypes.c_int(0))), "hb_index_set_xcd_weights": (None, 1, None), "hb_index_xcd_weights": (None, 0, (ctypes.c_double * 8)(), None), "hb_index_rerank_copy_bytes": (None, ctypes.byref(n64)), "hb_index_schedule_info": (None, info), "hb_index_set_cluster": (None, 2, 2, 16), "hb_index_set_cluster_sharing": (None, 2), "hb_index_set_label_denominator": (None, 196), "hb_index_labels_to_fp32": (None,),
        "hb_index_label_denominator": (None, ctypes.byref(ctypes.c_int(0))), "hb_index_copy_label_counts": (None, None, 1),
        "hb_index_set_label_count_table": (None, None, None, 0, 0, 0, 0),
        "hb_index_cluster_stats": (None, info),
    }
    for name, args in calls.items():
        assert getattr(L, name)(*args) != 0, name
        assert b"NULL" in L.hb_last_error(), (name, L.hb_last_error())
    assert L.hb_index_ntotal(None) == -1 and L.hb_index_nlabels(None) == -1
    assert L.hb_index_free(None) == 0                       # free(NULL) is a no-op, like free()
    taking_ix = {n for n, (_, a) in _lib.SIGNATURES.items() if n.startswith("hb_index_") and n != "hb_index_create"}
    assert taking_ix == set(calls) | {"hb_index_add", "hb_index_search", "hb_index_set_fp16", "hb_index_ntotal",
                                      "hb_index_nlabels", "hb_index_free"}, "an hb_index_* entry is not covered above"
    # the multi-GPU handle: NULL is an error (or -1 rows), a bad GPU list fails before any device is touched
    h = ctypes.c_void_p()
    assert L.hb_multi_create(8, 0, None, 0, 1, ctypes.byref(h)) != 0 and b"at least one GPU" in L.hb_last_error()
    for name, args in {"hb_multi_reserve": (None, 10), "hb_multi_add": (None, None, 1, 0), "hb_multi_set_fp16": (None, 1),
                       "hb_multi_search": (None, None, 1, 1, None, None), "hb_multi_shard_rows": (None, None, 0)}.items():
        assert getattr(L, name)(*args) != 0 and b"NULL" in L.hb_last_error(), name
    assert L.hb_multi_ntotal(None) == -1 and L.hb_multi_free(None) == 0
    # the packed-list helpers validate their arguments too
    assert L.hb_packed_list_bytes(21904, 30) == (21904 * 30 * 12 + 15) // 16 * 16
    assert L.hb_merge_topk_packed(None, 64, 2, 4, 1, 0, None, None, None) != 0
    assert L.hb_merge_topk(None, None, 2, 4, 1, 0, None, None, None) != 0


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_gpus_n_starts_n_ranks_before_touching_a_gpu(monkeypatch):
    """`python bench.py --gpus 2` without a torchrun environment must start 2 ranks itself (child process tree running
    torch.distributed.run), not print a 1-GPU line; under torchrun a --gpus / WORLD_SIZE mismatch is an error."""
    bench = _load_bench()
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 7

    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("HBIRD_BENCH_ONE_GPU", "1")          # no GPU count check (this container has none)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched before the launch")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                  # the children's status is the parent's
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # more ranks than GPUs is refused (normal mode)
    monkeypatch.delenv("HBIRD_BENCH_ONE_GPU")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "GPU(s) visible" in str(e.value.code)
    # under a launcher: WORLD_SIZE must agree with --gpus
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=4" in str(e.value.code)


def test_bench_kernel_source_hash_tracks_the_hip_sources():
    bench = _load_bench()
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    t = json.load(open(os.path.join(ROOT, "profiles", "latest_knn_traffic.json")))
    assert "traffic_bytes_per_launch" in t and "workload" in t
