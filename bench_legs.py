#!/usr/bin/env python3
"""The legs of bench.py beside the timed steps, one function each: A/B legs on the resident bank (equal XCD shares, no clusters, use_fp16, end to
end), the live counter passes, the CPU baseline, the mIoU replay, the N-rank selftest and the pre-registered scaling model.  bench.py calls
every leg through `safe()`: a leg that fails reports {"failed": ...} and never costs the bench line.

Only `cpu_baseline` and `selftest_against_oracle` touch oracle/ -- as the thing timed beside the GPU number and as the checker."""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")

PEAK_FP32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_FP16_MFMA_TFLOPS = 2516.6   # 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz (dense, no sparsity)
NOMINAL_GHZ = 2.4


def safe(name, fn, *args, **kw):
    """One try per leg."""
    try:
        return fn(*args, **kw)
    except SystemExit:
        raise
    except Exception as e:          # an extra leg never costs the bench line
        return {"failed": f"{name}: {e!r}"}


def spread(xs):
    """min / median / max of a list of per-step values."""
    xs = sorted(float(x) for x in xs)
    return {"min": xs[0], "median": xs[len(xs) // 2], "max": xs[-1]} if xs else {"min": None, "median": None, "max": None}


def host_cpu_budget():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU boxes of this pool show 256 hardware
    threads but grant a container 16 CPUs of quota (cpu.max "1600000 100000"): 128 OpenMP threads on that are 8-fold oversubscribed
    -- round 3's "0.50 TFLOP/s on 128 cores"."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota, src = None, "none"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            quota, src = float(q) / float(per), f"cgroup v2 cpu.max {q} {per}"
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota, src = q / per, f"cgroup v1 cfs_quota_us {q} / {per}"
        except Exception:
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"cores": cores, "hardware_threads": os.cpu_count(), "affinity": aff, "cgroup_quota_cpus": quota, "quota_source": src}


def scann_cpu_leg(bank, q, k, exact_idx, threads):
    """The reference's default CPU backend, restated call by call (hbird/nn/search_scann.py:18-33 builder chain with its default
    parameters, :40 search_batched), timed on the same sample -- when `scann` can be imported at all."""
    try:
        import scann
    except Exception as e:                                    # absent from this image (no network to install it)
        return f"unavailable ({type(e).__name__}: {e})"
    t0 = time.time()
    b = scann.scann_ops_pybind.builder(bank, k, "dot_product")
    b = b.tree(num_leaves=512, num_leaves_to_search=32, training_sample_size=bank.shape[0])
    b = b.score_ah(2, anisotropic_quantization_threshold=0.2, dimensions_per_block=4)
    index = b.reorder(120).build()
    t_build = time.time() - t0
    t0 = time.time()
    nb, _ = index.search_batched(q)
    dt = time.time() - t0
    recall = float(np.mean([len(set(a.tolist()) & set(b_.tolist())) / float(k) for a, b_ in zip(np.asarray(nb), exact_idx)]))
    return {"build_seconds": round(t_build, 2), "value_on_sample": q.shape[0] / dt, "unit": "query-patches/s", "recall_at_k": recall,
            "bank_rows": int(bank.shape[0]), "queries": int(q.shape[0]), "threads": threads,
            "parameters": "num_leaves 512, num_leaves_to_search 32, AH(2, 0.2, dimensions_per_block 4), reorder 120 (search_scann.py defaults)"}


def host_mem_budget():
    """Bytes of host memory this process may still take: min(MemAvailable, cgroup limit - cgroup usage)."""
    avail, src = None, "none"
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail, src = int(ln.split()[1]) * 1024, "MemAvailable"
    except Exception:
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            room = int(mx) - int(open("/sys/fs/cgroup/memory.current").read())
            if avail is None or room < avail:
                avail, src = room, "cgroup v2 memory.max - memory.current"
    except Exception:
        pass
    return {"available_bytes": avail, "source": src}


def fetch_bank_to_host(index, M, D, need_free_factor=1.3):
    """The bench's own bank as plain fp32 rows on the host (hb_index_reconstruct in chunks), for the full-size CPU baseline -- only when the
    host has the memory for it beside this process (an out-of-memory kill would take the box down): -> (array or None, note)."""
    mem = host_mem_budget()
    need = int(M) * int(D) * 4
    if mem["available_bytes"] is None or mem["available_bytes"] < need * need_free_factor + (4 << 30):
        return None, {"host_memory": mem, "bank_bytes": need, "fetched": False}
    t0 = time.time()
    out = np.empty((M, D), dtype=np.float32)
    step = 250_000
    dev = torch.device("cuda", index.device)
    for r in range(0, M, step):
        ids = torch.arange(r, min(M, r + step), device=dev)
        out[r:r + ids.numel()] = index.reconstruct(ids).cpu().numpy()
    return out, {"host_memory": mem, "bank_bytes": need, "fetched": True, "seconds": round(time.time() - t0, 1)}


def cpu_full_bank_leg(bank, q, k, threads):
    """The chain-oracle port against the WHOLE bank (no extrapolation in the rows) on 256 of the step's queries: 16 register blocks of 16
    queries, one per granted core, each streaming all rows once."""
    import oracle
    nqs = min(int(q.shape[0]), 16 * max(1, threads))
    oracle.knn_chain_f32(q[:16], bank[:100_000], k)           # warm up
    t0 = time.time()
    idx, _ = oracle.knn_chain_f32(q[:nqs], bank, k)
    dt = time.time() - t0
    return {"value": nqs / dt, "unit": "query-patches/s", "bank_rows": int(bank.shape[0]), "queries": nqs, "seconds": round(dt, 2), "threads": threads,
            "first_query_neighbours": idx[0, :4].tolist()}


def cpu_baseline(D, k, M_total, bank_full=None, q_step=None, fetch_note=None):
    """The oracle's exact fp32 brute force (oracle/hbird_oracle.c, OpenMP + AVX2) on the cores the host grants (host_cpu_budget).
    With the whole bank on the host (bank_full: the bench's own rows, fetched when memory allows): measured at FULL bank size on 256 of the
    step's queries -- `value` is that rate, nothing extrapolated.  Otherwise (and beside it) the bounded sample of round 5: 6,144 queries x
    400,000 rows scaled linearly in the rows, with a half-size sample that shows the scaling; the host BLAS on the same product; ScaNN
    when it is installed.  Reported baseline only."""
    import oracle
    import torch as _t
    budget = host_cpu_budget()
    threads = budget["cores"]
    oracle.set_num_threads(threads)
    _t.set_num_threads(threads)
    full = None
    if bank_full is not None and q_step is not None:
        full = safe("cpu_full_bank", cpu_full_bank_leg, bank_full, q_step, k, threads)
    rng = np.random.default_rng(0)
    ms, nqs = 400_000, 6144            # ~15 s of CPU work on the 16 cores this pool grants (oracle 6 + 3 s, BLAS legs ~5 s)
    bank = rng.standard_normal((ms, D), dtype=np.float32)
    bank /= np.linalg.norm(bank, axis=1, keepdims=True)
    q = 3.0 * rng.standard_normal((nqs, D), dtype=np.float32)
    oracle.knn_chain_f32(q[:64], bank[:10000], k)            # warm up threads
    t0 = time.time()
    ex_idx, _ = oracle.knn_chain_f32(q, bank, k)
    dt = time.time() - t0
    qps_sample = nqs / dt
    t0 = time.time()
    oracle.knn_chain_f32(q, bank[:ms // 2], k)               # linearity: half the rows
    dt_half = time.time() - t0
    # second CPU reference point: torch mm + topk, same sample
    qb, bb = _t.from_numpy(q), _t.from_numpy(bank)
    (qb[:256] @ bb.T).topk(k, dim=1)
    t1 = time.time()
    nqt = 1024
    for i in range(0, nqt, 256):
        (qb[i:i + 256] @ bb.T).topk(k, dim=1)
    dt_t = time.time() - t1
    # the contraction alone (no k-select): what the host's BLAS sustains on this shape with the granted cores -- the CPU's own
    # ceiling for the dominant term, so that the un-tuned port above can be read against it
    (qb[:256] @ bb.T)
    t3 = time.time()
    for i in range(0, nqs, 1024):
        (qb[i:i + 1024] @ bb.T)
    dt_m = time.time() - t3
    mm_tflops = 2.0 * nqs * ms * D / dt_m / 1e12
    blas = [ln.strip() for ln in _t.__config__.parallel_info().splitlines() if "Math Kernel" in ln or "get_num_threads" in ln or "OpenBLAS" in ln]
    # the reference-equivalent CPU stage after the search (hbird_eval.py:631-637, 575-609, 235-243), one 37 x 37 image
    S, C = 37, 151
    idx1 = rng.integers(0, ms, size=(S * S, k))
    lab = rng.random((ms, C), dtype=np.float32)
    t2 = time.time()
    kf, kl = oracle.gather_neighbours(idx1, bank, lab, 1, S * S)
    lh = oracle.cross_attention(q[:1].repeat(S * S, 0)[None], kf, kl)
    oracle.upsample_argmax(lh, S, 14 * S, 14 * S)
    dt_p = time.time() - t2
    scann_leg = scann_cpu_leg(bank, q, k, ex_idx, threads)
    res = {
        "value": qps_sample * ms / M_total,
        "unit": "query-patches/s",
        "cores": threads,
        "kind": "port",
        "tuned": False,                             # the chain oracle is a parity tool (one fmaf chain per score), not a tuned SGEMM
        "extrapolated": True,                       # value = measured sample rate x (sample rows / bank rows)
        "sample": f"oracle exact fp32 brute force on {nqs} queries x {ms} rows x {D} dims took {dt:.2f}s on {threads} threads "
                  f"({qps_sample:.1f} q/s), scaled x{ms}/{M_total} to the full bank",
        "scann": scann_leg if isinstance(scann_leg, str) else "measured (see scann_leg)",
        "host": budget,
        "measured_on_sample": {"value": qps_sample, "unit": "query-patches/s", "bank_rows": ms, "queries": nqs, "seconds": round(dt, 2)},
        "linearity_check": {"rows": [ms // 2, ms], "seconds": [round(dt_half, 2), round(dt, 2)],
                            "seconds_ratio": dt / dt_half, "expected": 2.0,
                            "what": "same queries against half the sample and the whole sample: brute force is linear in the bank rows, which is what the extrapolation uses"},
        "scann_leg": scann_leg,
        "torch_mm_topk": {"value": nqt / dt_t * ms / M_total, "unit": "query-patches/s", "threads": _t.get_num_threads(),
                          "sample_seconds": round(dt_t, 2)},
        "torch_mm_only": {"tflops": mm_tflops, "gflops_per_core": mm_tflops * 1e3 / threads, "value": nqs / dt_m * ms / M_total,
                          "unit": "query-patches/s (no k-select)", "threads": _t.get_num_threads(), "sample_seconds": round(dt_m, 2),
                          "blas": blas,
                          "what": f"fp32 [{nqs},{D}] x [{D},{ms}] products only: the host BLAS ceiling for the contraction on the granted cores"},
        "post_knn_stage": {"value": S * S / dt_p, "unit": "query-patches/s",
                           "what": "gather + cross-attention + bilinear upsample + argmax of one 37x37-token image, C=151"},
        "full_bank_fetch": fetch_note,
    }
    if isinstance(full, dict) and "value" in full:
        # the number without extrapolation takes the headline slot of the object; the sample's extrapolation stays beside it
        res.update({"value": full["value"], "extrapolated": False, "value_extrapolated_from_sample": qps_sample * ms / M_total,
                    "sample": f"oracle exact fp32 brute force on {full['queries']} of the step's queries x ALL {full['bank_rows']} rows x {D} dims took "
                              f"{full['seconds']:.2f}s on {threads} threads (the bench's own bank, fetched to the host; no extrapolation)",
                    "full_bank": full})
    elif full is not None:
        res["full_bank"] = full
    return res


def miou_parity(device):
    """BASELINE.json's second metric, 'mIoU delta vs ref': replay the fixtures that tests/golden/gen_golden.py produced
    with the reference's own HbirdEvaluation (bank build + evaluation from recorded tokens) through this engine."""
    path = os.path.join(ROOT, "tests", "golden", "g67_memory_evaluate.npz")
    if not os.path.exists(path):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ReplayExtractor, golden_case
    from hbird_mi.hbird_eval import HbirdEvaluation
    g = np.load(path)
    out = {}
    for name in ("unb", "bnd", "ade"):
        c = golden_case(g, name)
        torch.set_rng_state(torch.from_numpy(g[f"rng_state_{name}"]))      # the reference run started from this state
        ext = ReplayExtractor(c["tr_tok"] + c["va_tok"], c["S"], c["D"])
        ev = HbirdEvaluation(ext, c["train"], num_classes=c["C"], n_neighbours=c["k"], augmentation_epoch=c["aug"],
                             device=str(device), nn_method="hip", memory_size=c["mem"], dataset_size=c["nb"] * c["B"])
        jac = ev.evaluate(c["val"], c["S"], ignore_index=c["ign"])
        out[name] = abs(float(jac) - float(g[f"jac_{name}"]))
    return {"max_abs_miou_delta_vs_reference": max(out.values()), "cases": out,
            "fixture": "tests/golden/g67_memory_evaluate.npz (reference HbirdEvaluation outputs)", "tolerance": 1e-4}


def pmc_pass(a, kernel, counters, fp16):
    """One `rocprofv3 --kernel-trace --pmc <counters>` child pass of this script in --pmc-child mode (same bank, ONE untimed search; the
    program itself follows `--`).  -> ({counter: sum over the kernel family's dispatches, "ms": their total duration, "launches": n,
    "kernel": the name of the dispatch that ran longest}, None) or (None, why)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    import csv
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    args = ["--rows", str(a.rows), "--dim", str(a.dim), "--classes", str(a.classes), "--nq", str(a.nq), "--k", str(a.k),
            "--workgroups", str(a.workgroups), "--panel", str(a.panel), "--variant", str(a.variant),
            ] + (["--fp16"] if fp16 else ["--cluster-shape", str(getattr(a, "cluster_q", 0)), str(getattr(a, "cluster_b", 0))])
    out = tempfile.mkdtemp(prefix="hbird_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "--",
               sys.executable, BENCH, "--pmc-child"] + args
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        vals, dur, names = {}, {}, {}
        for root, _, files in os.walk(out):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    for row in csv.DictReader(open(os.path.join(root, f))):
                        if kernel in row["Kernel_Name"] and row["Counter_Name"] in counters:
                            vals[row["Counter_Name"]] = vals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                            names[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not dur:
            return None, f"rocprofv3 --pmc {' '.join(counters)}: no {kernel} rows (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
        vals["ms"] = sum(dur.values()); vals["launches"] = len(dur); vals["kernel"] = names[max(dur, key=dur.get)]
        return vals, None
    except Exception as e:     # optional evidence, never a reason to lose the bench line
        return None, f"rocprofv3 --pmc {' '.join(counters)} failed: {e!r}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def matrix_pipe_counters(a, kernel, fp16):
    """clock_ghz and mfma_busy of the kNN kernel family from one counter pass (MI355X_MICROARCH.md, rocprofv3 section): GRBM_GUI_ACTIVE
    counts busy cycles per XCD (8 of them) -> clock = GRBM_GUI_ACTIVE / 8 / kernel time; SQ_VALU_MFMA_BUSY_CYCLES sums the cycles each
    of the 1024 SIMDs had its matrix pipe busy -> mfma_busy = that / 1024 / (GRBM_GUI_ACTIVE / 8).  frac of the nominal peak =
    mfma_busy x clock / 2.4 GHz: the decomposition of a power-limited kernel's roofline fraction."""
    v, why = pmc_pass(a, kernel, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], fp16)
    if v is None:
        return {"clock_ghz": None, "mfma_busy": None, "source": why}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    return {"clock_ghz": cyc / (v["ms"] * 1e-3) / 1e9, "mfma_busy": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc,
            "kernel_ms_under_counters": v["ms"], "launches": v["launches"],
            "source": f"live: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on {v['kernel']} (one search); frac of the nominal peak = mfma_busy x clock_ghz / 2.4"}


def measure_traffic(a, kernel):
    """roofline.traffic measured LIVE: two `rocprofv3 --pmc` child passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one
    pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") of this script in --pmc-child mode (same bank, one untimed step;
    the program itself follows `--`).  FETCH_SIZE is doubled (gfx950 tallies 128-B requests of wide streaming reads at
    64 B).  Returns (bytes per launch, note) or (None, why)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    import csv
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    args = ["--rows", str(a.rows), "--dim", str(a.dim), "--classes", str(a.classes), "--nq", str(a.nq), "--k", str(a.k),
            "--workgroups", str(a.workgroups), "--panel", str(a.panel), "--variant", str(a.variant),
            ] + (["--fp16"] if a.fp16 else ["--cluster-shape", str(getattr(a, "cluster_q", 0)), str(getattr(a, "cluster_b", 0))])
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="hbird_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--",
                   sys.executable, BENCH, "--pmc-child"] + args
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            per_dispatch, names = {}, {}
            for root, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        for row in csv.DictReader(open(os.path.join(root, f))):
                            if kernel in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                                per_dispatch[row["Dispatch_Id"]] = per_dispatch.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
                                names[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if not per_dispatch:
                return None, f"rocprofv3 --pmc {ctr}: no {kernel} rows (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
            # the child runs ONE search: a pool search (use_fp16, k > 32, small banks) launches its kernel once per phase, so the
            # search's traffic is the sum over the family's dispatches (the LDS-list searches are one launch); the name is the
            # dispatch's that moved the most
            top = max(per_dispatch, key=per_dispatch.get)
            vals[ctr], vals["kernel"], vals["launches"] = sum(per_dispatch.values()), names[top], len(per_dispatch)     # KiB per search
        except Exception as e:     # the measurement is optional evidence, never a reason to lose the bench line
            return None, f"rocprofv3 --pmc {ctr} failed: {e!r}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024, \
        f"live: rocprofv3 --pmc on {vals['kernel']} ({vals['launches']} launch(es) of one search): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE = {vals['FETCH_SIZE']:.0f} KiB x2 + {vals['WRITE_SIZE']:.0f} KiB"


class DinoV2LikeViT(torch.nn.Module):
    """A plain pre-norm ViT with the DINOv2 interface (`forward_features(x)["x_norm_patchtokens"]`; the class name makes
    hbird_mi.models.FeatureExtractor pick its dinov2 path, models.py:199-206) and random weights: there is no network for checkpoints, and
    the end-to-end leg measures throughput, which does not depend on them."""

    def __init__(self, img, patch, dim, depth, heads):
        super().__init__()
        nn = torch.nn
        self.patch_embed = nn.Conv2d(3, dim, patch, patch)
        n = (img // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(0.02 * torch.randn(1, n + 1, dim))
        self.blocks = nn.ModuleList([nn.TransformerEncoderLayer(dim, heads, 4 * dim, dropout=0.0, activation="gelu", batch_first=True,
                                                                norm_first=True) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)

    def forward_features(self, x):
        t = self.patch_embed(x).flatten(2).transpose(1, 2)
        t = torch.cat([self.cls_token.expand(t.shape[0], -1, -1), t], dim=1) + self.pos_embed
        for b in self.blocks:
            t = b(t)
        t = self.norm(t)
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:]}


def e2e_leg(index, D, C, k, nq, device, n_batches):
    """BASELINE.json's configs are whole evaluations: images -> ViT -> kNN -> label aggregation -> upsample + argmax -> confusion matrix.
    This leg times HbirdEvaluation.evaluate (hbird_eval.py:184-265 of the reference) on the bench's own bank with a random-weight ViT of
    the config's architecture and synthetic images / masks from pinned host memory: images/s and where a batch's time goes."""
    from hbird_mi.hbird_eval import HbirdEvaluation
    from hbird_mi.models import FeatureExtractor
    arch = {384: ("ViT-S/16", 224, 16, 12, 6), 768: ("ViT-B/14", 518, 14, 12, 12), 1024: ("ViT-L/14", 518, 14, 24, 16),
            1536: ("ViT-g/14", 518, 14, 40, 24)}.get(D)
    if arch is None:
        return {"skipped": f"no ViT of width {D} in the reference's model list"}
    name, img, patch, depth, heads = arch
    S = img // patch
    if nq % (S * S) != 0:
        return {"skipped": f"{nq} queries per step are not whole {S} x {S}-token images"}
    B = nq // (S * S)
    torch.manual_seed(0)
    vit = DinoV2LikeViT(img, patch, D, depth, heads).to(device).eval()
    ext = FeatureExtractor(vit, eval_spatial_resolution=S, d_model=D)            # fp16 autocast + inference_mode: the reference's API default
    ev = HbirdEvaluation.from_index(ext, index, C, n_neighbours=k, device=str(device))
    g = torch.Generator().manual_seed(11)

    def loader(n):
        out = []
        for _ in range(n):
            x = torch.randn((B, 3, img, img), generator=g).pin_memory()
            y = (torch.randint(0, C, (B, 1, img, img), generator=g).float() / 255.0).pin_memory()      # masks as the reference's ToTensor delivers them
            out.append((x, y))
        return out
    res = {"model": f"{name} (random init), {img} px, batch {B}, FeatureExtractor (fp16 autocast)", "queries_per_batch": nq}
    for mode, fp16, n in (("fp32", False, n_batches), ("use_fp16", 2, 2 * n_batches)):
        index.set_fp16(fp16)
        ev.profile = False
        ev.evaluate(loader(1), S, ignore_index=255)                               # warm-up (kernels, fp16 copies of the bank, allocator)
        val = loader(n)
        ev.profile = True
        torch.cuda.synchronize(device)
        t0 = time.time()
        jac = ev.evaluate(val, S, ignore_index=255)
        torch.cuda.synchronize(device)
        dt = time.time() - t0
        st = ev.stage_times() or {}
        stages = {key: round(st[key], 3) for key in ("h2d_ms", "vit_forward_ms", "knn_k5_ms", "k6_k7_ms") if key in st}
        stages["loader_wait_ms"] = round(1e3 * st.get("loader_wait_s_total", 0.0) / max(1, n), 3)
        on_stream = {key: v for key, v in stages.items() if key in ("vit_forward_ms", "knn_k5_ms", "k6_k7_ms")}
        gpu_ms = sum(on_stream.values())
        res[mode] = {"images_per_s": B * n / dt, "ms_per_batch": dt / n * 1e3, "batches": n, "per_batch_ms": stages,
                     # the wall clock of a few batches also carries the one-off tail of evaluate() (confusion matrix to the host, Hungarian
                     # matching); over a real validation set the rate tends to the batches' own GPU time
                     "images_per_s_steady_state": B / (gpu_ms * 1e-3) if gpu_ms > 0 else None,
                     "bound_by": max(on_stream, key=on_stream.get) if on_stream else None,
                     "h2d_and_loader": "overlapped: the next batch is fetched and copied on a side stream during the current search",
                     "miou_of_random_weights": float(jac),
                     # use_fp16: what the LAST batch's search did about its certificates (hb_index_last_fp16_escalated / _fallbacks)
                     "last_batch_first_certificate_failed": index.last_fp16_escalated() if fp16 else None,
                     "last_batch_reached_fp32": index.last_fp16_fallbacks() if fp16 else None}
    index.set_fp16(False)
    del ev, ext, vit
    torch.cuda.empty_cache()
    return res


def selftest_against_oracle(index, q, k, lo, hi, world, rank, device, search_merged):
    """Before the timed steps of an N-rank run: the merged neighbour lists of 64 queries (what the ranks' kernels + the packed all-gather + the
    in-place merge produce) against the CPU chain oracle -- every rank searches ITS shard's rows with oracle.knn_chain_f32 (the checker, not
    the thing measured), the per-rank lists are gathered and merged on the host by (score descending, id ascending).  Ids AND score bits must
    agree on every rank; a mismatch ends the run with a message and a non-zero status (nothing is re-executed)."""
    td = torch.distributed
    nsel = min(64, q.shape[0])
    sel = torch.linspace(0, q.shape[0] - 1, nsel, device=device).long()
    mi, md = search_merged()                                    # [nq, k] merged ids / ordering scores, identical on every rank
    got_i, got_d = mi[sel].cpu().numpy(), md[sel].cpu().numpy()
    # the local, fallible part first (the oracle needs its C library; a rank without it must not leave the others waiting in a collective):
    # every rank reports whether its checker ran, and all of them go on or none
    err, ci, cd = None, None, None
    try:
        import oracle
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import chain_oracle_topk_chunked
        oracle.set_num_threads(max(1, host_cpu_budget()["cores"] // max(1, world)))
        n_local = hi - lo
        if n_local > 0:
            ci, cd = chain_oracle_topk_chunked(index, q[sel], n_local, min(k, n_local))
            ci = ci + lo
            if ci.shape[1] < k:
                pad = k - ci.shape[1]
                ci = np.concatenate([ci, np.full((nsel, pad), -1, dtype=np.int64)], axis=1)
                cd = np.concatenate([cd, np.full((nsel, pad), -np.inf, dtype=np.float32)], axis=1)
        else:
            ci = np.full((nsel, k), -1, dtype=np.int64); cd = np.full((nsel, k), -np.inf, dtype=np.float32)
    except Exception as e:
        err = repr(e)
    failed = torch.tensor([0 if err is None else 1], device=device)
    if world > 1:
        td.all_reduce(failed)
    if int(failed.item()) != 0:
        return {"unavailable": err or "the checker could not run on another rank"}
    parts_i = [torch.empty((nsel, k), dtype=torch.int64, device=device) for _ in range(world)]
    parts_d = [torch.empty((nsel, k), dtype=torch.float32, device=device) for _ in range(world)]
    if world > 1:
        td.all_gather(parts_i, torch.from_numpy(ci).to(device)); td.all_gather(parts_d, torch.from_numpy(cd).to(device))
    else:
        parts_i, parts_d = [torch.from_numpy(ci)], [torch.from_numpy(cd)]
    ai = np.concatenate([p.cpu().numpy() for p in parts_i], axis=1); ad = np.concatenate([p.cpu().numpy() for p in parts_d], axis=1)
    order = np.lexsort((np.where(ai < 0, np.iinfo(np.int64).max, ai), -ad.astype(np.float64)), axis=1)[:, :k]
    ref_i, ref_d = np.take_along_axis(ai, order, axis=1), np.take_along_axis(ad, order, axis=1)
    ok = bool(np.array_equal(got_i, ref_i) and np.array_equal(got_d.view(np.uint32), ref_d.view(np.uint32)))
    flag = torch.tensor([0 if ok else 1], device=device)
    if world > 1:
        td.all_reduce(flag)
    if int(flag.item()) != 0:
        bad = np.argwhere(got_i != ref_i)
        sys.stderr.write(f"bench.py selftest FAILED on rank {rank}/{world}: merged neighbour lists differ from the chain oracle "
                         f"({len(bad)} id mismatches on this rank, first {bad[:3].tolist()})\n")
        sys.stderr.flush()
        raise SystemExit(4)
    return {"queries": int(nsel), "k": int(k), "ids_and_score_bits_equal_the_chain_oracle": True,
            "how": "every rank: oracle.knn_chain_f32 over its own shard rows (chunked reconstruction); all-gather; host merge by (score desc, id asc)"}




# ---- A/B legs on the resident bank (N = 1) -----------------------------------------------------------------------------------------
def timed_searches(index, q, k, n, device, warm=1):
    """`warm` untimed + n timed search_aggregate steps with kernel timing on: (per-step kernel ms, per-step un-profiled clock GHz, wall ms / step)."""
    for _ in range(warm):
        index.search_aggregate(q, k, beta=0.02); torch.cuda.synchronize(device)
    index.set_timing(True)
    kms, ghz = [], []
    torch.cuda.synchronize(device)
    t0 = time.time()
    for _ in range(n):
        index.search_aggregate(q, k, beta=0.02)
        kms.append(index.last_knn_ms())
        ghz.append(index.kernel_clock()["ghz"])
    torch.cuda.synchronize(device)
    wall = (time.time() - t0) / max(1, n) * 1e3
    index.set_timing(False)
    return kms, ghz, wall


def equal_shares_leg(index, q, k, device, fp16=False, n=3):
    """A/B of the per-XCD work shares on THIS box: n steps with equal shares, then n with the calibrated ones (mode 0 starts from the
    shares this device is known to need and keeps calibrating under its guard).  Kernel ms by HIP events, same bank, same queries."""
    cal_w, cal_rounds = index.xcd_weights(fp16)
    index.set_xcd_weights(1)
    ke, ge, _ = timed_searches(index, q, k, n, device)
    index.set_xcd_weights(0)
    kc, gc, _ = timed_searches(index, q, k, n, device, warm=2)
    st = index.xcd_stats(fp16)
    e, c = float(np.mean(ke)), float(np.mean(kc))
    return {"equal_shares_kernel_ms": e, "calibrated_shares_kernel_ms": c, "calibrated_over_equal": c / e,
            "equal_shares_clock_ghz": float(np.median(ge)), "calibrated_shares_clock_ghz": float(np.median(gc)),
            "shares_before_the_leg": [round(v, 4) for v in cal_w], "rounds_before_the_leg": cal_rounds,
            "shares_after_the_leg": [round(v, 4) for v in index.xcd_weights(fp16)[0]], "calibration": st,
            "steps_each": n, "verdict": "calibrated shares faster" if c < e else "equal shares not slower on this box (the guard drops shares that measure slower)"}


def clusters_ab_leg(index, q, k, device, flops, peak, n=3):
    """The biggest fp32 searches can run in L2-sharing clusters (2 x 4 workgroups of one XCD on a common clock: -60 % L2-miss traffic for some
    cycles).  The index keeps them only where they MEASURE faster on this box (two calibrated launches with, two without, during the first
    five searches: hb_index_xcd_stats [10]); this leg times the OTHER form, so that both numbers and the decision are on the line."""
    cur = tuple(index.schedule_info().get("cluster", (1, 1)))
    st = index.xcd_stats(False)
    other = (1, 1, 0) if cur != (1, 1) else (2, 4, -1)
    index.set_cluster(*other)
    try:
        kms, ghz, _ = timed_searches(index, q, k, n, device)
        other_shape = tuple(index.schedule_info().get("cluster", (1, 1)))
    finally:
        index.set_cluster(0, 0, -1)
    m = float(np.mean(kms))
    res = {"cluster_in_timed_steps": list(cur), "other_form": list(other_shape), "other_form_kernel_ms": m, "other_form_frac": flops / (m * 1e-3) / 1e12 / peak,
           "other_form_clock_ghz_unprofiled": float(np.median(ghz)),
           "decision": {1: "clusters kept (measured faster on this box)", 0: "clusters dropped (measured slower on this box)", -1: "still measuring"}.get(st.get("clusters_kept", -1)),
           "clusters_kept": st.get("clusters_kept", -1), "clustered_minus_unclustered_ms_at_decision": st.get("clustered_minus_unclustered_ms"),
           "note": "same outputs either way; roofline.traffic is of the form the timed steps ran"}
    res["clustered_kernel_ms"] = m if cur == (1, 1) else None            # (the timed steps' own number fills the other slot: bench.py)
    res["unclustered_kernel_ms"] = m if cur != (1, 1) else None
    return res


def use_fp16_leg(index, q, k, device, flops, nq, n=3, warm=8):
    """The same step in use_fp16 mode (fp16 candidate pass + certified exact fp32 re-rank: the identical bits, DESIGN.md `use_fp16`)."""
    index.set_fp16(2)          # what the plugin's use_fp16=True sets: the candidate pass where it pays, adaptive on banks it cannot certify
    try:
        kms, ghz, wall = timed_searches(index, q, k, n, device, warm=warm)    # (finished warm-up searches calibrate the fp16 kernel's own shares)
        fb = index.last_fp16_fallbacks()
        m = float(np.mean(kms))
        return {"value": nq / (wall * 1e-3), "unit": "query-patches/s", "ms_per_step": wall, "fallback_queries": fb,
                "first_certificate_failed": index.last_fp16_escalated(),
                "candidate_kernel_ms": m, "candidate_kernel_ms_spread": spread(kms),
                "candidate_kernel_frac_of_fp16_mfma_peak": flops / (m * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                "clock_ghz_unprofiled": float(np.median(ghz)),
                "xcd_shares": [round(v, 4) for v in index.xcd_weights(True)[0]], "calibration": index.xcd_stats(True),
                "note": "certified-exact fast mode, same outputs as the fp32 search; synthetic N(0,1) rows (gap rank 30 -> 64 about 8 E).  Clustered "
                        "banks (token worlds, profiles/r06/final/fp16_cliff_10Mx768.json, same shape): 2.3 % first-pass failures 76.0 k q-p/s, 51 % "
                        "51.8 k, 99.9 % 71.4 k (one k' = 256 pass), banks fp16 cannot certify at all 9.7 k = the fp32 search"}
    finally:
        index.set_fp16(False)


# ---- the pre-registered scaling model ------------------------------------------------------------------------
# Measured on one MI355X each (profiles/r05/final/bench/bench_shard_*.json, bench_default.json: whole steps = kNN kernel + merge + K5, ms)
# at 2.377-2.383 GHz in-kernel clock: what ONE rank of an N-rank run searches.
SHARD_STEP_MS = {1: 2256.8, 2: 1130.4, 4: 566.4, 8: 284.7}
SHARD_CLOCK_GHZ = 2.38
XGMI_LINK_GBS = 60.0            # sustained per-direction rate assumed for one xGMI link under RCCL's ring (153 GB/s raw per link)
RING_STEP_MS = 0.02             # per ring step: launch + handshake
RING_BASE_MS = 0.05
MERGE_MS_AT_8 = 0.30            # merge_parts_kernel, 8 x 30 candidates per query, 21,904 queries (one-GPU dry run, profiles/r05/final/bench)
K5_MS_AT_1 = 0.15               # label aggregation of all 21,904 queries (a rank aggregates 1 / N of them)


def scaling_model(n_ranks, nq=21904, k=30, rows=10_000_000, dim=768, clock_ghz=None):
    """Predicted whole-job query-patches/s of `bench.py --gpus N` on the headline workload, written down BEFORE any multi-GPU run
    (this pool has one GPU per box): step = the rank's kNN share (measured per shard size, scaled by clock: this kernel's speed IS its
    clock, and eight GPUs share a node's power budget) + the exposed exchange = ring all-gather of the packed lists over xGMI + k-way merge
    + this rank's label aggregation."""
    n = int(n_ranks)
    if n in SHARD_STEP_MS and (rows, dim) == (10_000_000, 768):
        knn = SHARD_STEP_MS[n] * (nq / 21904.0)
    else:       # other shapes: the headline's rate per flop at the shard's size class
        knn = SHARD_STEP_MS[1] * (nq / 21904.0) * (rows / 1e7) * (dim / 768.0) / n
    if clock_ghz:
        knn *= SHARD_CLOCK_GHZ / float(clock_ghz)
    part_bytes = nq * k * 12.0
    gather = 0.0 if n == 1 else RING_BASE_MS + (n - 1) * (RING_STEP_MS + part_bytes / (XGMI_LINK_GBS * 1e9) * 1e3)
    merge = 0.0 if n == 1 else MERGE_MS_AT_8 * (n * k / 240.0) ** 2 * (nq / 21904.0)
    k5 = K5_MS_AT_1 * (nq / 21904.0) / n
    step = knn + gather + merge + (0.0 if n == 1 else k5)      # (N = 1: K5 is inside the measured step)
    one = SHARD_STEP_MS[1] * (nq / 21904.0) * (rows / 1e7) * (dim / 768.0) * ((SHARD_CLOCK_GHZ / float(clock_ghz)) if clock_ghz else 1.0)
    return {"n_gpus": n, "ms_per_step": step, "value": nq / (step * 1e-3), "efficiency_vs_n_times_one_gpu": one / n / step,
            "knn_ms": knn, "all_gather_ms": gather, "merge_ms": merge, "aggregate_ms": k5, "clock_ghz_assumed": float(clock_ghz) if clock_ghz else SHARD_CLOCK_GHZ,
            "constants": {"xgmi_link_GBs": XGMI_LINK_GBS, "ring_step_ms": RING_STEP_MS, "ring_base_ms": RING_BASE_MS, "merge_ms_at_8": MERGE_MS_AT_8,
                          "k5_ms_at_1": K5_MS_AT_1, "shard_step_ms": SHARD_STEP_MS}}
