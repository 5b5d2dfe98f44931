#!/usr/bin/env python3
"""Headline benchmark: query-patches/sec of the Hummingbird retrieval hot path on MI355X.

Workload (BASELINE.json metric: "768-d k=30 over 10M-patch bank", configs[2] shapes): a synthetic
10,000,000 x 768 fp32 memory bank (rows N(0,1), L2-normalised by the fused append kernel) with 151-class
soft labels, and query batches of 16 images x 1369 patches = 21,904 un-normalised 768-d tokens.
One step = one pass of the hot path over one query batch: query tiling -> exact brute-force kNN
(fused fp32-MFMA top-k kernel) -> partial-list merge -> cosine-softmax label aggregation, all inputs
already resident in HBM.  With N GPUs the bank is row-sharded (10M / N rows per rank, one process per
GPU), every rank searches all queries on its shard, the per-rank top-k lists are exchanged with ONE packed RCCL
all-gather per step and merged, and each rank aggregates the labels for its slice of the queries ("strong"
scaling: total work is fixed).  The exchange of step i (all-gather + merge + aggregation) runs on the kNN stream,
exposed after the kernel: the persistent kNN kernel owns every CU's registers and LDS, so nothing could run beside it
(DESIGN.md section 5).  HBIRD_BENCH_OVERLAP=1 (experiments only) moves the exchange to a side stream.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before this process touches a GPU); under
torchrun (RANK / WORLD_SIZE set, the driver's form) it is one of the ranks.  WORLD_SIZE != --gpus is an error.

Prints ONE JSON line on rank 0 (see README / the driver contract).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "open-hummingbird-eval_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_FP16_MFMA_TFLOPS = 2516.6   # 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz (dense, no sparsity)
LABEL_P = 196                    # pixels per patch of the synthetic soft labels (14 x 14): every label value is j / 196


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: WORLD_SIZE under torchrun, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total bank rows M")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--classes", type=int, default=151)
    ap.add_argument("--nq", type=int, default=16 * 1369, help="query patches per step")
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--panel", type=int, default=0)
    ap.add_argument("--fp16", action="store_true", help="use_fp16: fp16 candidate pass + exact fp32 re-rank")
    ap.add_argument("--variant", type=int, default=0, help="kNN kernel variant for A/B runs (hb_index_set_variant: 0 default, 2 / 3 / 4 / 6)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: keep the exchange on the kNN stream even when "
                    "HBIRD_BENCH_OVERLAP=1 asks for the side stream (the default is the kNN stream anyway)")
    ap.add_argument("--checksum", action="store_true", help="add label_hat_checksum (bit sum + float64 sum of the last step's "
                    "label_hat over all queries): equal for any number of ranks")
    ap.add_argument("--no-selftest", action="store_true", help="N > 1: skip the check of the merged neighbour lists of 64 queries against the "
                    "chain oracle (every rank searches its shard on the CPU, rank 0 merges) that runs before the timed steps")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the end-to-end leg (random-weight ViT -> HbirdEvaluation.evaluate on this bank)")
    ap.add_argument("--e2e-batches", type=int, default=3, help="validation batches of the end-to-end leg in fp32 mode (use_fp16 mode: twice as many)")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 --pmc pass behind clock_ghz / mfma_busy")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # one untimed step under rocprofv3 --pmc
    return ap.parse_args()


def images_note(nq):
    """' (B x N)' when the query count is a whole number of 37 x 37- or 14 x 14-token images, else nothing."""
    for n in (1369, 196):
        if nq % n == 0:
            return f" ({nq // n} x {n})"
    return ""


def kernel_source_hash():
    """sha256 over the HIP sources: a committed traffic figure is only quoted for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "open-hummingbird-eval_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(a):
    """--gpus N > 1 without a torchrun environment: start the N ranks as a CHILD process tree.  Nothing in this
    process has touched a GPU yet (torch.cuda.device_count() does not initialise one), and the parent only waits."""
    one_gpu = os.environ.get("HBIRD_BENCH_ONE_GPU") == "1"
    import torch
    have = torch.cuda.device_count()
    if not one_gpu and have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible on this node")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    argv = [x for x in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    # The ranks exchange device buffers through RCCL, whose intra-node transport maps its peers' buffers with HIP IPC
    # handles.  The host driver of this pool only supports dmabuf IPC: with the legacy mode (the ROCr default)
    # hipIpcGetMemHandle fails with "invalid argument" at communicator set-up.  The image exports the variable already;
    # a launcher that scrubs the environment must not lose it.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "VERSION")      # RCCL prints its version line to stderr: the run's own record of the backend
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    raise SystemExit(subprocess.call(cmd, env=env))


def build_bank(index, rows_lo, rows_hi, D, C, device):
    """Synthetic bank shard [rows_lo, rows_hi): rows in chunks of 500k, seeded by the global chunk id so
    the same global bank is produced for any sharding."""
    chunk = 500_000
    index.reserve(rows_hi - rows_lo)
    g = torch.Generator(device=device)
    r = rows_lo
    while r < rows_hi:
        c0 = (r // chunk) * chunk
        g.manual_seed(1000 + r // chunk)
        full = torch.randn((chunk, D), generator=g, device=device, dtype=torch.float32)
        lo, hi = r - c0, min(rows_hi, c0 + chunk) - c0
        index.add(full[lo:hi], normalize=True)        # K1: fused L2-normalise + fragment-tiled append
        # soft labels with values j/196 on two classes per row (K2 output shape/values)
        g.manual_seed(5000 + r // chunk)
        c1 = torch.randint(0, C, (chunk,), generator=g, device=device)
        c2 = torch.randint(0, C, (chunk,), generator=g, device=device)
        j = torch.randint(0, LABEL_P + 1, (chunk,), generator=g, device=device)
        j = torch.where(c1 == c2, torch.full_like(j, LABEL_P), j)        # one class: the whole patch
        # exactly what K2 produces: (float)count / (float)P per class (hbird_eval.py:319-320) -- a division by a 0-dim TENSOR (torch turns a
        # Python-scalar divisor into a multiplication by the reciprocal on the GPU, another rounding), so that the index can keep the rows
        # as uint16 counts (hb_index_set_label_denominator: half the table, what the evaluator does by default)
        P = torch.tensor(float(LABEL_P), device=device)
        lab = torch.zeros((chunk, C), device=device)
        lab.scatter_(1, c2[:, None], ((LABEL_P - j).float() / P)[:, None])
        lab.scatter_(1, c1[:, None], (j.float() / P)[:, None])
        index.add_labels(lab[lo:hi])
        del full, lab
        r = c0 + hi
    torch.cuda.synchronize(device)


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


def cpu_baseline(D, k, M_total):
    """The oracle's exact fp32 brute force (oracle/hbird_oracle.c, OpenMP + AVX2) on a bounded sample, on the cores the host grants
    (host_cpu_budget), scaled linearly in the bank size -- with a second, half-size sample that shows the scaling instead of
    asserting it; the host BLAS on the same product; ScaNN when it is installed.  Reported baseline only."""
    import oracle
    import torch as _t
    budget = host_cpu_budget()
    threads = budget["cores"]
    oracle.set_num_threads(threads)
    _t.set_num_threads(threads)
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
    return {
        "value": qps_sample * ms / M_total,
        "unit": "query-patches/s",
        "cores": threads,
        "host": budget,
        "kind": "port",
        "tuned": False,                             # the chain oracle is a parity tool (one fmaf chain per score), not a tuned SGEMM
        "extrapolated": True,                       # value = measured sample rate x (sample rows / bank rows)
        "measured_on_sample": {"value": qps_sample, "unit": "query-patches/s", "bank_rows": ms, "queries": nqs, "seconds": round(dt, 2)},
        "linearity_check": {"rows": [ms // 2, ms], "seconds": [round(dt_half, 2), round(dt, 2)],
                            "seconds_ratio": dt / dt_half, "expected": 2.0,
                            "what": "same queries against half the sample and the whole sample: brute force is linear in the bank rows, which is what the extrapolation uses"},
        "sample": f"oracle exact fp32 brute force on {nqs} queries x {ms} rows x {D} dims took {dt:.2f}s on {threads} threads "
                  f"({qps_sample:.1f} q/s), scaled x{ms}/{M_total} to the full bank",
        "scann": scann_leg,
        "torch_mm_topk": {"value": nqt / dt_t * ms / M_total, "unit": "query-patches/s", "threads": _t.get_num_threads(),
                          "sample_seconds": round(dt_t, 2)},
        "torch_mm_only": {"tflops": mm_tflops, "gflops_per_core": mm_tflops * 1e3 / threads, "value": nqs / dt_m * ms / M_total,
                          "unit": "query-patches/s (no k-select)", "threads": _t.get_num_threads(), "sample_seconds": round(dt_m, 2),
                          "blas": blas,
                          "what": f"fp32 [{nqs},{D}] x [{D},{ms}] products only: the host BLAS ceiling for the contraction on the granted cores"},
        "post_knn_stage": {"value": S * S / dt_p, "unit": "query-patches/s",
                           "what": "gather + cross-attention + bilinear upsample + argmax of one 37x37-token image, C=151"},
    }


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
            "--workgroups", str(a.workgroups), "--panel", str(a.panel), "--variant", str(a.variant)] + (["--fp16"] if fp16 else [])
    out = tempfile.mkdtemp(prefix="hbird_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child"] + args
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
            "--workgroups", str(a.workgroups), "--panel", str(a.panel), "--variant", str(a.variant)] + (["--fp16"] if a.fp16 else [])
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="hbird_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child"] + args
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
    for mode, fp16, n in (("fp32", False, n_batches), ("use_fp16", True, 2 * n_batches)):
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
                     "miou_of_random_weights": float(jac)}
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


def first_collective_or_die(td, device, backend, world, rank, seconds=None):
    """RCCL builds its communicator on the first collective.  If that does not complete (a peer that never arrives, IPC handles the
    driver refuses, a wedged link) the run would hang until the driver's clock kills it without a word: a watchdog thread prints
    ONE line to stderr and ends the process with status 3.  Nothing is re-executed."""
    import threading
    seconds = float(os.environ.get("HBIRD_BENCH_COMM_TIMEOUT", "180")) if seconds is None else seconds
    done = threading.Event()

    def watchdog():
        if not done.wait(seconds):
            sys.stderr.write(f"bench.py: rank {rank}/{world}: the {backend} communicator did not come up within {seconds:.0f} s "
                             f"(first all-reduce on {device}); check HSA_ENABLE_IPC_MODE_LEGACY=0, NCCL_DEBUG=INFO, visible GPUs\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    t = torch.ones(1, device=device)
    td.all_reduce(t)
    torch.cuda.synchronize(device)
    done.set()
    if int(t.item()) != world:
        raise SystemExit(f"bench.py: first all-reduce returned {t.item()} on rank {rank}, expected {world}")


_JSON_FD = None


def claim_stdout():
    """Rank 0's stdout must carry ONE line, the JSON.  Libraries print there too (RCCL's version banner, gloo's
    connection notes, tqdm), so file descriptor 1 is pointed at stderr for everybody else and the result line is written
    to a private duplicate of the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


_T_PROCESS = time.time()


def emit(res):
    # whole run of this process, imports excluded: the timed steps are ms_per_step x steps of it, the rest is the bank build, the warm-up
    # and the reported-only legs (use_fp16, end to end, counter passes, CPU baseline)
    res["wall_s"] = round(time.time() - _T_PROCESS, 1)
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, (json.dumps(res) + "\n").encode())


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if a.gpus is not None and a.gpus > 1:
            launch_ranks(a)                       # never returns: exits with the child's status
        world = 1
    else:
        world = int(env_world)
        if a.gpus is not None and a.gpus != world:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    claim_stdout()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # HBIRD_BENCH_ONE_GPU=1 (testing only): all ranks share cuda:0 and talk over gloo, since RCCL refuses two
    # ranks on one device; the normal path is one rank per GPU over RCCL.  HBIRD_BENCH_FORCE_DIST=1 (testing only) runs
    # the N-rank code path -- process group, packed all-gather, merge -- with a single rank.
    one_gpu = os.environ.get("HBIRD_BENCH_ONE_GPU") == "1"
    dist_on = world > 1 or os.environ.get("HBIRD_BENCH_FORCE_DIST") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    td = torch.distributed
    backend = None
    if dist_on:
        os.environ.setdefault("NCCL_DEBUG", "VERSION")      # RCCL's version line on stderr (read at communicator set-up)
        backend = "gloo" if one_gpu else "nccl"
        kw = {} if one_gpu else {"device_id": device}
        if env_world is None:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            td.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
        else:
            td.init_process_group(backend, **kw)
        assert td.get_world_size() == world and td.get_rank() == rank
        first_collective_or_die(td, device, backend, world, rank)
    from hbird_mi import dist as hdist
    from hbird_mi.nn.search_hip import HipFlatIndex, merge_topk_packed

    M, D, C, nq, k = a.rows, a.dim, a.classes, a.nq, a.k
    lo, hi = hdist.shard_range(M, rank, world)
    index = HipFlatIndex(D, 0, dev_index)
    index.set_num_classes(C)
    index.use_current_stream()
    if a.workgroups or a.panel:
        index.set_tuning(a.workgroups, a.panel)
    if a.variant:
        index.set_variant(a.variant)
    if a.fp16:
        index.set_fp16(True)
    index.set_label_denominator(LABEL_P)       # label rows j / 196 as uint16 counts: what HbirdEvaluation does by default (half the table)
    t_build = time.time()
    build_bank(index, lo, hi, D, C, device)
    agg = index
    if dist_on:
        # label rows and bank-row norms are small (3 GB of counts / 40 MB at cfg-3): replicate them once so that any
        # rank can aggregate the labels of any merged neighbour list -- as uint16 counts, travelling as bytes (gloo has no int16
        # collectives).  They hang off a second, row-less handle, whose workspace is independent of the searching index (the
        # aggregation may run on a side stream).
        cnt_local = index.copy_label_counts() if hi > lo else torch.zeros((0, C), dtype=torch.int16, device=device)
        cnt_all, counts = hdist.allgather_rows(cnt_local.contiguous().view(torch.uint8))
        nrm_all, _ = hdist.allgather_rows(index.copy_norms())
        agg = HipFlatIndex(D, 0, dev_index)
        agg.set_label_count_table(torch.cat([cnt_all[r, :counts[r]] for r in range(world)]).contiguous().view(torch.int16),
                                  torch.cat([nrm_all[r, :counts[r]] for r in range(world)]), LABEL_P, 0)
        del cnt_all, nrm_all, cnt_local
    torch.cuda.synchronize(device)
    t_build = time.time() - t_build

    g = torch.Generator(device=device)
    g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=device, dtype=torch.float32)
    qs_lo, qs_hi = (nq * rank) // world, (nq * (rank + 1)) // world

    if a.pmc_child:     # one untimed step for the counter passes of measure_traffic()
        index.search_aggregate(q, k, beta=0.02)
        torch.cuda.synchronize(device)
        return

    knn_ms, xchg_ms = [], []
    main_s = torch.cuda.current_stream(device)
    overlap = dist_on and not a.no_overlap and os.environ.get("HBIRD_BENCH_OVERLAP") == "1"
    side = torch.cuda.Stream(device) if overlap else main_s
    ex = [hdist.PackedTopK(nq, k, device, world) for _ in range(2)] if dist_on else None
    ev_knn = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev_ag = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev_mg = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev_done = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    split_ms = []                       # per timed step: (all-gather, merge, aggregation) ms of the exposed exchange
    used = [False, False]
    pending = []

    def step(i):
        if not dist_on:
            return index.search_aggregate(q, k, beta=0.02)
        # every rank searches all queries on its shard straight into its packed list; ONE all-gather of the packed
        # lists; merge in place; each rank then aggregates the labels for its own slice of the queries
        b = i & 1
        if used[b]:
            main_s.wait_event(ev_done[b])          # the packed buffers of step i-2 have been merged
        index.use_current_stream()
        index.search_scores(q, k, lo, out=(ex[b].idx, ex[b].dist))
        ev_knn[b].record(main_s)
        with torch.cuda.stream(side):
            side.wait_event(ev_knn[b])
            ex[b].gather()
            ev_ag[b].record(side)
            mi, md = merge_topk_packed(ex[b].recv, ex[b].part_bytes, world, nq, k, 0)
            ev_mg[b].record(side)
            agg.use_current_stream()
            out = agg.aggregate(q[qs_lo:qs_hi], mi[qs_lo:qs_hi].contiguous(), md[qs_lo:qs_hi].contiguous(), beta=0.02)
            ev_done[b].record(side)
        used[b] = True
        pending.append((ev_knn[b], ev_done[b]))
        return out

    def sync():
        torch.cuda.synchronize(device)
        if dist_on:
            td.barrier(**({} if one_gpu else {"device_ids": [dev_index]}))
            torch.cuda.synchronize(device)

    selftest = None
    if dist_on and not a.no_selftest:
        def search_merged():
            index.use_current_stream()
            index.search_scores(q, k, lo, out=(ex[0].idx, ex[0].dist))
            ex[0].gather()
            return merge_topk_packed(ex[0].recv, ex[0].part_bytes, world, nq, k, 0)
        selftest = selftest_against_oracle(index, q, k, lo, hi, world, rank, device, search_merged)   # exits with status 4 on a mismatch
    for i in range(a.warmup):
        step(i)
        torch.cuda.synchronize(device)      # (a finished warm-up search lets the next one calibrate its per-XCD work shares: hb_index_set_xcd_weights)
    sync()
    index.set_timing(True)
    t0 = time.time()
    last_out = None
    for i in range(a.steps):
        last_out = step(i)
        knn_ms.append(index.last_knn_ms())        # waits for this step's kNN kernel (HIP events on its stream)
        if dist_on and not overlap:
            b = i & 1
            ev_done[b].synchronize()
            xchg_ms.append(ev_knn[b].elapsed_time(ev_done[b]))
            split_ms.append((ev_knn[b].elapsed_time(ev_ag[b]), ev_ag[b].elapsed_time(ev_mg[b]), ev_mg[b].elapsed_time(ev_done[b])))
    sync()
    dt = time.time() - t0
    index.set_timing(False)
    per_rank = None
    checksum = None
    if a.checksum and last_out is not None:
        # every query's label_hat row is computed by exactly one rank from the merged (rank-count independent) neighbour
        # list, so both sums are the same for any number of ranks: int64 sum of the fp32 bit patterns + float64 sum
        lo_ = last_out.contiguous()
        cs = torch.stack([lo_.view(torch.int32).to(torch.int64).sum(), lo_.double().sum().view(torch.int64)])
        if world > 1:
            parts = torch.empty(world * 2, device=device, dtype=torch.int64)
            td.all_gather_into_tensor(parts, cs)
            parts = parts.view(world, 2)
            checksum = {"bits": int(parts[:, 0].sum().item()), "sum": float(parts[:, 1].contiguous().view(torch.float64).sum().item()),
                        "rows_per_rank": [(nq * (r + 1)) // world - (nq * r) // world for r in range(world)]}
        else:
            checksum = {"bits": int(cs[0].item()), "sum": float(cs[1:].view(torch.float64).item()), "rows_per_rank": [nq]}
    extra_legs = {}
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())

        def timed_leg(n_steps=3):
            """n_steps whole N-rank steps (search + exchange) of the current index settings: (max-over-ranks ms per step, this rank's kNN ms)."""
            for _ in range(5):      # (warm-up; finished searches let the next ones calibrate the per-XCD work shares)
                step(0); sync()
            index.set_timing(True)
            t1 = time.time(); km = []
            for i in range(n_steps):
                step(i); km.append(index.last_knn_ms())
            sync()
            d = torch.tensor([(time.time() - t1) / n_steps * 1e3], device=device, dtype=torch.float64)
            index.set_timing(False)
            td.all_reduce(d, op=td.ReduceOp.MAX)
            return float(d.item()), float(np.mean(km))
        legs = []                       # every rank runs the same legs (they contain collectives); the numbers travel in `mine`
        cl_auto = tuple(index.schedule_info().get("cluster", (1, 1)))
        if not a.fp16 and cl_auto != (1, 1):
            index.set_cluster(1, 1, 0)
            legs.append(("without_clusters",) + timed_leg())
            index.set_cluster(0, 0, -1)
        if not a.fp16:
            index.set_fp16(True)
            ms16, k16 = timed_leg()
            legs.append(("use_fp16_mode", ms16, k16))
            fb16 = index.last_fp16_fallbacks()
            index.set_fp16(False)
        sp = np.mean(np.array(split_ms), axis=0) if split_ms else np.array([-1.0, -1.0, -1.0])
        mine = torch.tensor([float(np.mean(knn_ms)), float(np.mean(xchg_ms)) if xchg_ms else -1.0, float(hi - lo), float(dev_index),
                             float(sp[0]), float(sp[1]), float(sp[2])] + [x for lg in legs for x in lg[1:]],
                            device=device, dtype=torch.float64)
        ncol = mine.numel()
        allr = torch.empty(world * ncol, device=device, dtype=torch.float64)
        td.all_gather_into_tensor(allr, mine)
        per_rank = allr.view(world, ncol).cpu().tolist()
        for j, lg in enumerate(legs):
            extra_legs[lg[0]] = {"ms_per_step": lg[1], "value": nq / (lg[1] * 1e-3), "unit": "query-patches/s",
                                 "knn_ms_per_rank": [round(r[8 + 2 * j], 3) for r in per_rank]}
        if "use_fp16_mode" in extra_legs:
            extra_legs["use_fp16_mode"]["fallback_queries_rank0"] = fb16
            extra_legs["use_fp16_mode"]["note"] = "certified-exact fast mode, same outputs as the fp32 search; whole N-rank steps incl. the exchange"
        if "without_clusters" in extra_legs:
            extra_legs["without_clusters"]["cluster_in_timed_steps"] = list(cl_auto)

    if rank == 0:
        kms = float(np.mean(knn_ms))
        flops = 2.0 * nq * (hi - lo) * D
        ach = flops / (kms * 1e-3) / 1e12
        # --fp16 prices the candidate kernel against the dense fp16 matrix peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s)
        peak = PEAK_FP16_MFMA_TFLOPS if a.fp16 else PEAK_FP32_MFMA_TFLOPS
        # family prefix of the dominant kernel as rocprofv3 names it (fp32: knn_fused_bd_kernel<WIDE>, small searches
        # knn_fused_kernel<...>; --fp16: knn_f16v2_kernel<4> / knn_f16_kernel)
        kernel = "knn_f16" if a.fp16 else "knn_fused"
        res = {
            "metric": "query-patches/sec", "value": nq * a.steps / dt, "unit": "query-patches/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16-candidates+f32-rerank" if a.fp16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"exact kNN + label aggregation, {M} x {D} fp32 bank, k={k}, "
                                   f"{nq} query patches/step{images_note(nq)}, C={C}",
                       "bank_rows": M, "dim": D, "k": k, "queries_per_step": nq, "classes": C,
                       "parallelism": f"bank-shard{world}" if world > 1 else "single-gpu",
                       "bank_build_s": round(t_build, 2), "schedule": index.schedule_info(),
                       "use_fp16": bool(a.fp16), "fp16_fallback_queries": index.last_fp16_fallbacks() if a.fp16 else None,
                       # work share per XCD group of the fp32 work list, calibrated from the workgroups' own durations during the warm-up
                       # steps (hb_index_set_xcd_weights: the XCDs of one chip differ by 1-2 % in speed), and the calibration rounds
                       "xcd_shares": [round(v, 4) for v in index.xcd_weights(bool(a.fp16))[0]], "xcd_calibration_rounds": index.xcd_weights(bool(a.fp16))[1]},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": None,
                         "traffic_unit": "bytes/launch (L2-miss side, rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; Infinity-Cache hits included)",
                         "kernel": kernel, "avg_kernel_ms": kms,
                         "algorithmic_flops_per_launch": flops,
                         "algorithmic_bytes_per_launch": 4.0 * (hi - lo) * D + 4.0 * nq * D + 12.0 * nq * k},
        }
        if checksum is not None:
            res["label_hat_checksum"] = checksum
        if dist_on:
            res["multi_gpu"] = {
                "backend": backend + (" (RCCL)" if backend == "nccl" else " (test mode, ranks share cuda:0)"),
                "world_size": td.get_world_size(), "device_per_rank": [int(r[3]) for r in per_rank], "rows_per_rank": [int(r[2]) for r in per_rank],
                "knn_ms_per_rank": [round(r[0], 3) for r in per_rank],
                "exchange_ms_per_rank": None if overlap else [round(r[1], 3) for r in per_rank],
                "exchange": "one packed all-gather of (id int64, score fp32) [nq,k] per rank + in-place k-way merge + "
                            "label aggregation of this rank's query slice" + (", on a side stream under the next step's kNN kernel" if overlap else
                            ", exposed after the kNN kernel (it owns every CU's registers, nothing can run beside it)"),
                "packed_list_bytes_per_rank": ex[0].part_bytes,
                "selftest": selftest if selftest is not None else "skipped (--no-selftest)",
                "replicated_label_table": f"uint16 counts (denominator {LABEL_P}), {M * C * 2 / 1e9:.2f} GB per rank",
                "exchange_split_ms_per_rank": None if overlap else {
                    "all_gather": [round(r[4], 3) for r in per_rank], "merge": [round(r[5], 3) for r in per_rank],
                    "aggregate": [round(r[6], 3) for r in per_rank]},
                # what ONE GPU would need for the same step = the shards' kernels one after the other (the kNN kernel is linear in
                # the rows: 10 M / 5 M / 2.5 M / 1.25 M rows measured 2297.9 / 1146.5 / 577.5 / 289.5 ms)
                "n1_equivalent_ms": round(sum(r[0] for r in per_rank), 3),
                "efficiency": sum(r[0] for r in per_rank) / world / (dt / a.steps * 1e3),
                "efficiency_definition": "(sum of the ranks' kNN kernel ms = the N=1-equivalent step) / N / measured ms_per_step",
            }
            res.update(extra_legs)
        if world == 1 and not dist_on and not a.fp16 and tuple(index.schedule_info().get("cluster", (1, 1))) != (1, 1):
            # extra, not the headline: the timed steps ran with the automatic L2-sharing clusters (the biggest searches:
            # -60 % fabric reads for under 1 % of kernel time); the same step without them, so that the price is on the line
            cl_auto = tuple(index.schedule_info()["cluster"])
            index.set_cluster(1, 1, 0)
            index.search_aggregate(q, k, beta=0.02); torch.cuda.synchronize(device)
            index.set_timing(True)
            kms_off = []
            for _ in range(3):
                index.search_aggregate(q, k, beta=0.02)
                kms_off.append(index.last_knn_ms())
            index.set_timing(False)
            index.set_cluster(0, 0, -1)
            res["without_clusters"] = {"cluster_in_timed_steps": list(cl_auto), "avg_kernel_ms": float(np.mean(kms_off)),
                                       "frac": flops / (float(np.mean(kms_off)) * 1e-3) / 1e12 / peak,
                                       "note": "hb_index_set_cluster(ix, 1, 1, 0); same outputs; roofline.traffic is of the timed (clustered) kernel"}
        if world == 1 and not dist_on and not a.fp16:
            # extra, not the headline: the same step in use_fp16 mode (fp16 candidate pass + certified exact fp32
            # re-rank; returns the identical bits, see DESIGN.md, `use_fp16`)
            index.set_fp16(True)
            for _ in range(8):      # (finished warm-up searches let the next ones calibrate the fp16 kernel's per-XCD work shares)
                index.search_aggregate(q, k, beta=0.02); torch.cuda.synchronize(device)
            index.set_timing(True)
            t1 = time.time()
            k16 = []
            for _ in range(3):
                index.search_aggregate(q, k, beta=0.02)
                k16.append(index.last_knn_ms())
            torch.cuda.synchronize(device)
            dt16 = (time.time() - t1) / 3
            index.set_timing(False)
            res["use_fp16_mode"] = {"value": nq / dt16, "unit": "query-patches/s", "ms_per_step": dt16 * 1e3,
                                    "fallback_queries": index.last_fp16_fallbacks(),
                                    "candidate_kernel_ms": float(np.mean(k16)),
                                    "candidate_kernel_frac_of_fp16_mfma_peak": flops / (float(np.mean(k16)) * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                    "xcd_shares": [round(v, 4) for v in index.xcd_weights(True)[0]],
                                    "note": "certified-exact fast mode, same outputs as the fp32 search"}
            index.set_fp16(False)
        if world == 1 and not dist_on and not a.no_e2e:
            try:
                res["e2e"] = e2e_leg(index, D, C, k, nq, device, max(1, a.e2e_batches))
            except Exception as e:          # an extra leg never costs the bench line
                res["e2e"] = {"failed": repr(e)}
    if world == 1 and not dist_on:
        # roofline.traffic: live counter passes (children of this process); else the committed figure of the same
        # workload IF it was measured on these very kernel sources; else null
        del index, agg
        torch.cuda.empty_cache()
        if not a.no_counters and not a.no_traffic:
            res["roofline"].update({key: v for key, v in matrix_pipe_counters(a, kernel, a.fp16).items()
                                    if key in ("clock_ghz", "mfma_busy")})
            res["roofline"]["clock_and_busy_source"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (child pass, one search)"
            if "use_fp16_mode" in res:
                res["use_fp16_mode"].update(matrix_pipe_counters(a, "knn_f16", True))
        traffic, note = (None, "skipped (--no-traffic)") if a.no_traffic else measure_traffic(a, kernel)
        if traffic is not None and " on " in note:     # the counter pass saw the instantiation's full name
            res["roofline"]["kernel"] = note.split(" on ", 1)[1].split(":", 1)[0]
        if traffic is None:
            tpath = os.path.join(ROOT, "profiles", "latest_knn_traffic.json")
            if os.path.exists(tpath):
                t = json.load(open(tpath))
                if t.get("workload") == {"bank_rows": M, "dim": D, "k": k, "queries_per_step": nq} and not a.fp16 \
                        and t.get("kernel_source_hash") == kernel_source_hash():
                    traffic, note = t["traffic_bytes_per_launch"], note + "; committed profiles/latest_knn_traffic.json (same kernel sources)"
        res["roofline"]["traffic"] = traffic
        res["roofline"]["traffic_source"] = note
        if not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(D, k, M)
        res["miou_parity"] = miou_parity(device)
    if dist_on:
        td.destroy_process_group()
    if rank == 0:
        emit(res)


if __name__ == "__main__":
    main()
