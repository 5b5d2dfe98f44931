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

import bench_legs as legs  # noqa: E402
from bench_legs import PEAK_FP16_MFMA_TFLOPS, PEAK_FP32_MFMA_TFLOPS, NOMINAL_GHZ, safe, spread  # noqa: E402

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
    ap.add_argument("--no-ab", action="store_true", help="N = 1: skip the equal-XCD-shares A/B leg")
    ap.add_argument("--timed-only", action="store_true", help="N = 1: warm-up + timed steps and nothing else (no A/B legs, use_fp16, end to end, counter passes, CPU "
                    "baseline): what tools/gpu_profile.sh runs under rocprofv3, so that the kernel statistics are those of the timed kernel")
    ap.add_argument("--cpu-full-bank", choices=["auto", "never"], default="auto", help="CPU baseline at the full bank size (the bench's own rows fetched to "
                    "the host) when the host has the memory; never = the bounded sample only")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # one untimed step under rocprofv3 --pmc
    ap.add_argument("--cluster-shape", type=int, nargs=2, default=[0, 0], metavar=("Q", "B"), help="L2-sharing cluster shape of the fp32 searches "
                    "(0 0 = automatic: kept where they measure faster; 1 1 = none; the counter passes repeat the timed steps' form)")
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
    if a.timed_only:
        a.no_ab = a.no_e2e = a.no_counters = a.no_traffic = a.no_cpu_baseline = True
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
    if a.cluster_shape[0] > 0:
        index.set_cluster(a.cluster_shape[0], a.cluster_shape[1], 0 if tuple(a.cluster_shape) == (1, 1) else -1)
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
        selftest = legs.selftest_against_oracle(index, q, k, lo, hi, world, rank, device, search_merged)   # exits with status 4 on a mismatch
    for i in range(a.warmup):
        step(i)
        torch.cuda.synchronize(device)      # (a finished warm-up search lets the next one calibrate its per-XCD work shares: hb_index_set_xcd_weights)
    sync()
    plans_before = index.xcd_stats(bool(a.fp16))["work_lists_built"]
    index.set_timing(True)
    clock_ghz = []                          # per timed step: the kNN kernel's clock without a profiler (shader cycles per real-time tick, median workgroup)
    t0 = time.time()
    last_out = None
    for i in range(a.steps):
        last_out = step(i)
        knn_ms.append(index.last_knn_ms())        # waits for this step's kNN kernel (HIP events on its stream)
        clock_ghz.append(index.kernel_clock())    # (8 KB of stamps; the kernel has ended)
        if dist_on and not overlap:
            b = i & 1
            ev_done[b].synchronize()
            xchg_ms.append(ev_knn[b].elapsed_time(ev_done[b]))
            split_ms.append((ev_knn[b].elapsed_time(ev_ag[b]), ev_ag[b].elapsed_time(ev_mg[b]), ev_mg[b].elapsed_time(ev_done[b])))
    sync()
    dt = time.time() - t0
    index.set_timing(False)
    replans = index.xcd_stats(bool(a.fp16))["work_lists_built"] - plans_before
    ghz_steps = [c["ghz"] for c in clock_ghz if c["ghz"] > 0]
    ghz_med = float(np.median(ghz_steps)) if ghz_steps else None
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
        leg_list = []                   # every rank runs the same legs (they contain collectives); the numbers travel in `mine`
        cl_auto = tuple(index.schedule_info().get("cluster", (1, 1)))
        if not a.fp16 and cl_auto != (1, 1):
            index.set_cluster(1, 1, 0)
            leg_list.append(("without_clusters",) + timed_leg())
            index.set_cluster(0, 0, -1)
        if not a.fp16:
            index.set_fp16(True)
            ms16, k16 = timed_leg()
            leg_list.append(("use_fp16_mode", ms16, k16))
            fb16 = index.last_fp16_fallbacks()
            index.set_fp16(False)
        sp = np.mean(np.array(split_ms), axis=0) if split_ms else np.array([-1.0, -1.0, -1.0])
        mine = torch.tensor([float(np.mean(knn_ms)), float(np.mean(xchg_ms)) if xchg_ms else -1.0, float(hi - lo), float(dev_index),
                             float(sp[0]), float(sp[1]), float(sp[2]), float(ghz_med or 0.0)] + [x for lg in leg_list for x in lg[1:]],
                            device=device, dtype=torch.float64)
        ncol = mine.numel()
        allr = torch.empty(world * ncol, device=device, dtype=torch.float64)
        td.all_gather_into_tensor(allr, mine)
        per_rank = allr.view(world, ncol).cpu().tolist()
        for j, lg in enumerate(leg_list):
            extra_legs[lg[0]] = {"ms_per_step": lg[1], "value": nq / (lg[1] * 1e-3), "unit": "query-patches/s",
                                 "knn_ms_per_rank": [round(r[9 + 2 * j], 3) for r in per_rank]}
        if "use_fp16_mode" in extra_legs:
            extra_legs["use_fp16_mode"]["fallback_queries_rank0"] = fb16
            extra_legs["use_fp16_mode"]["note"] = "certified-exact fast mode, same outputs as the fp32 search; whole N-rank steps incl. the exchange"
        if "without_clusters" in extra_legs:
            extra_legs["without_clusters"]["cluster_in_timed_steps"] = list(cl_auto)

    res = None
    if rank == 0:
        kms = float(np.mean(knn_ms))
        flops = 2.0 * nq * (hi - lo) * D
        ach = flops / (kms * 1e-3) / 1e12
        # --fp16 prices the candidate kernel against the dense fp16 matrix peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s)
        peak = PEAK_FP16_MFMA_TFLOPS if a.fp16 else PEAK_FP32_MFMA_TFLOPS
        # family prefix of the dominant kernel as rocprofv3 names it (fp32: knn_fused_bd_kernel<WIDE>, small searches
        # knn_fused_kernel<...>; --fp16: knn_f16v2_kernel<4>)
        kernel = "knn_f16" if a.fp16 else "knn_fused"
        ksp = spread(knn_ms)
        shares, rounds = index.xcd_weights(bool(a.fp16))
        xst = index.xcd_stats(bool(a.fp16))
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
                       # work share per PHYSICAL XCD of the kNN work list, calibrated from the workgroups' own durations during the warm-up
                       # steps (hb_index_set_xcd_weights: the XCDs of one chip differ by 1-2 % in speed); as a string so that flat records keep it
                       "xcd_shares": ",".join(f"{v:.4f}" for v in shares), "xcd_calibration_rounds": rounds,
                       "xcd_guard_locked": xst["locked"], "xcd_guard_reverts": xst["reverts"], "xcd_stamp_sets_rejected": xst["rejected"],
                       "xcd_of_block0": xst["xcd_of_block0"]},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": None,
                         "traffic_unit": "bytes/launch (L2-miss side, rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; Infinity-Cache hits included)",
                         "kernel": kernel, "avg_kernel_ms": kms,
                         "kernel_ms_min": ksp["min"], "kernel_ms_median": ksp["median"], "kernel_ms_max": ksp["max"],
                         "algorithmic_flops_per_launch": flops,
                         "algorithmic_bytes_per_launch": 4.0 * (hi - lo) * D + 4.0 * nq * D + 12.0 * nq * k,
                         # the clock the timed kernels ran at, read by the kernels themselves (s_memtime / s_memrealtime stamps of every
                         # workgroup, no profiler attached): median over the steps of the median workgroup; slowest / fastest workgroup of any step
                         "clock_ghz_unprofiled": ghz_med,
                         "clock_ghz_unprofiled_min_wg": min((c["ghz_min"] for c in clock_ghz if c["ghz_min"] > 0), default=None),
                         "clock_ghz_unprofiled_max_wg": max((c["ghz_max"] for c in clock_ghz if c["ghz_max"] > 0), default=None),
                         "frac_of_peak_at_measured_clock": (ach / (peak * ghz_med / NOMINAL_GHZ)) if ghz_med else None,
                         "work_list_replans_in_timed_steps": replans},
        }
        if checksum is not None:
            res["label_hat_checksum"] = checksum
        if dist_on:
            pred = legs.scaling_model(world, nq, k, M, D)
            clocks = [r[7] for r in per_rank]
            pred_clk = legs.scaling_model(world, nq, k, M, D, clock_ghz=min(c for c in clocks if c > 0)) if any(c > 0 for c in clocks) else None
            measured_ms = dt / a.steps * 1e3
            one_gpu_equiv = sum(r[0] for r in per_rank)
            res["multi_gpu"] = {
                "backend": backend + (" (RCCL)" if backend == "nccl" else " (test mode, ranks share cuda:0)"),
                "world_size": td.get_world_size(), "ranks_seen_by_first_all_reduce": world,
                "device_per_rank": [int(r[3]) for r in per_rank], "rows_per_rank": [int(r[2]) for r in per_rank],
                "knn_ms_per_rank": [round(r[0], 3) for r in per_rank],
                "clock_ghz_unprofiled_per_rank": [round(r[7], 4) for r in per_rank],
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
                "n1_equivalent_ms": round(one_gpu_equiv, 3),
                "efficiency": one_gpu_equiv / world / measured_ms,
                "efficiency_definition": "(sum of the ranks' kNN kernel ms = the N=1-equivalent step) / N / measured ms_per_step",
                # the pre-registered model (bench_legs.scaling_model) beside the measurement
                "predicted": pred, "predicted_at_the_slowest_ranks_clock": pred_clk,
                "measured_over_predicted_ms": measured_ms / pred["ms_per_step"],
                "measured_over_predicted_ms_at_measured_clock": (measured_ms / pred_clk["ms_per_step"]) if pred_clk else None,
            }
            res["config"].update({"predicted_value": pred["value"], "predicted_efficiency": pred["efficiency_vs_n_times_one_gpu"],
                                  "measured_over_predicted_ms": measured_ms / pred["ms_per_step"],
                                  "ranks_seen": world, "min_rank_clock_ghz": min(clocks) if clocks else None,
                                  "exchange_ms_max_rank": None if overlap else max(r[1] for r in per_rank)})
            res.update(extra_legs)
        single = world == 1 and not dist_on
        if single and not a.fp16 and not a.no_ab:
            res["xcd_shares_ab"] = safe("equal_shares_leg", legs.equal_shares_leg, index, q, k, device)
        if single and not a.fp16 and not a.timed_only:
            ab = safe("clusters_ab_leg", legs.clusters_ab_leg, index, q, k, device, flops, peak)
            if "failed" not in ab:        # the timed steps' own number fills the slot of the form they ran
                ab["clustered_kernel_ms" if ab["clustered_kernel_ms"] is None else "unclustered_kernel_ms"] = kms
            res["clusters_ab"] = ab
            res["use_fp16_mode"] = safe("use_fp16_leg", legs.use_fp16_leg, index, q, k, device, flops, nq)
        if single and not a.no_e2e:
            res["e2e"] = safe("e2e_leg", legs.e2e_leg, index, D, C, k, nq, device, max(1, a.e2e_batches))
    if world == 1 and not dist_on:
        a.cluster_q, a.cluster_b = res["config"]["schedule"].get("cluster") or (1, 1)      # the fp32 counter passes repeat the timed steps' form
        # the CPU baseline at the full bank size wants the bench's own rows on the host: fetched before the index goes
        bank_host, fetch_note, q_host = None, None, None
        if not a.no_cpu_baseline and a.cpu_full_bank == "auto":
            got = safe("fetch_bank_to_host", legs.fetch_bank_to_host, index, M, D)
            if isinstance(got, tuple):
                bank_host, fetch_note = got
            else:
                fetch_note = got
            q_host = q[:256].cpu().numpy()
        # roofline.traffic: live counter passes (children of this process); else the committed figure of the same
        # workload IF it was measured on these very kernel sources; else null
        del index, agg
        torch.cuda.empty_cache()
        if not a.no_counters and not a.no_traffic:
            mp = safe("matrix_pipe_counters", legs.matrix_pipe_counters, a, kernel, a.fp16)
            res["roofline"].update({key: v for key, v in mp.items() if key in ("clock_ghz", "mfma_busy")})
            res["roofline"]["clock_and_busy_source"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (child pass, one search)"
            if isinstance(res.get("use_fp16_mode"), dict) and "failed" not in res["use_fp16_mode"]:
                res["use_fp16_mode"].update(safe("matrix_pipe_counters_fp16", legs.matrix_pipe_counters, a, "knn_f16", True))
        got = (None, "skipped (--no-traffic)") if a.no_traffic else safe("measure_traffic", legs.measure_traffic, a, kernel)
        traffic, note = got if isinstance(got, tuple) else (None, str(got))
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
            res["cpu_baseline"] = safe("cpu_baseline", legs.cpu_baseline, D, k, M, bank_host, q_host, fetch_note)
        del bank_host
        res["miou_parity"] = None if a.timed_only else safe("miou_parity", legs.miou_parity, device)
        flatten_into_roofline(res)
    if dist_on:
        td.destroy_process_group()
    if rank == 0:
        emit(res)


def flatten_into_roofline(res):
    """The driver's record keeps the scalars of `config`, `roofline` and `cpu_baseline` and only the NAMES of other top-level objects: the
    numbers that explain the line (the A/B legs, use_fp16, end to end) are therefore repeated as flat scalars inside `roofline`."""
    r = res["roofline"]

    def get(obj, *path):
        for p in path:
            if not isinstance(obj, dict) or p not in obj:
                return None
            obj = obj[p]
        return obj
    ab = res.get("xcd_shares_ab")
    r["equal_shares_kernel_ms"] = get(ab, "equal_shares_kernel_ms")
    r["calibrated_shares_kernel_ms"] = get(ab, "calibrated_shares_kernel_ms")
    r["calibrated_over_equal"] = get(ab, "calibrated_over_equal")
    r["clustered_kernel_ms"] = get(res, "clusters_ab", "clustered_kernel_ms"); r["unclustered_kernel_ms"] = get(res, "clusters_ab", "unclustered_kernel_ms")
    r["clusters_kept_by_measurement"] = get(res, "clusters_ab", "clusters_kept")
    r["cluster_shape_in_timed_steps"] = "x".join(str(v) for v in (get(res, "clusters_ab", "cluster_in_timed_steps") or [])) or None
    f = res.get("use_fp16_mode")
    r["fp16_value"] = get(f, "value"); r["fp16_ms_per_step"] = get(f, "ms_per_step")
    r["fp16_candidate_kernel_ms"] = get(f, "candidate_kernel_ms")
    r["fp16_frac_of_fp16_peak"] = get(f, "candidate_kernel_frac_of_fp16_mfma_peak")
    r["fp16_clock_ghz_unprofiled"] = get(f, "clock_ghz_unprofiled")
    r["fp16_clock_ghz_counters"] = get(f, "clock_ghz"); r["fp16_mfma_busy"] = get(f, "mfma_busy")
    r["fp16_fallback_queries"] = get(f, "fallback_queries"); r["fp16_first_certificate_failed"] = get(f, "first_certificate_failed")
    r["fp16_guard_locked"] = get(f, "calibration", "locked")
    e = res.get("e2e")
    r["e2e_fp32_images_per_s"] = get(e, "fp32", "images_per_s_steady_state"); r["e2e_fp16_images_per_s"] = get(e, "use_fp16", "images_per_s_steady_state")
    r["e2e_bound_by"] = get(e, "fp32", "bound_by")
    r["e2e_vit_forward_ms"] = get(e, "fp32", "per_batch_ms", "vit_forward_ms")
    r["miou_max_abs_delta_vs_reference"] = get(res, "miou_parity", "max_abs_miou_delta_vs_reference")


if __name__ == "__main__":
    main()
