"""`NearestNeighborSearchHIP`: the exact flat kNN backend on MI355X.

Drop-in for the reference's `NearestNeighborSearchFaiss` (hbird/nn/search_faiss.py:6-90): same
constructor keywords, same `find_nearest_neighbors(q, k=None) -> (indices, distances)` contract, same
exception types -- but the arithmetic is libhbird_hip.so's fused MFMA top-k kernel instead of
faiss-gpu.  One process drives ONE GPU; with torch.distributed initialised, `idx_shard=True` row-shards
the bank over the ranks (faiss.IndexShards, search_faiss.py:53-63) and merges the per-rank top-k after an
RCCL all-gather, `idx_shard=False` keeps a full replica per rank (faiss.IndexReplicas, 65-74).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch

from hbird_mi import _lib
from hbird_mi.nn.search_base import NearestNeighborSearchBase

_METRICS = {"dot_product": 0, "l2": 1, "euclidean": 1}
MAX_K = 256          # of the fused search + label aggregation (HbirdEvaluation) and of sharded searches: HB_MAX_K_AGGREGATE
MAX_K_SEARCH = 2048  # of a plain search on one GPU -- faiss-gpu's own limit (the reference forwards any k, search_faiss.py:84-85): HB_MAX_K


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


class HipFlatIndex:
    """Thin owner of one `hb_index_t*` (one GPU)."""

    def __init__(self, d: int, metric: int, device: int):
        self._h = ctypes.c_void_p()
        self.d, self.metric, self.device = int(d), int(metric), int(device)
        L = _lib.lib()
        rc = L.hb_index_create(self.d, self.metric, self.device, ctypes.byref(self._h))
        if rc != 0:
            msg = _lib.last_error()
            # same exception types as search_faiss.py:16 (no GPU) and :25 (bad GPU id)
            if "no GPUs" in msg:
                raise RuntimeError("No GPUs available for the HIP index.")
            raise ValueError(msg)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().hb_index_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ntotal(self) -> int:
        return int(_lib.lib().hb_index_ntotal(self._h))

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().hb_index_set_stream(self._h, ctypes.c_void_p(s)))

    def reserve(self, n: int):
        _lib.check(_lib.lib().hb_index_reserve(self._h, int(n)))

    def reset(self):
        _lib.check(_lib.lib().hb_index_reset(self._h))

    def add(self, x, normalize: bool = False):
        """x: float32 [n, d], numpy / CPU tensor (host path) or CUDA tensor on this GPU (device path)."""
        on_dev, x = self._as_f32(x)
        assert x.shape[1] == self.d, f"expected [n, {self.d}] rows, got {tuple(x.shape)}"
        _lib.check(_lib.lib().hb_index_add(self._h, _ptr(x), x.shape[0], int(on_dev), int(bool(normalize))))

    def add_labels(self, lab):
        on_dev, lab = self._as_f32(lab)
        _lib.check(_lib.lib().hb_index_add_labels(self._h, _ptr(lab), lab.shape[0], lab.shape[1], int(on_dev)))

    def _as_f32(self, x):
        if isinstance(x, torch.Tensor):
            x = x.detach()
            if x.dtype != torch.float32:
                x = x.float()
            x = x.contiguous()
            if x.is_cuda:
                assert x.device.index == self.device, "tensor lives on another GPU than the index"
                return True, x
            return False, x
        return False, np.ascontiguousarray(x, dtype=np.float32)

    def search_scores(self, q, k: int, id_base: int = 0, out=None):
        """Like `search`, but the second output holds the ORDERING scores (larger is better; L2: q.b - |b|^2/2), the
        input of a cross-shard merge (hb_index_set_score_output); convert with `distances_from_scores`."""
        lib = _lib.lib()
        _lib.check(lib.hb_index_set_score_output(self._h, 1))
        try:
            return self.search(q, k, id_base, out=out)
        finally:
            _lib.check(lib.hb_index_set_score_output(self._h, 0))

    def distances_from_scores(self, q: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        """Merged ordering scores [nq,k] (CUDA) -> the metric's distances, in place (no-op for the inner product)."""
        assert q.is_cuda and scores.is_cuda and scores.is_contiguous() and scores.dtype == torch.float32
        q = q.contiguous().float()
        _lib.check(_lib.lib().hb_index_distances_from_scores(self._h, _ptr(q), q.shape[0], scores.shape[1], _ptr(scores)))
        return scores

    def search(self, q, k: int, id_base: int = 0, out=None):
        """-> (idx int64 [nq,k], dist float32 [nq,k]); torch CUDA tensors for CUDA queries, numpy otherwise.
        out = (idx, dist): contiguous CUDA tensors to write into (e.g. the views of a dist.PackedTopK)."""
        on_dev, q = self._as_f32(q)
        nq = q.shape[0]
        if out is not None:
            idx, dist = out
            assert on_dev and idx.is_cuda and dist.is_cuda and idx.is_contiguous() and dist.is_contiguous()
            assert idx.dtype == torch.int64 and dist.dtype == torch.float32 and tuple(idx.shape) == tuple(dist.shape) == (nq, k)
        elif on_dev:
            idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        else:
            idx = np.empty((nq, k), dtype=np.int64)
            dist = np.empty((nq, k), dtype=np.float32)
        _lib.check(_lib.lib().hb_index_search(self._h, _ptr(q), nq, int(k), int(id_base), _ptr(idx), _ptr(dist),
                                               int(on_dev)))
        return idx, dist

    def search_aggregate(self, q, k: int, beta: float = 0.02, id_base: int = 0, want_neighbours: bool = False):
        on_dev, q = self._as_f32(q)
        nq = q.shape[0]
        c = self.num_classes
        if on_dev:
            out = torch.empty((nq, c), dtype=torch.float32, device=q.device)
            idx = torch.empty((nq, k), dtype=torch.int64, device=q.device) if want_neighbours else None
            dist = torch.empty((nq, k), dtype=torch.float32, device=q.device) if want_neighbours else None
        else:
            out = np.empty((nq, c), dtype=np.float32)
            idx = np.empty((nq, k), dtype=np.int64) if want_neighbours else None
            dist = np.empty((nq, k), dtype=np.float32) if want_neighbours else None
        _lib.check(_lib.lib().hb_index_search_aggregate(self._h, _ptr(q), nq, int(k), int(id_base), float(beta),
                                                         _ptr(out), _ptr(idx), _ptr(dist), int(on_dev)))
        return (out, idx, dist) if want_neighbours else out

    def aggregate(self, q, idx, dist, beta: float = 0.02, id_base: int = 0):
        """Label aggregation on given neighbours (CUDA tensors)."""
        assert q.is_cuda and idx.is_cuda and dist.is_cuda
        q = q.contiguous().float(); idx = idx.contiguous(); dist = dist.contiguous()
        out = torch.empty((q.shape[0], self.num_classes), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().hb_index_aggregate(self._h, _ptr(q), q.shape[0], _ptr(idx), _ptr(dist), idx.shape[1],
                                                  int(id_base), float(beta), _ptr(out), 1))
        return out

    def aggregate_partial(self, q, idx, dist, norms_all: torch.Tensor, beta: float = 0.02, id_base: int = 0):
        """Label-sharded aggregation: the softmax-weighted label sum over the neighbours THIS index owns (global ids id_base ..),
        with the weights of the full neighbour list (norms_all: the bank-row norms of all rows, global ids from 0).  The sum
        over the shards (an all-reduce) is label_hat."""
        assert q.is_cuda and idx.is_cuda and dist.is_cuda and norms_all.is_cuda
        q = q.contiguous().float(); idx = idx.contiguous(); dist = dist.contiguous(); norms_all = norms_all.contiguous().float()
        if self.ntotal == 0:                 # an empty shard owns no neighbour
            return torch.zeros((q.shape[0], self.num_classes), dtype=torch.float32, device=q.device)
        out = torch.empty((q.shape[0], self.num_classes), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().hb_index_aggregate_partial(self._h, _ptr(q), q.shape[0], _ptr(idx), _ptr(dist), idx.shape[1], int(id_base),
                                                         float(beta), _ptr(norms_all), norms_all.shape[0], _ptr(out)))
        return out

    @property
    def num_classes(self) -> int:
        return int(self._c) if hasattr(self, "_c") else self._query_c()

    def _query_c(self):
        raise RuntimeError("labels were not added to this index")

    def set_num_classes(self, c: int):
        self._c = int(c)

    def reconstruct(self, ids, id_base: int = 0):
        on_dev = isinstance(ids, torch.Tensor) and ids.is_cuda
        if on_dev:
            ids = ids.contiguous().to(torch.int64)
            out = torch.empty((ids.numel(), self.d), dtype=torch.float32, device=ids.device)
        else:
            ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int64)
            out = np.empty((ids.size, self.d), dtype=np.float32)
        n = ids.numel() if on_dev else ids.size
        _lib.check(_lib.lib().hb_index_reconstruct(self._h, _ptr(ids), n, int(id_base), _ptr(out), int(on_dev)))
        return out

    def gather_labels(self, ids):
        on_dev = isinstance(ids, torch.Tensor) and ids.is_cuda
        c = self.num_classes
        if on_dev:
            ids = ids.contiguous().to(torch.int64)
            out = torch.empty((ids.numel(), c), dtype=torch.float32, device=ids.device)
        else:
            ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int64)
            out = np.empty((ids.size, c), dtype=np.float32)
        n = ids.numel() if on_dev else ids.size
        _lib.check(_lib.lib().hb_index_gather_labels(self._h, _ptr(ids), n, 0, _ptr(out), int(on_dev)))
        return out

    def copy_norms(self) -> torch.Tensor:
        """L2 norms of this shard's stored rows (CUDA tensor [ntotal])."""
        out = torch.empty((self.ntotal,), dtype=torch.float32, device=torch.device("cuda", self.device))
        if self.ntotal:
            _lib.check(_lib.lib().hb_index_copy_norms(self._h, _ptr(out), 1))
        return out

    def set_label_denominator(self, P: int):
        """Every label value is j / P (P = patch_size ** 2: one_hot(...).mean over a patch's P pixels, hbird_eval.py:319-320): the
        index then stores the uint16 counts j -- half the table -- and hands back the same fp32 values.  Before the first label row."""
        _lib.check(_lib.lib().hb_index_set_label_denominator(self._h, int(P)))

    def labels_to_fp32(self):
        """The stored counts back to fp32 rows, in place (hb_index_labels_to_fp32): a bank whose batches come in a second patch size."""
        _lib.check(_lib.lib().hb_index_labels_to_fp32(self._h))

    @property
    def label_denominator(self) -> int:
        P = ctypes.c_int(0)
        _lib.check(_lib.lib().hb_index_label_denominator(self._h, ctypes.byref(P)))
        return int(P.value)

    def copy_label_counts(self) -> torch.Tensor:
        """This shard's label rows as counts: int16 tensor [nlabels, C] holding the uint16 bit patterns (CUDA)."""
        n = int(_lib.lib().hb_index_nlabels(self._h))
        out = torch.empty((n, self.num_classes), dtype=torch.int16, device=torch.device("cuda", self.device))
        if n:
            _lib.check(_lib.lib().hb_index_copy_label_counts(self._h, _ptr(out), 1))
        return out

    def set_label_count_table(self, counts: torch.Tensor, norms: torch.Tensor, P: int, id_base: int = 0):
        """set_label_table for a table held as counts (int16 tensor of uint16 bit patterns, denominator P)."""
        counts = counts.contiguous(); norms = norms.contiguous().float()
        assert counts.is_cuda and norms.is_cuda and counts.dtype == torch.int16 and counts.shape[0] == norms.shape[0]
        self._tables = (counts, norms)     # keep alive: the index only borrows the pointers
        self._c = int(counts.shape[1])
        _lib.check(_lib.lib().hb_index_set_label_count_table(self._h, _ptr(counts), _ptr(norms), counts.shape[0], counts.shape[1],
                                                             int(P), int(id_base)))

    def set_label_table(self, labels: Optional[torch.Tensor], norms: Optional[torch.Tensor], id_base: int = 0):
        """Borrow all-gathered label / norm tables that cover global ids [id_base, id_base + n)."""
        if labels is None:
            self._tables = None
            _lib.check(_lib.lib().hb_index_set_label_table(self._h, None, None, 0, 0, 0))
            return
        labels = labels.contiguous().float(); norms = norms.contiguous().float()
        assert labels.is_cuda and norms.is_cuda and labels.shape[0] == norms.shape[0]
        self._tables = (labels, norms)     # keep alive: the index only borrows the pointers
        self._c = int(labels.shape[1])
        _lib.check(_lib.lib().hb_index_set_label_table(self._h, _ptr(labels), _ptr(norms), labels.shape[0],
                                                       labels.shape[1], int(id_base)))

    def set_timing(self, on: bool):
        _lib.check(_lib.lib().hb_index_set_timing(self._h, int(on)))

    def last_knn_ms(self) -> float:
        ms = ctypes.c_double(0.0)
        _lib.check(_lib.lib().hb_index_last_knn_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def set_tuning(self, workgroups: int = 0, panel_tiles: int = 0):
        _lib.check(_lib.lib().hb_index_set_tuning(self._h, int(workgroups), int(panel_tiles)))

    def set_fp16(self, enable):
        """fp16 candidate pass + exact fp32 re-rank (GpuIndexFlatConfig.useFloat16 of the reference).  True / 1: always;
        2: only where it is faster than the fp32 kernel -- banks of at least 4,096 rows with rows x queries x D >= 1.5e10 x (k' / 64)^2 (same results
        either way)."""
        _lib.check(_lib.lib().hb_index_set_fp16(self._h, 2 if enable == 2 and enable is not True else int(bool(enable))))

    def last_fp16_fallbacks(self) -> int:
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().hb_index_last_fp16_fallbacks(self._h, ctypes.byref(n)))
        return int(n.value)

    def last_fp16_escalated(self) -> int:
        """Queries of the last use_fp16 search whose first certificate failed (they got the second, wider fp16 pass; last_fp16_fallbacks():
        those that reached the fp32 kernel)."""
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().hb_index_last_fp16_escalated(self._h, ctypes.byref(n)))
        return int(n.value)

    def set_fp16_escalation(self, on: bool = True):
        """use_fp16: uncertified queries get a second fp16 pass (k' = 256, seeded floors) before the fp32 kernel (default), or go straight to it."""
        _lib.check(_lib.lib().hb_index_set_fp16_escalation(self._h, 0 if on else 1))

    def set_variant(self, variant: int):
        _lib.check(_lib.lib().hb_index_set_variant(self._h, int(variant)))

    def set_search_options(self, phases: bool = True, small_limit_stages: int = 0):
        """A/B switches (same results): pool searches in one launch instead of phases; the size below which a search is "small"."""
        _lib.check(_lib.lib().hb_index_set_search_options(self._h, int(bool(phases)), int(small_limit_stages)))

    def set_xcd_weights(self, mode: int = 0, w8=None):
        """Work share per XCD group (blocks equal mod 8) of every panel of the work list (speed only, same results): mode 0 = calibrated from
        the workgroups' own durations (default), 1 = equal shares, 2 = the eight shares `w8`."""
        arr = None if w8 is None else (ctypes.c_double * 8)(*[float(v) for v in w8])
        _lib.check(_lib.lib().hb_index_set_xcd_weights(self._h, int(mode), arr))

    def xcd_weights(self, fp16_kernel: bool = False):
        """(the eight shares in use by the fp32 kernels / by the fp16 candidate kernel, calibration rounds so far)"""
        arr = (ctypes.c_double * 8)(); r = ctypes.c_int(0)
        _lib.check(_lib.lib().hb_index_xcd_weights(self._h, int(bool(fp16_kernel)), arr, ctypes.byref(r)))
        return [float(v) for v in arr], int(r.value)

    def wg_stamps(self):
        """After a search with set_timing(True): int64 array [workgroups, 3] = (start, end) in 100 MHz ticks relative to the earliest start,
        XCC id -- of the last kNN launch (hb_index_wg_stamps)."""
        buf = np.zeros((1024, 4), dtype=np.uint32)
        g = ctypes.c_int(0)
        _lib.check(_lib.lib().hb_index_wg_stamps(self._h, _ptr(buf), 1024, ctypes.byref(g)))
        t = buf[: g.value].astype(np.int64)
        if t.size:
            t0 = t[:, 0].min()
            t[:, 0] = (t[:, 0] - t0) & 0xFFFFFFFF; t[:, 1] = (t[:, 1] - t0) & 0xFFFFFFFF
        return t[:, :3]

    def kernel_clock(self) -> dict:
        """Clock of the last stamped kNN launch without a profiler attached (hb_index_kernel_clock): median / slowest / fastest workgroup in
        GHz (shader cycles per real-time tick) and the launch's span in ms; zeros when the launch did not stamp."""
        out = (ctypes.c_double * 4)()
        _lib.check(_lib.lib().hb_index_kernel_clock(self._h, out))
        return {"ghz": float(out[0]), "ghz_min": float(out[1]), "ghz_max": float(out[2]), "span_ms": float(out[3])}

    def xcd_stats(self, fp16_kernel: bool = False) -> dict:
        """State of the share calibration (hb_index_xcd_stats): rounds, the guard's lock / reverts, stamp sets read / rejected, shortest launch
        with the best / the current shares (ms), work lists built for this index."""
        out = (ctypes.c_double * 12)()
        _lib.check(_lib.lib().hb_index_xcd_stats(self._h, int(bool(fp16_kernel)), out))
        keys = ["rounds", "locked", "reverts", "samples", "rejected"]
        d = {k: int(out[i]) for i, k in enumerate(keys)}
        d.update({"best_span_ms": float(out[5]), "cur_span_ms": float(out[6]), "work_lists_built": int(out[7]), "xcd_map_moves": int(out[8]),
                  "xcd_of_block0": int(out[9]), "clusters_kept": int(out[10]), "clustered_minus_unclustered_ms": float(out[11])})
        return d

    def set_rerank_copy(self, mode: int = 0):
        """use_fp16 searches: a second, row-major fp32 copy of the bank for the exact re-rank (speed only; 0 automatic -- made for banks of up
        to 16 GB when 2.5 x the bank stays within 55 % of the device's memory: beyond that it buys under 2 % --, 1 always, 2 never)."""
        _lib.check(_lib.lib().hb_index_set_rerank_copy(self._h, int(mode)))

    def rerank_copy_bytes(self) -> int:
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().hb_index_rerank_copy_bytes(self._h, ctypes.byref(n)))
        return int(n.value)

    def set_cluster(self, cluster_q: int = 0, cluster_b: int = 0, sync_lag: int = -1):
        """L2-sharing clusters of the work list (speed only, opt-in): 0 x 0 / 1 x 1 off, e.g. 2 x 2; sync_lag in stages."""
        _lib.check(_lib.lib().hb_index_set_cluster(self._h, int(cluster_q), int(cluster_b), int(sync_lag)))

    def set_cluster_sharing(self, mode: int = 0):
        """How a clustered work list is dealt (speed only): 0 automatic, 1 per-cluster ranges, 2 XCD-level query-tile sharing."""
        _lib.check(_lib.lib().hb_index_set_cluster_sharing(self._h, int(mode)))

    def cluster_stats(self) -> dict:
        out = (ctypes.c_int64 * 4)()
        _lib.check(_lib.lib().hb_index_cluster_stats(self._h, out))
        return {"checks": int(out[0]), "waits": int(out[1]), "timeouts": int(out[2])}

    def schedule_info(self) -> dict:
        out = (ctypes.c_int64 * 8)()
        _lib.check(_lib.lib().hb_index_schedule_info(self._h, out))
        keys = ["workgroups", "segments", "slots", "panel_tiles", "max_slots_per_qtile", "query_tiles", "bank_tiles"]
        d = dict(zip(keys, list(out)[:7]))
        d["cluster"] = [int(out[7]) // 16, int(out[7]) % 16]
        return d


def merge_topk(dist_parts: torch.Tensor, idx_parts: torch.Tensor, metric: int):
    """[parts, nq, k] CUDA tensors -> merged (idx [nq,k], dist [nq,k]); hb_merge_topk."""
    parts, nq, k = dist_parts.shape
    dist_parts = dist_parts.contiguous(); idx_parts = idx_parts.contiguous()
    idx = torch.empty((nq, k), dtype=torch.int64, device=dist_parts.device)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dist_parts.device)
    s = torch.cuda.current_stream(dist_parts.device).cuda_stream
    _lib.check(_lib.lib().hb_merge_topk(_ptr(dist_parts), _ptr(idx_parts), parts, nq, k, int(metric), _ptr(idx),
                                        _ptr(dist), ctypes.c_void_p(s)))
    return idx, dist


def merge_topk_packed(recv: torch.Tensor, part_bytes: int, parts: int, nq: int, k: int, metric: int):
    """The gathered buffer of a dist.PackedTopK (CUDA, `parts` packed lists part_bytes apart) -> merged (idx, dist);
    hb_merge_topk_packed reads it in place."""
    assert recv.is_cuda and recv.is_contiguous() and recv.numel() * recv.element_size() >= parts * part_bytes
    idx = torch.empty((nq, k), dtype=torch.int64, device=recv.device)
    dist = torch.empty((nq, k), dtype=torch.float32, device=recv.device)
    s = torch.cuda.current_stream(recv.device).cuda_stream
    _lib.check(_lib.lib().hb_merge_topk_packed(_ptr(recv), int(part_bytes), int(parts), int(nq), int(k), int(metric), _ptr(idx),
                                               _ptr(dist), ctypes.c_void_p(s)))
    return idx, dist


class HipMultiIndex:
    """Several `hb_index_t` of ONE process behind the `HipFlatIndex` interface: what the reference's single-process call
    does with faiss (search_faiss.py:50-76).  `shard=True` = faiss.IndexShards (53-63): contiguous row ranges with
    successive ids, one per listed GPU, every GPU searches all queries, the [nq, k] lists are peer-copied to the first
    GPU and merged there on the ordering scores (the single-index bits).  `shard=False` = faiss.IndexReplicas (65-74):
    every GPU holds all rows, the queries are split.  One host thread per index (faiss `threaded=True`, 57).

    The FIRST listed device is the home device: queries arrive there, results are returned there, and the label rows
    and bank-row norms of ALL rows live there (the aggregation runs on it; 6.2 GB at BASELINE cfg-3).  A GPU may be
    listed more than once (several indices on one device: how the 1-GPU test boxes exercise this path)."""

    def __init__(self, d: int, metric: int, devices, shard: bool):
        assert len(devices) >= 1
        self.d, self.metric, self.devices, self.shard = int(d), int(metric), [int(g) for g in devices], bool(shard)
        self.device = self.devices[0]
        self.home = torch.device("cuda", self.device)
        self.indexes = [HipFlatIndex(d, metric, g) for g in self.devices]
        self.agg = HipFlatIndex(d, metric, self.device)        # row-less handle: aggregation against the home tables
        # peer access between the home device and every other listed GPU (xGMI on one node): without it a cross-device copy is staged
        # through pinned host memory explicitly -- one warning, same results
        self._staged = set()
        for g in sorted(set(self.devices)):
            if g != self.device and torch.cuda.is_available() and not (torch.cuda.can_device_access_peer(self.device, g)
                                                                         and torch.cuda.can_device_access_peer(g, self.device)):
                self._staged.add(g)
        if self._staged:
            import warnings
            warnings.warn(f"HipMultiIndex: no peer access between cuda:{self.device} and {sorted(self._staged)}: rows, queries and neighbour "
                          "lists are staged through host memory (slower copies, same results)", RuntimeWarning, stacklevel=2)
        self._warned_no_plan = False
        self._label_P = 0               # > 0: label rows are kept as int16 counts j of values j / P (set_label_denominator)
        self._quota = None              # shard mode: planned rows per shard (reserve); None = everything into the first
        self._labels = None             # [cap, C] on the home device, co-indexed with the global row ids
        self._nlab = 0
        self._norms = None              # [ntotal] on the home device (rebuilt lazily after adds)
        self._pool = None
        self._c = None
        self._fallbacks = 0

    # -- bookkeeping ------------------------------------------------------------------------------------------------
    def close(self):
        for ix in self.indexes + [self.agg]:
            ix.close()
        if self._pool is not None:
            self._pool.shutdown(wait=False)
            self._pool = None

    @property
    def ntotal(self) -> int:
        return sum(ix.ntotal for ix in self.indexes) if self.shard else self.indexes[0].ntotal

    @property
    def shard_rows(self):
        return [ix.ntotal for ix in self.indexes]

    @property
    def shard_bases(self):
        if not self.shard:
            return [0] * len(self.indexes)
        out, b = [], 0
        for ix in self.indexes:
            out.append(b); b += ix.ntotal
        return out

    @property
    def num_classes(self) -> int:
        if self._c is None:
            raise RuntimeError("labels were not added to this index")
        return self._c

    def set_num_classes(self, c: int):
        self._c = int(c)

    def use_current_stream(self):
        """Work on the home device follows the caller's current stream there; the per-GPU searches run on their worker
        threads' streams and are synchronised before their lists are handed over."""
        self.agg.use_current_stream()

    def set_fp16(self, enable):
        for ix in self.indexes:
            ix.set_fp16(enable)

    def set_rerank_copy(self, mode: int = 0):
        for ix in self.indexes:
            ix.set_rerank_copy(mode)

    def last_fp16_fallbacks(self) -> int:
        return self._fallbacks

    def schedule_info(self) -> dict:
        return self.indexes[0].schedule_info()

    def reserve(self, n: int):
        """Plan for n rows in all: shard mode cuts them into equal contiguous ranges (the last shard also takes whatever
        arrives beyond the plan)."""
        n = max(1, int(n))
        if self.shard:
            per = (n + len(self.indexes) - 1) // len(self.indexes)
            self._quota = per
            for ix in self.indexes:
                ix.reserve(per)
        else:
            for ix in self.indexes:
                ix.reserve(n)

    def reset(self):
        for ix in self.indexes:
            ix.reset()
        self._nlab, self._norms = 0, None
        self.agg.set_label_table(None, None)

    # -- bank build -------------------------------------------------------------------------------------------------
    def _on(self, i):
        return torch.cuda.device(torch.device("cuda", self.devices[i]))

    def _move(self, t: torch.Tensor, dev: torch.device) -> torch.Tensor:
        """t on `dev`: a peer copy (xGMI) where the two GPUs can reach each other, else staged through the host."""
        if not t.is_cuda or t.device == dev:
            return t.to(dev)
        if t.device.index in self._staged or dev.index in self._staged:
            return t.cpu().to(dev)             # (.cpu() synchronises the source stream; the upload is ordered on the destination's)
        return t.to(dev)

    def _put(self, i, x, normalize):
        """Rows x (CUDA tensor on any device, CPU tensor or numpy) appended to index i on ITS device and stream."""
        ix = self.indexes[i]
        with self._on(i):
            if isinstance(x, torch.Tensor) and x.is_cuda and x.device.index != self.devices[i]:
                x = self._move(x, torch.device("cuda", self.devices[i]))   # peer copy (xGMI), ordered by torch's streams
            ix.use_current_stream()
            ix.add(x, normalize=normalize)

    def add(self, x, normalize: bool = False):
        self._norms = None
        n = x.shape[0]
        if not self.shard:
            for i in range(len(self.indexes)):
                self._put(i, x, normalize)
            return
        if self._quota is None and len(self.indexes) > 1 and not self._warned_no_plan:
            import warnings
            warnings.warn("HipMultiIndex.add in shard mode without reserve(): no row plan exists, every row goes to the first GPU "
                          f"(cuda:{self.devices[0]}) and the other {len(self.indexes) - 1} search empty shards -- correct results, "
                          "nothing sharded.  Call reserve(total_rows) first (HbirdEvaluation does when the loader has a length or "
                          "memory_size is set).", RuntimeWarning, stacklevel=2)
            self._warned_no_plan = True
        lo = 0
        while lo < n:
            i = len(self.indexes) - 1
            if self._quota is None:
                i = 0
            else:
                for j, ix in enumerate(self.indexes[:-1]):
                    if ix.ntotal < self._quota:
                        i = j
                        break
            room = n - lo
            if self._quota is not None and i < len(self.indexes) - 1:
                room = min(room, self._quota - self.indexes[i].ntotal)
            self._put(i, x[lo:lo + room], normalize)
            lo += room

    def set_label_denominator(self, P: int):
        """As HipFlatIndex.set_label_denominator: the home device's label table holds int16 counts (P <= 32767)."""
        if self._nlab and int(P) != self._label_P:
            raise RuntimeError("set_label_denominator: the index already holds label rows")
        if not 0 <= int(P) <= 32767:
            raise ValueError("set_label_denominator: P must be in [0, 32767]")
        if (int(P) == 0) != (self._label_P == 0):
            self._labels = None         # fp32 values <-> int16 counts: a table kept by reset() has the other form's dtype
        self._label_P = int(P)

    def labels_to_fp32(self):
        """As HipFlatIndex.labels_to_fp32: int16 counts j -> fp32 values j / P (float32 division: the quotient K2 computed), denominator 0."""
        if self._label_P and self._labels is not None:
            self._labels = self._labels.to(torch.float32) / torch.tensor(float(self._label_P), device=self._labels.device)
        self._label_P = 0

    @property
    def label_denominator(self) -> int:
        return self._label_P

    def add_labels(self, lab):
        """label_memory rows (hbird_eval.py:329, 354) of the rows just added, in global row order, on the home device."""
        if not isinstance(lab, torch.Tensor):
            lab = torch.from_numpy(np.ascontiguousarray(lab, dtype=np.float32))
        lab = lab.detach().float()
        n, c = lab.shape
        self._c = int(c)
        need = self._nlab + n
        dt = torch.int16 if self._label_P else torch.float32
        if self._labels is None or self._labels.shape[0] < need or self._labels.shape[1] != c or self._labels.dtype != dt:
            planned = (self._quota * len(self.indexes)) if (self.shard and self._quota) else 0
            cap = max(need, planned, 0 if self._labels is None else self._labels.shape[0] * 3 // 2)
            grown = torch.empty((cap, c), dtype=dt, device=self.home)
            if self._nlab:
                grown[:self._nlab] = self._labels[:self._nlab]
            self._labels = grown
        lab = lab.to(self.home)
        if self._label_P:
            # (a 0-dim TENSOR divisor: torch turns a division by a Python scalar into a multiplication by its reciprocal on the GPU,
            # which rounds differently from the (float)j / (float)P that K2 and the library compute)
            Pt = torch.tensor(float(self._label_P), device=lab.device)
            cnt = torch.round(lab * Pt)
            if not bool(((cnt / Pt) == lab).all()):       # same exactness rule as the library (labels_to_counts_kernel)
                raise RuntimeError(f"label rows are not multiples of 1 / {self._label_P} (set_label_denominator): store them as fp32 instead")
            lab = cnt.to(torch.int16)
        self._labels[self._nlab:need] = lab
        self._nlab = need
        self._norms = None

    def _tables(self):
        """(labels [ntotal, C], norms [ntotal]) on the home device, handed to the aggregation handle."""
        if self._norms is None:
            n = self.ntotal
            with torch.cuda.device(self.home):
                if self.shard:
                    parts = []
                    for i, ix in enumerate(self.indexes):
                        with self._on(i):
                            ix.use_current_stream()
                            nr = ix.copy_norms()
                            torch.cuda.current_stream(nr.device).synchronize()
                        parts.append(self._move(nr, self.home))
                    self._norms = torch.cat(parts) if parts else torch.zeros(0, device=self.home)
                else:
                    with self._on(0):
                        self.indexes[0].use_current_stream()
                        self._norms = self.indexes[0].copy_norms()
                        torch.cuda.current_stream(self.home).synchronize()
                if self._labels is not None and self._nlab >= n and n > 0:
                    if self._label_P:
                        self.agg.set_label_count_table(self._labels[:n], self._norms, self._label_P, 0)
                    else:
                        self.agg.set_label_table(self._labels[:n], self._norms, 0)
        return (None if self._labels is None else self._labels[:self.ntotal]), self._norms

    def copy_norms(self) -> torch.Tensor:
        return self._tables()[1]

    # -- search -----------------------------------------------------------------------------------------------------
    def _threads(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.indexes), thread_name_prefix="hbird-gpu")
        return self._pool

    def _search(self, q, k: int, id_base: int, scores: bool):
        if isinstance(q, torch.Tensor) and q.is_cuda:
            q = q.detach().float().contiguous()
        else:
            q = torch.as_tensor(np.ascontiguousarray(q.detach().cpu().numpy() if isinstance(q, torch.Tensor) else q,
                                                     dtype=np.float32)).to(self.home)
        n = len(self.indexes)
        nq = q.shape[0]
        bases = self.shard_bases
        # q is complete before other devices' streams (and the workers' own streams on q's device) read it
        torch.cuda.current_stream(q.device).synchronize()
        # ... and so is every index: the workers run on THEIR threads' current streams (the default streams), while add() / peer
        # copies were enqueued on the CALLER's current stream of each device -- not the same when the bank was built under a
        # torch.cuda.stream(...) context
        for d in set(self.devices):
            torch.cuda.current_stream(torch.device("cuda", d)).synchronize()

        def run(i):
            index, dev = self.indexes[i], torch.device("cuda", self.devices[i])
            with torch.cuda.device(dev):
                if self.shard:
                    qi = q if dev == q.device else self._move(q, dev)
                else:
                    a, b = (nq * i) // n, (nq * (i + 1)) // n       # replicas: a slice of the queries each
                    qi = q[a:b] if dev == q.device else self._move(q[a:b], dev)
                index.use_current_stream()
                # shards always return ordering scores: the merge must see what the single index orders by
                idx, d = (index.search_scores if (self.shard or scores) else index.search)(qi.contiguous(), k, id_base + bases[i])
                fb = index.last_fp16_fallbacks()
                torch.cuda.current_stream(dev).synchronize()
                idx, d = self._move(idx, self.home), self._move(d, self.home)   # peer copy of the [nq, k] lists (xGMI)
                # the copy was enqueued on THIS thread's current streams (torch runs a peer copy on the source device's stream and
                # makes the destination's wait for it), the caller merges on its own: the lists must have landed before the
                # worker hands them over
                torch.cuda.current_stream(dev).synchronize()
                torch.cuda.current_stream(self.home).synchronize()
                return idx, d, fb

        parts = list(self._threads().map(run, range(n)))
        self._fallbacks = sum(p[2] for p in parts)
        with torch.cuda.device(self.home):
            if not self.shard:
                return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
            idx, sc = merge_topk(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]), 0)
            if scores:
                return idx, sc
            qh = q if q.device == self.home else q.to(self.home)
            self.agg.use_current_stream()
            return idx, self.agg.distances_from_scores(qh, sc.contiguous())

    def search(self, q, k: int, id_base: int = 0, out=None):
        host = not (isinstance(q, torch.Tensor) and q.is_cuda)
        idx, dist = self._search(q, k, id_base, False)
        if out is not None:
            out[0].copy_(idx); out[1].copy_(dist)
            return out
        return (idx.cpu().numpy(), dist.cpu().numpy()) if host else (idx, dist)

    def search_scores(self, q, k: int, id_base: int = 0, out=None):
        idx, sc = self._search(q, k, id_base, True)
        if out is not None:
            out[0].copy_(idx); out[1].copy_(sc)
            return out
        return idx, sc

    def distances_from_scores(self, q, scores):
        return self.agg.distances_from_scores(q, scores)

    def aggregate(self, q, idx, dist, beta: float = 0.02, id_base: int = 0):
        self._tables()
        with torch.cuda.device(self.home):
            self.agg.use_current_stream()
            return self.agg.aggregate(q.to(self.home), idx, dist, beta=beta, id_base=id_base)

    def search_aggregate(self, q, k: int, beta: float = 0.02, id_base: int = 0, want_neighbours: bool = False):
        """K4 on every GPU, merge on the home device, K5 there (hb_index_aggregate on the merged lists)."""
        host = not (isinstance(q, torch.Tensor) and q.is_cuda)
        if host:
            q = (q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32))).to(self.home)
        idx, dist = self._search(q, k, id_base, False)
        out = self.aggregate(q.float().contiguous(), idx, dist, beta=beta, id_base=id_base)
        if host:
            out, idx, dist = out.cpu().numpy(), idx.cpu().numpy(), dist.cpu().numpy()
        return (out, idx, dist) if want_neighbours else out

    # -- row access (return_knn_details, persistence) -------------------------------------------------------------------
    def reconstruct(self, ids, id_base: int = 0):
        host = not (isinstance(ids, torch.Tensor) and ids.is_cuda)
        ids_h = (torch.as_tensor(np.asarray(ids)) if host else ids).to(self.home).to(torch.int64).contiguous().view(-1)
        if not self.shard:
            with self._on(0):
                self.indexes[0].use_current_stream()
                out = self.indexes[0].reconstruct(ids_h, id_base)
        else:
            out = torch.zeros((ids_h.numel(), self.d), dtype=torch.float32, device=self.home)
            bases = self.shard_bases
            for i, ix in enumerate(self.indexes):
                lo = id_base + bases[i]
                own = (ids_h >= lo) & (ids_h < lo + ix.ntotal)
                if ix.ntotal == 0:
                    continue
                with self._on(i):
                    dev = torch.device("cuda", self.devices[i])
                    mine = self._move(torch.where(own, ids_h, torch.full_like(ids_h, -1)), dev)
                    ix.use_current_stream()
                    part = ix.reconstruct(mine, lo)                 # rows of other shards (id -1) come back as zeros
                    torch.cuda.current_stream(dev).synchronize()
                    part = self._move(part, self.home)
                    torch.cuda.current_stream(dev).synchronize()
                out += part
        return out.cpu().numpy() if host else out

    def gather_labels(self, ids):
        from hbird_mi import ops
        host = not (isinstance(ids, torch.Tensor) and ids.is_cuda)
        ids_h = (torch.as_tensor(np.asarray(ids)) if host else ids).to(self.home).to(torch.int64).contiguous().view(-1)
        with torch.cuda.device(self.home):
            if self._label_P:       # counts -> the fp32 values (float32 division: the very quotient K2 computed)
                ok = (ids_h >= 0) & (ids_h < self._nlab)
                out = self._labels[:self._nlab][ids_h.clamp(0, max(0, self._nlab - 1))].to(torch.float32) / torch.tensor(float(self._label_P), device=self.home)
                out[~ok] = 0.0
            else:
                out = ops.gather_rows(self._labels[:self._nlab], ids_h)
        return out.cpu().numpy() if host else out

    def set_label_table(self, labels, norms, id_base: int = 0):
        raise RuntimeError("HipMultiIndex keeps its own label table on the home device")


class NearestNeighborSearchHIP(NearestNeighborSearchBase):
    """Exact flat search on MI355X behind the reference's plugin interface.

    Keyword surface of search_faiss.py:7: `distance_measure` ("dot_product" | "l2" | "euclidean"),
    `idx_shard`, `use_fp16` (fp16 candidate pass + exact fp32 re-rank: same answers as fp32, several times
    faster), `gpu_ids`.
    Unknown keywords are swallowed like the reference's **kwargs.  Like the Faiss class it does not call
    the base constructor (search_faiss.py:7-32) and copies the bank to the GPU(s) at construction (78-81).

    GPUs.  In ONE process (the reference's call shape) the plugin drives every GPU of `gpu_ids` (default: all, as
    search_faiss.py:19-20) with one `hb_index_t` per entry and one host thread per index (faiss `threaded=True`, 57):
    `idx_shard=True` cuts the bank into contiguous row ranges with successive ids (faiss.IndexShards, 53-63) -- every
    GPU searches all queries on its range, the [nq, k] lists are copied to the first GPU (peer copy over xGMI) and
    merged there by hb_merge_topk on the ordering scores, which reproduces the single-index result bit for bit;
    `idx_shard=False` puts a full replica on every GPU and splits the queries (faiss.IndexReplicas, 65-74).
    Under torch.distributed (one process per GPU) each rank drives its own GPU only and `idx_shard=True` shards over
    the RANKS instead (RCCL all-gather of the packed lists + merge).
    """

    def __init__(self, feature_memory, n_neighbors=30, distance_measure="dot_product", idx_shard=False,
                 use_fp16=False, gpu_ids=None, **kwargs):
        self.n_neighbors = n_neighbors
        self.distance_measure = distance_measure.lower()
        self.idx_shard = idx_shard
        self.use_fp16 = use_fp16
        self.rerank_copy = int(kwargs.pop("rerank_copy", 0))     # use_fp16 only: the re-rank's row-major copy of the bank (0 automatic, 1 always, 2 never)
        self.embed_d = feature_memory.size(1)

        self.n_gpus = _lib.device_count()
        if self.n_gpus < 1:
            raise RuntimeError("No GPUs available for the HIP index.")            # search_faiss.py:15-16
        if gpu_ids is None:
            gpu_ids = list(range(self.n_gpus))
        else:
            for gpu_id in gpu_ids:                                                 # search_faiss.py:22-25
                if gpu_id >= self.n_gpus or gpu_id < 0:
                    raise ValueError(f"Invalid GPU ID: {gpu_id}. Available GPUs: 0-{self.n_gpus - 1}")
        self.gpu_ids = list(gpu_ids)

        self.rank, self.world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        cur = torch.cuda.current_device() if torch.cuda.is_available() else self.gpu_ids[0]
        if self.world > 1:
            # one process per GPU: this rank's GPU is the current torch device when it is listed, else the first listed
            self.local_gpus = [cur if cur in self.gpu_ids else self.gpu_ids[0]]
        else:
            self.local_gpus = list(self.gpu_ids)
        self.gpu = self.local_gpus[0]
        self.id_base = 0
        self.indexes = self._initialize_index()
        self.index = self.indexes[0]           # the first shard / replica (the only one with a single GPU)
        self._pool = None
        self._add_features_to_index(feature_memory)

    def _initialize_index(self):
        if self.distance_measure not in _METRICS:
            raise ValueError(f"Unsupported distance measure: {self.distance_measure}")   # search_faiss.py:48
        self.multi = None
        if len(self.local_gpus) > 1:
            self.multi = HipMultiIndex(self.embed_d, _METRICS[self.distance_measure], self.local_gpus, shard=bool(self.idx_shard))
            out = self.multi.indexes
        else:
            out = [HipFlatIndex(self.embed_d, _METRICS[self.distance_measure], self.local_gpus[0])]
        for index in out:
            index.set_fp16(2 if self.use_fp16 else 0)                                   # search_faiss.py:40; only where it pays
            index.set_rerank_copy(self.rerank_copy)
        return out

    def _add_features_to_index(self, feature_memory):
        M = feature_memory.size(0)
        lo, hi = 0, M
        if self.idx_shard and self.world > 1:
            # contiguous row ranges, successive ids (faiss.IndexShards.add, search_faiss.py:56-63)
            per = (M + self.world - 1) // self.world
            lo, hi = min(M, self.rank * per), min(M, (self.rank + 1) * per)
        self.id_base = lo
        target = self.multi if self.multi is not None else self.index
        target.reserve(max(hi - lo, 1))       # several local GPUs + idx_shard: equal contiguous ranges, successive ids
        target.add(feature_memory[lo:hi])
        self.shard_bases = [lo + b for b in self.multi.shard_bases] if self.multi is not None else [lo]

    def find_nearest_neighbors(self, q, k=None):
        if k is None:
            k = self.n_neighbors
        sharded = (self.idx_shard and self.world > 1) or self.multi is not None
        top = MAX_K if sharded else MAX_K_SEARCH
        if not 1 <= k <= top:
            # faiss-gpu raises for k > 2048 (its k-select limit); here one GPU takes the same 2048 (beyond 256 in ceil(k / 256) passes),
            # a sharded or multi-GPU index 256 (the merge of the shards' lists)
            raise ValueError(f"k={k} outside the supported range [1, {top}]" + (" of a sharded index (one GPU: up to 2048, faiss-gpu's own limit)" if sharded else " (faiss-gpu's own limit)"))
        if isinstance(q, torch.Tensor) and q.is_cuda:
            idx, dist = self._search_device(q, k)
            return idx, dist
        q_np = q.cpu().numpy() if isinstance(q, torch.Tensor) else np.asarray(q)   # search_faiss.py:88
        if (self.idx_shard and self.world > 1) or self.multi is not None:
            idx, dist = self._search_device(torch.from_numpy(np.ascontiguousarray(q_np, dtype=np.float32)).cuda(self.gpu), k)
            return idx.cpu().numpy(), dist.cpu().numpy()
        indices, distances = self.index.search(q_np, k, self.id_base)
        return indices, distances                                                  # (I, D) order: search_faiss.py:89-90

    def _search_local(self, q: torch.Tensor, k: int, id_base_unused: int = 0, scores: bool = False):
        """All local GPUs on q (a CUDA tensor) -> (idx, dist | ordering scores) on the first GPU."""
        if self.multi is not None:
            return (self.multi.search_scores if scores else self.multi.search)(q, k, self.id_base)
        self.index.use_current_stream()
        return (self.index.search_scores if scores else self.index.search)(q, k, self.id_base)

    def _search_device(self, q: torch.Tensor, k: int):
        if self.idx_shard and self.world > 1:
            from hbird_mi import dist as hdist
            self.index.use_current_stream()
            return hdist.sharded_search(lambda qq, kk, base: self._search_local(qq, kk, base, scores=True), merge_topk, q, k,
                                        self.id_base, _METRICS[self.distance_measure],
                                        finish=self.index.distances_from_scores, merge_packed=merge_topk_packed)
        return self._search_local(q, k)
