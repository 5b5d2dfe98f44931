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
MAX_K = 256


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

    def search_scores(self, q, k: int, id_base: int = 0):
        """Like `search`, but the second output holds the ORDERING scores (larger is better; L2: q.b - |b|^2/2), the
        input of a cross-shard merge (hb_index_set_score_output); convert with `distances_from_scores`."""
        lib = _lib.lib()
        _lib.check(lib.hb_index_set_score_output(self._h, 1))
        try:
            return self.search(q, k, id_base)
        finally:
            _lib.check(lib.hb_index_set_score_output(self._h, 0))

    def distances_from_scores(self, q: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        """Merged ordering scores [nq,k] (CUDA) -> the metric's distances, in place (no-op for the inner product)."""
        assert q.is_cuda and scores.is_cuda and scores.is_contiguous() and scores.dtype == torch.float32
        q = q.contiguous().float()
        _lib.check(_lib.lib().hb_index_distances_from_scores(self._h, _ptr(q), q.shape[0], scores.shape[1], _ptr(scores)))
        return scores

    def search(self, q, k: int, id_base: int = 0):
        """-> (idx int64 [nq,k], dist float32 [nq,k]); torch CUDA tensors for CUDA queries, numpy otherwise."""
        on_dev, q = self._as_f32(q)
        nq = q.shape[0]
        if on_dev:
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

    def set_fp16(self, enable: bool):
        """fp16 candidate pass + exact fp32 re-rank (GpuIndexFlatConfig.useFloat16 of the reference)."""
        _lib.check(_lib.lib().hb_index_set_fp16(self._h, int(bool(enable))))

    def last_fp16_fallbacks(self) -> int:
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().hb_index_last_fp16_fallbacks(self._h, ctypes.byref(n)))
        return int(n.value)

    def set_variant(self, variant: int):
        _lib.check(_lib.lib().hb_index_set_variant(self._h, int(variant)))

    def schedule_info(self) -> dict:
        out = (ctypes.c_int64 * 8)()
        _lib.check(_lib.lib().hb_index_schedule_info(self._h, out))
        keys = ["workgroups", "segments", "slots", "panel_tiles", "max_slots_per_qtile", "query_tiles", "bank_tiles"]
        return dict(zip(keys, list(out)[:7]))


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


class NearestNeighborSearchHIP(NearestNeighborSearchBase):
    """Exact flat search on MI355X behind the reference's plugin interface.

    Keyword surface of search_faiss.py:7: `distance_measure` ("dot_product" | "l2" | "euclidean"),
    `idx_shard`, `use_fp16` (fp16 candidate pass + exact fp32 re-rank: same answers as fp32, several times
    faster), `gpu_ids`.
    Unknown keywords are swallowed like the reference's **kwargs.  Like the Faiss class it does not call
    the base constructor (search_faiss.py:7-32) and copies the bank to the GPU at construction (78-81).
    """

    def __init__(self, feature_memory, n_neighbors=30, distance_measure="dot_product", idx_shard=False,
                 use_fp16=False, gpu_ids=None, **kwargs):
        self.n_neighbors = n_neighbors
        self.distance_measure = distance_measure.lower()
        self.idx_shard = idx_shard
        self.use_fp16 = use_fp16
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
        self.gpu_ids = gpu_ids

        # one process per GPU: this rank's GPU is the current torch device when it is listed, else the
        # first listed id
        cur = torch.cuda.current_device() if torch.cuda.is_available() else gpu_ids[0]
        self.gpu = cur if cur in gpu_ids else gpu_ids[0]
        self.rank, self.world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        self.id_base = 0
        self.index = self._initialize_index()
        self._add_features_to_index(feature_memory)

    def _initialize_index(self):
        if self.distance_measure not in _METRICS:
            raise ValueError(f"Unsupported distance measure: {self.distance_measure}")   # search_faiss.py:48
        index = HipFlatIndex(self.embed_d, _METRICS[self.distance_measure], self.gpu)
        index.set_fp16(bool(self.use_fp16))                                             # search_faiss.py:40
        return index

    def _add_features_to_index(self, feature_memory):
        M = feature_memory.size(0)
        lo, hi = 0, M
        if self.idx_shard and self.world > 1:
            # contiguous row ranges, successive ids (faiss.IndexShards.add, search_faiss.py:56-63)
            per = (M + self.world - 1) // self.world
            lo, hi = min(M, self.rank * per), min(M, (self.rank + 1) * per)
        self.id_base = lo
        self.index.reserve(max(hi - lo, 1))
        self.index.add(feature_memory[lo:hi])

    def find_nearest_neighbors(self, q, k=None):
        if k is None:
            k = self.n_neighbors
        if not 1 <= k <= MAX_K:
            raise ValueError(f"k={k} outside the supported range [1, {MAX_K}]")
        if isinstance(q, torch.Tensor) and q.is_cuda:
            idx, dist = self._search_device(q, k)
            return idx, dist
        q_np = q.cpu().numpy() if isinstance(q, torch.Tensor) else np.asarray(q)   # search_faiss.py:88
        if self.idx_shard and self.world > 1:
            idx, dist = self._search_device(torch.from_numpy(np.ascontiguousarray(q_np)).cuda(self.gpu), k)
            return idx.cpu().numpy(), dist.cpu().numpy()
        indices, distances = self.index.search(q_np, k, self.id_base)
        return indices, distances                                                  # (I, D) order: search_faiss.py:89-90

    def _search_device(self, q: torch.Tensor, k: int):
        self.index.use_current_stream()
        if self.idx_shard and self.world > 1:
            from hbird_mi import dist as hdist
            return hdist.sharded_search(self.index.search_scores, merge_topk, q, k, self.id_base,
                                        _METRICS[self.distance_measure], finish=self.index.distances_from_scores)
        return self.index.search(q, k, self.id_base)
