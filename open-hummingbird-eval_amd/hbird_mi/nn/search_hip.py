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

    def set_fp16(self, enable):
        """fp16 candidate pass + exact fp32 re-rank (GpuIndexFlatConfig.useFloat16 of the reference).  True / 1: always;
        2: only for banks of at least 131,072 rows, where it is faster than the fp32 kernel (same results either way)."""
        _lib.check(_lib.lib().hb_index_set_fp16(self._h, 2 if enable == 2 and enable is not True else int(bool(enable))))

    def last_fp16_fallbacks(self) -> int:
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().hb_index_last_fp16_fallbacks(self._h, ctypes.byref(n)))
        return int(n.value)

    def set_variant(self, variant: int):
        _lib.check(_lib.lib().hb_index_set_variant(self._h, int(variant)))

    def set_cluster(self, cluster_q: int = 0, cluster_b: int = 0, sync_lag: int = -1):
        """L2-sharing clusters of the work list (speed only, opt-in): 0 x 0 / 1 x 1 off, e.g. 2 x 2; sync_lag in stages."""
        _lib.check(_lib.lib().hb_index_set_cluster(self._h, int(cluster_q), int(cluster_b), int(sync_lag)))

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
        out = []
        for g in self.local_gpus:
            index = HipFlatIndex(self.embed_d, _METRICS[self.distance_measure], g)
            index.set_fp16(2 if self.use_fp16 else 0)                                   # search_faiss.py:40; only where it pays
            out.append(index)
        return out

    def _add_features_to_index(self, feature_memory):
        M = feature_memory.size(0)
        lo, hi = 0, M
        if self.idx_shard and self.world > 1:
            # contiguous row ranges, successive ids (faiss.IndexShards.add, search_faiss.py:56-63)
            per = (M + self.world - 1) // self.world
            lo, hi = min(M, self.rank * per), min(M, (self.rank + 1) * per)
        self.id_base = lo
        n = len(self.indexes)
        self.shard_bases = []
        for i, index in enumerate(self.indexes):
            if self.idx_shard and n > 1:
                per = (hi - lo + n - 1) // n
                a, b = min(hi, lo + i * per), min(hi, lo + (i + 1) * per)
            else:
                a, b = lo, hi                                   # replicas (or a single index): all local rows
            self.shard_bases.append(a)
            index.reserve(max(b - a, 1))
            index.add(feature_memory[a:b])

    def _threads(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.indexes), thread_name_prefix="hbird-gpu")
        return self._pool

    def find_nearest_neighbors(self, q, k=None):
        if k is None:
            k = self.n_neighbors
        if not 1 <= k <= MAX_K:
            raise ValueError(f"k={k} outside the supported range [1, {MAX_K}]")
        if isinstance(q, torch.Tensor) and q.is_cuda:
            idx, dist = self._search_device(q, k)
            return idx, dist
        q_np = q.cpu().numpy() if isinstance(q, torch.Tensor) else np.asarray(q)   # search_faiss.py:88
        if (self.idx_shard and self.world > 1) or len(self.indexes) > 1:
            idx, dist = self._search_device(torch.from_numpy(np.ascontiguousarray(q_np, dtype=np.float32)).cuda(self.gpu), k)
            return idx.cpu().numpy(), dist.cpu().numpy()
        indices, distances = self.index.search(q_np, k, self.id_base)
        return indices, distances                                                  # (I, D) order: search_faiss.py:89-90

    def _search_local(self, q: torch.Tensor, k: int, id_base_unused: int = 0, scores: bool = False):
        """All local GPUs on q (a CUDA tensor on the first one) -> (idx, dist | ordering scores) on the first GPU."""
        n = len(self.indexes)
        if n == 1:
            self.index.use_current_stream()
            return (self.index.search_scores if scores else self.index.search)(q, k, self.shard_bases[0])
        dev0 = torch.device("cuda", self.local_gpus[0])
        torch.cuda.current_stream(dev0).synchronize()            # q is complete before other devices' streams read it
        nq = q.shape[0]

        def run(i):
            index, g = self.indexes[i], self.local_gpus[i]
            dev = torch.device("cuda", g)
            with torch.cuda.device(dev):
                if self.idx_shard:
                    qi = q if dev == q.device else q.to(dev)
                else:
                    a, b = (nq * i) // n, (nq * (i + 1)) // n       # replicas: a slice of the queries each
                    qi = q[a:b] if dev == q.device else q[a:b].to(dev)
                index.use_current_stream()
                # shards always return ordering scores: the merge must see what the single index orders by
                idx, d = (index.search_scores if (self.idx_shard or scores) else index.search)(qi.contiguous(), k, self.shard_bases[i])
                torch.cuda.current_stream(dev).synchronize()
                return idx.to(dev0), d.to(dev0)                    # peer copy of the [nq, k] lists (xGMI)

        parts = list(self._threads().map(run, range(n)))
        with torch.cuda.device(dev0):
            if not self.idx_shard:
                return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
            idx, sc = merge_topk(torch.stack([p[1] for p in parts]), torch.stack([p[0] for p in parts]), 0)
            if scores:
                return idx, sc
            self.index.use_current_stream()
            return idx, self.index.distances_from_scores(q if q.device == dev0 else q.to(dev0), sc.contiguous())

    def _search_device(self, q: torch.Tensor, k: int):
        if self.idx_shard and self.world > 1:
            from hbird_mi import dist as hdist
            self.index.use_current_stream()
            return hdist.sharded_search(lambda qq, kk, base: self._search_local(qq, kk, base, scores=True), merge_topk, q, k,
                                        self.id_base, _METRICS[self.distance_measure],
                                        finish=self.index.distances_from_scores, merge_packed=merge_topk_packed)
        return self._search_local(q, k)
