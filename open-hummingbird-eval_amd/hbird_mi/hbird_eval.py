"""Hummingbird dense-retrieval evaluation on MI355X: the counterpart of the reference's
hbird/hbird_eval.py (class HbirdEvaluation 54-637, hbird_evaluation 640-722), same public names,
arguments, return values and error behaviour -- with the whole hot path device-resident:

  bank build   ViT tokens -> fused L2-normalise + fragment-tiled append (K1), patch soft labels from the
               mask histogram (K2), optional bounded-memory sampling (K3); nothing crosses to the host
               except the uniform noise the reference draws from torch's CPU generator (hbird_eval.py:500).
  query path   tokens -> exact brute-force kNN (K4, fused fp32-MFMA top-k) -> cosine-softmax label
               aggregation (K5) -> bilinear upsample + argmax (K6) -> confusion matrix (K7); only the
               [C,C] confusion matrix reaches the host, where the Hungarian mIoU tail runs.

With torch.distributed initialised (one process per GPU, backend "nccl" = RCCL) the bank is row-sharded
over the ranks in the reference's row order (contiguous ranges of training batches, successive ids), the
validation batches are dealt round-robin to the ranks, and every search is: all-gather(queries) -> local
shard search -> all-gather(top-k) -> merge -> aggregation against the all-gathered label table.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, List, Optional, Tuple

import torch

from hbird_mi import dist as hdist
from hbird_mi import ops
from hbird_mi import tiling
from hbird_mi.models import FeatureExtractor, FeatureExtractorSimple
from hbird_mi.nn.search_hip import HipFlatIndex, HipMultiIndex, MAX_K, merge_topk, merge_topk_packed, _METRICS
from hbird_mi.utils.eval_metrics import PredsmIoU

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(iterator, *args, **kwargs):
        return iterator

logger = logging.getLogger(__name__)
if not logger.handlers:
    _handler = logging.StreamHandler()
    _handler.setFormatter(logging.Formatter(fmt="%(asctime)s | %(levelname)s | %(name)s: %(message)s",
                                            datefmt="%H:%M:%S"))
    logger.addHandler(_handler)
    logger.setLevel(getattr(logging, os.environ.get("HBIRD_LOG_LEVEL", "WARNING").upper(), logging.WARNING))

_NN_METHODS = ("hip", "faiss", "scann")
# what the last hbird_evaluation() call did about its bank (the CLI prints it): bank_loaded, bank_build_s, bank_rows, train_batches_loaded
last_run_info: Dict[str, Any] = {}


class HbirdEvaluation:
    """Same constructor as the reference (hbird_eval.py:97-111).

    `nn_method`: "hip" (this engine), or the reference's names "faiss" (exact flat search: identical
    semantics, served by the HIP engine) and "scann" (the reference's approximate CPU backend is not
    re-implemented; the exact HIP search is used and a warning is logged).  `nn_params` keeps the Faiss
    keywords: distance_measure, idx_shard, use_fp16, gpu_ids (search_faiss.py:7); ScaNN-only keywords are
    accepted and ignored.

    `reuse_memory` (trailing keyword, not in the reference, whose constructor always rebuilds and overwrites the files,
    hbird_eval.py:165-172): when both `f_mem_p` and `l_mem_p` name existing files the bank is LOADED from them
    (`load_memory`) and the training loader is never touched; otherwise the bank is built and saved there as in the reference.
    `bank_loaded` / `bank_build_s` say which happened and how long it took.
    """

    def __init__(self, feature_extractor: torch.nn.Module, train_loader, num_classes: int, n_neighbours: int = 30,
                 augmentation_epoch: int = 1, device: torch.device | str = "cpu", nn_method: str = "scann",
                 nn_params: Optional[Dict[str, Any]] = None, memory_size: Optional[int] = None,
                 dataset_size: Optional[int] = None, f_mem_p: Optional[str] = None,
                 l_mem_p: Optional[str] = None, reuse_memory: bool = False) -> None:
        if nn_params is None:
            nn_params = {}
        self.nn_params = nn_params
        self.device = device
        self.nn_method = nn_method
        assert self.nn_method in _NN_METHODS, "Only hip, faiss and scann are supported"   # hbird_eval.py:121
        if self.nn_method == "scann":
            logger.warning("nn_method='scann': the approximate ScaNN CPU backend is replaced by the exact HIP search")
        if not torch.cuda.is_available():
            raise RuntimeError("HbirdEvaluation needs an MI355X: the HIP path has no CPU fallback")
        dev = torch.device(device)
        self.gpu = dev.index if dev.type == "cuda" and dev.index is not None else torch.cuda.current_device()
        self.gpu_device = torch.device("cuda", self.gpu)
        self.feature_extractor = feature_extractor.to(device)
        self.feature_extractor.eval()

        self.augmentation_epoch = augmentation_epoch
        self.memory_size = memory_size
        self.n_neighbours = n_neighbours
        self.num_classes = num_classes
        self.f_mem_p = f_mem_p
        self.l_mem_p = l_mem_p
        self.num_sampled_features: Optional[int] = None
        self.rank, self.world = hdist.rank_world()

        distance = str(nn_params.get("distance_measure", "dot_product")).lower()
        if distance not in _METRICS:
            raise ValueError(f"Unsupported distance measure: {distance}")                 # search_faiss.py:48
        self.metric = _METRICS[distance]
        gpu_ids = nn_params.get("gpu_ids")
        n_dev = torch.cuda.device_count()
        if gpu_ids is not None:
            for g in gpu_ids:
                if g >= n_dev or g < 0:
                    raise ValueError(f"Invalid GPU ID: {g}. Available GPUs: 0-{n_dev - 1}")    # search_faiss.py:25
        # GPUs of THIS process.  One process per GPU under torch.distributed; otherwise the reference's call shape
        # (hbird_eval.py:267-281 -> search_faiss.py:19-20, 50-76): every listed GPU, ALL of them by default, as row shards
        # (idx_shard=True) or replicas -- with the extractor's GPU first, the home of labels, merge and aggregation.
        if self.world > 1:
            self.local_gpus = [self.gpu]
        else:
            ids = [int(g) for g in gpu_ids] if gpu_ids is not None else list(range(n_dev))
            if self.gpu in ids:
                ids.remove(self.gpu)
                ids.insert(0, self.gpu)
            self.local_gpus = ids or [self.gpu]
            if self.local_gpus[0] != self.gpu:
                self.gpu = self.local_gpus[0]
                self.gpu_device = torch.device("cuda", self.gpu)
        if not 1 <= n_neighbours <= MAX_K:
            raise ValueError(f"n_neighbours={n_neighbours} outside the supported range [1, {MAX_K}]")
        # idx_shard keeps the reference's default (False, search_faiss.py:7): with several ranks that means a full bank
        # replica per rank and data-parallel validation batches (faiss.IndexReplicas, 65-74); idx_shard=True row-shards
        # the bank over the ranks (faiss.IndexShards, 53-63) -- the mode for banks that should not be built N times
        self.sharded = self.world > 1 and bool(nn_params.get("idx_shard", False))
        # label_shard (with idx_shard under torch.distributed; not in the reference): label_memory stays sharded with the bank instead of
        # being replicated on every rank (6.2 GB at BASELINE cfg-3, 16.7 GB for the full ADE20K bank); only the bank-row norms are
        # replicated (4 B per row) and every search ends in one more all-reduce of [queries, classes] partial sums
        self.label_shard = self.sharded and bool(nn_params.get("label_shard", False))
        self.compress_labels = bool(nn_params.get("compress_labels", True))
        if self.world > 1:
            logger.warning("torch.distributed world of %d ranks: bank %s (nn_params['idx_shard']=%s)", self.world,
                           "row-sharded over the ranks" if self.sharded else "replicated on every rank", self.sharded)
            if self.sharded or self.memory_size is not None:
                # sharded: every rank replays the ONE random stream of the reference; replicas with a bounded memory:
                # every rank samples the SAME patches (hbird_eval.py:500), so that all replicas are the same bank and the
                # all-reduced mIoU is that of a single process, whatever the ranks' own seeds were
                self._align_cpu_rng()

        eval_spatial_resolution = self.feature_extractor.eval_spatial_resolution
        logger.info("Initializing memory: nn_method=%s, memory_size=%s, augmentation_epoch=%s", self.nn_method,
                    str(self.memory_size), self.augmentation_epoch)
        if self.memory_size is not None:
            if dataset_size is None:
                raise ValueError("dataset_size must be provided when memory_size is set.")  # hbird_eval.py:144-145
            denom = dataset_size * self.augmentation_epoch
            self.num_sampled_features = max(1, self.memory_size // max(1, denom))           # 146-147
            # rows the build will fill (the reference preallocates memory_size and trims, 156-172): what shards split
            self._planned_rows = min(self.memory_size, self.num_sampled_features * max(1, denom))

        if len(self.local_gpus) > 1:
            self.index = HipMultiIndex(self.feature_extractor.d_model, self.metric, self.local_gpus,
                                       shard=bool(nn_params.get("idx_shard", False)))
            logger.info("one process, %d GPUs %s: bank %s", len(self.local_gpus), self.local_gpus,
                        "row-sharded (faiss.IndexShards)" if self.index.shard else "replicated (faiss.IndexReplicas)")
        else:
            self.index = HipFlatIndex(self.feature_extractor.d_model, self.metric, self.gpu)
        self.index.set_num_classes(num_classes)
        if nn_params.get("use_fp16", False):          # search_faiss.py:40; certified-exact fast mode, only where it pays
            self.index.set_fp16(2)                    # (rows x queries x D >= 1.5e10 (k' / 64)^2), like the plugin: never slower than fp32
            self.index.set_rerank_copy(int(nn_params.get("rerank_copy", 0)))   # 0 automatic (when memory allows), 1 always, 2 never
        self.id_base = 0            # global id of this rank's first bank row
        self.total_rows = 0         # bank rows over all ranks
        self._label_table = None    # all-gathered labels / norms in sharded mode
        self.bank_loaded = False
        import time as _time
        t0 = _time.perf_counter()
        with torch.cuda.device(self.gpu_device):
            if reuse_memory and self._reusable_memory():
                # the saved bank of an earlier run: every rank takes its row range of the one file pair
                self.batches_loaded = 0
                self.bank_loaded = self.load_memory()
            if not self.bank_loaded:
                filled = self._create_memory(train_loader, num_classes=self.num_classes,
                                             eval_spatial_resolution=eval_spatial_resolution)
                logger.info("Memory rows: %s", filled)
                self._save_memory()
                self._finalize_shards()
            torch.cuda.synchronize(self.gpu_device)
        self.bank_build_s = _time.perf_counter() - t0

    @classmethod
    def from_index(cls, feature_extractor: torch.nn.Module, index, num_classes: int, n_neighbours: int = 30,
                   device: torch.device | str = "cuda") -> "HbirdEvaluation":
        """An evaluator over a bank that is already resident in a `HipFlatIndex` (rows + label rows added by the caller): `evaluate` runs
        as usual, nothing is built.  Not in the reference (its bank only ever comes from `_create_memory`); used by `bench.py --e2e` to time
        the evaluation loop against the synthetic 10 M-row bank, and handy for banks produced elsewhere."""
        self = cls.__new__(cls)
        self.nn_params, self.device, self.nn_method = {}, device, "hip"
        dev = torch.device(device)
        self.gpu = dev.index if dev.type == "cuda" and dev.index is not None else torch.cuda.current_device()
        self.gpu_device = torch.device("cuda", self.gpu)
        self.feature_extractor = feature_extractor.to(device).eval()
        self.augmentation_epoch, self.memory_size, self.n_neighbours, self.num_classes = 1, None, n_neighbours, num_classes
        self.f_mem_p = self.l_mem_p = None
        self.num_sampled_features = None
        self.rank, self.world = 0, 1
        self.metric = index.metric
        self.local_gpus = [self.gpu]
        self.sharded = self.label_shard = False
        self.compress_labels = True
        self.index = index
        self.index.set_num_classes(num_classes)
        self.id_base, self.total_rows, self._label_table = 0, index.ntotal, None
        self.bank_loaded, self.bank_build_s, self.batches_loaded = True, 0.0, 0
        return self

    def _align_cpu_rng(self) -> None:
        """Sharded bank build: every rank replays the reference's single CPU random stream (the sampling noise of
        hbird_eval.py:500 is drawn for EVERY batch, owned or not), so all ranks must start from rank 0's state."""
        st = torch.get_rng_state().to(self.gpu_device)
        torch.distributed.broadcast(st, 0)
        torch.set_rng_state(st.cpu())

    # ------------------------------------------------------------------------------------------------
    # bank build (reference _create_memory, hbird_eval.py:283-369)
    # ------------------------------------------------------------------------------------------------
    def _create_memory(self, train_loader, num_classes: int, eval_spatial_resolution: int) -> Optional[int]:
        S = eval_spatial_resolution
        try:
            n_batches = len(train_loader)
        except TypeError:
            n_batches = None
        total_flat = None if n_batches is None else n_batches * self.augmentation_epoch
        if self.sharded:
            if total_flat is None:
                raise ValueError("a sharded bank build needs a train_loader with a length")
            own_lo, own_hi = hdist.shard_range(total_flat, self.rank, self.world)
        else:
            own_lo, own_hi = 0, float("inf")
        if self.memory_size is not None:
            self.index.reserve(max(1, self._planned_rows // max(1, self.world if self.sharded else 1)))
        state = {"presized": self.memory_size is not None}
        # Which batches this rank must actually LOAD.  Unbounded sharded build: only its own (hdist.rank_batches fetches just those
        # index batches from a DataLoader: rank r of N decodes 1 / N of the images).  Bounded memory: every batch, because the
        # reference's one random stream (500) is consumed for every batch by an amount that depends on its masks -- all ranks
        # replay it -- and the masks only come with their images.
        need_all = self.memory_size is not None or not self.sharded
        self.batches_loaded = 0
        self._mixed_patch_sizes = False
        failure = None
        with torch.no_grad():
            for ep in tqdm(range(self.augmentation_epoch), desc="Augmentation loop"):
                base = ep * (n_batches or 0)
                batches = enumerate(train_loader) if need_all else hdist.rank_batches(train_loader, lambda i, b=base: own_lo <= b + i < own_hi)
                try:
                    self._create_memory_epoch(batches, base, n_batches, own_lo, own_hi, total_flat, num_classes, S, state)
                except Exception as e:          # noqa: BLE001 -- re-raised below, on every rank
                    failure = e
                # Sharded build: a rank that raised must not leave its peers waiting in the next collective (_finalize_shards, or the
                # bounded build's aligned random stream): one all-reduce per epoch end, and all ranks go on or none
                if self.sharded:
                    flag = torch.tensor([0 if failure is None else 1], device=self.gpu_device)
                    torch.distributed.all_reduce(flag)
                    if int(flag.item()) != 0:
                        if failure is not None:
                            raise failure
                        raise RuntimeError(f"the bank build failed on another rank (epoch {ep}); this rank stops with it")
                elif failure is not None:
                    raise failure
        self.id_base = 0          # a sharded build learns its id base from the ranks' row counts (_finalize_shards)
        return self.index.ntotal

    def _create_memory_epoch(self, batches, base, n_batches, own_lo, own_hi, total_flat, num_classes, S, state) -> None:
        """One pass over the training batches (the body of the reference's inner loop, hbird_eval.py:306-355)."""
        presized = state["presized"]
        for bi, (x, y) in tqdm(batches, desc="Memory Creation loop"):
            flat = base + bi if n_batches is not None else self.batches_loaded
            mine = own_lo <= flat < own_hi
            self.batches_loaded += 1
            y = y.to(self.gpu_device)
            y = (y * 255).long()                                   # hbird_eval.py:309
            bs = y.shape[0]
            input_size = x.shape[-1]
            patch_size = input_size // S                          # 313-314
            if self.index.ntotal == 0 and not self._mixed_patch_sizes:
                self._set_label_denominator(patch_size * patch_size)
            elif self.index.label_denominator not in (0, patch_size * patch_size):
                # the reference recomputes patch_size per batch (313-314) and its fp32 label rows do not care; the compressed table
                # holds counts of ONE denominator: on a second one the rows stored so far go back to fp32 (one pass) and the build
                # goes on as the reference's does
                logger.warning("training batches of a second input size (patch %d x %d after label denominator %d): the label "
                               "table continues as fp32 rows", patch_size, patch_size, self.index.label_denominator)
                self.index.labels_to_fp32()
                self._mixed_patch_sizes = True
            # K2: `y[y == 255] = 0` (310) + patchify (317) + one-hot mean (319-320)
            label = ops.patch_label_hist(y, patch_size, num_classes, map255=True)   # [bs,S,S,C]
            if self.memory_size is None:
                if not presized and total_flat is not None:
                    # the unbounded bank's size is known up front (batches x images x patches; a short last
                    # batch only over-reserves): one allocation instead of geometric growth copies
                    own_batches = (min(own_hi, total_flat) - own_lo) if self.sharded else total_flat
                    self.index.reserve(max(1, int(own_batches) * bs * S * S))
                    presized = True
                feats = self._tokens(x)                            # [bs, S*S, D] on the GPU
                self.index.use_current_stream()
                self.index.add(feats.reshape(-1, feats.shape[-1]), normalize=True)   # K1 (324-329)
                self.index.add_labels(label.reshape(-1, num_classes))
            else:
                K = int(self.num_sampled_features)
                lab = label.reshape(bs, S * S, num_classes)
                scores, nonempty, nz = ops.patch_scores(lab)       # K3a (471-493)
                nz_host = nz.cpu().tolist()                        # 497 (.tolist() sync in the reference too)
                total_nz = sum(nz_host)
                # the CPU generator is consumed for EVERY batch, in loader order, so that all ranks
                # stay aligned with the single-process stream of the reference (500)
                r = torch.rand(total_nz) if total_nz > 0 else torch.zeros(0)
                if not mine:
                    continue
                r_off = torch.tensor([0] + nz_host[:-1], dtype=torch.int64).cumsum(0)
                sidx = ops.patch_select(scores, nonempty, r.to(self.gpu_device), r_off.to(self.gpu_device), K)
                feats = self._tokens(x)
                D = feats.shape[-1]
                rows = (sidx + torch.arange(bs, device=sidx.device)[:, None] * (S * S)).reshape(-1)
                sampled = ops.gather_rows(feats.reshape(-1, D), rows)                  # 515
                self.index.use_current_stream()
                self.index.add(sampled, normalize=True)                                # 335, 352-353
                self.index.add_labels(ops.gather_rows(lab.reshape(-1, num_classes), rows))   # 344-354
        state["presized"] = presized

    def _set_label_denominator(self, P: int) -> None:
        """Every soft label is j / P, P = patch_size ** 2 (the mean of a one-hot over a patch's P pixels, hbird_eval.py:319-320): the
        index keeps the uint16 count j instead of the fp32 value -- half the label table (6.2 -> 3.1 GB at cfg-3, per rank when it is
        replicated) and half K5's gather traffic, the same fp32 values back.  nn_params["compress_labels"] = False keeps fp32."""
        if self.compress_labels and 0 < int(P) <= 32767 and self.index.ntotal == 0 and self.index.label_denominator != int(P):
            self.index.set_label_denominator(int(P))

    @staticmethod
    def _infer_label_denominator(lm: torch.Tensor) -> int:
        """P of a stored label_memory (rows of j / P summing to 1), or 0: the smallest positive value of a sample suggests 1 / P, the whole
        tensor is then checked exactly (chunked) -- a table that is not of that form stays fp32."""
        if lm.numel() == 0:
            return 0
        head = lm[:4096].float()
        pos = head[head > 0]
        if pos.numel() == 0:
            return 0
        P = int(round(1.0 / float(pos.min())))
        if not 0 < P <= 32767:
            return 0
        for lo in range(0, lm.shape[0], 1 << 18):
            v = lm[lo:lo + (1 << 18)].float()
            Pt = torch.tensor(float(P))
            c = torch.round(v * Pt)
            if not bool(((c / Pt) == v).all()) or bool((c < 0).any()) or bool((c > P).any()):
                return 0
        return P

    def _tokens(self, x: torch.Tensor) -> torch.Tensor:
        feats, _ = self.feature_extractor.forward_features(x.to(self.device))    # (BS, N, D)
        return feats.to(self.gpu_device, dtype=torch.float32).contiguous()

    def _finalize_shards(self) -> None:
        """Sharded mode: learn every rank's row range and replicate the (small) label / norm tables."""
        n_local = self.index.ntotal
        if not self.sharded:
            self.total_rows = n_local
            return
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.gpu_device)
        counts[self.rank] = n_local
        torch.distributed.all_reduce(counts)
        counts = counts.cpu().tolist()
        self.id_base = sum(counts[:self.rank])
        self.total_rows = sum(counts)
        self.index.use_current_stream()
        nrm_all, _ = hdist.allgather_rows(self.index.copy_norms())
        norms = torch.cat([nrm_all[r, :counts[r]] for r in range(self.world)]).contiguous()
        if self.label_shard:
            self._label_table = (None, norms)          # label rows stay with their owners
            return
        # every rank agrees on the storage form (an empty shard never saw a batch: it takes the others' denominator)
        # (ranks WITH rows that disagree -- one on fp32, one on counts, or two denominators -- fall back to fp32 rows together: a MAX alone
        # would send the fp32 rank into copy_label_counts, which fails there while its peers wait in the all-gather)
        own_P = self.index.label_denominator
        pden = torch.tensor([own_P, -own_P if n_local else -(1 << 40)], dtype=torch.int64, device=self.gpu_device)
        torch.distributed.all_reduce(pden, op=torch.distributed.ReduceOp.MAX)
        p_max, p_min = int(pden[0].item()), -int(pden[1].item())
        P = p_max if (p_min == p_max or p_min == (1 << 40)) else 0
        if P == 0 and p_max > 0:
            logger.warning("ranks disagree on the label denominator (%d .. %d): the replicated label table stays fp32", p_min, p_max)
        if P > 0:      # the replicated table travels and stays as uint16 counts: half the bytes on the wire and per rank
            if n_local == 0 and self.index.label_denominator != P:
                self.index.set_label_denominator(P)
            cnt_local = (self.index.copy_label_counts() if n_local
                         else torch.zeros((0, self.num_classes), dtype=torch.int16, device=self.gpu_device))
            cnt_all, _ = hdist.allgather_rows(cnt_local.contiguous().view(torch.uint8))      # bytes: gloo has no int16 collectives
            counts_tab = torch.cat([cnt_all[r, :counts[r]] for r in range(self.world)]).contiguous().view(torch.int16)
            self._label_table = (counts_tab, norms)
            self._label_table_P = P
            self.index.set_label_count_table(counts_tab, norms, P, 0)
            return
        lab_local = (self.index.gather_labels(torch.arange(n_local, device=self.gpu_device)) if n_local
                     else torch.zeros((0, self.num_classes), device=self.gpu_device))
        lab_all, _ = hdist.allgather_rows(lab_local)
        labels = torch.cat([lab_all[r, :counts[r]] for r in range(self.world)]).contiguous()
        self._label_table = (labels, norms)
        self._label_table_P = 0
        self.index.set_label_table(labels, norms, 0)

    # ------------------------------------------------------------------------------------------------
    # bank persistence (reference 371-400): same two-tensor torch.save format
    # ------------------------------------------------------------------------------------------------
    @property
    def feature_memory(self) -> torch.Tensor:
        """This rank's bank rows as a CPU tensor [M_local, D] (the reference keeps it on the CPU, 178)."""
        n = self.index.ntotal
        if n == 0:
            return torch.zeros((0, self.index.d))
        ids = torch.arange(n, device=self.gpu_device)
        return self.index.reconstruct(ids).cpu()

    @property
    def label_memory(self) -> torch.Tensor:
        n = self.index.ntotal
        if n == 0:
            return torch.zeros((0, self.num_classes))
        return self.index.gather_labels(torch.arange(n, device=self.gpu_device)).cpu()

    def _gather_to_rank0(self, local: torch.Tensor) -> Optional[torch.Tensor]:
        """Sharded mode: the shards' rows [M_r, W] (CPU) concatenated in rank order on rank 0 (None elsewhere).  Rows
        travel GPU to GPU in chunks (send / recv work on both RCCL and gloo), so no rank stages more than a chunk."""
        td = torch.distributed
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.gpu_device)
        counts[self.rank] = local.shape[0]
        td.all_reduce(counts)
        counts = counts.cpu().tolist()
        width = local.shape[1]
        chunk = max(1, (256 << 20) // (4 * max(1, width)))
        if self.rank == 0:
            out = torch.empty((sum(counts), width), dtype=torch.float32)
            out[:counts[0]] = local
            base = counts[0]
            for r in range(1, self.world):
                for lo in range(0, counts[r], chunk):
                    n = min(chunk, counts[r] - lo)
                    buf = torch.empty((n, width), dtype=torch.float32, device=self.gpu_device)
                    td.recv(buf, src=r)
                    out[base + lo: base + lo + n] = buf.cpu()
                base += counts[r]
            return out
        for lo in range(0, counts[self.rank], chunk):
            td.send(local[lo: lo + chunk].to(self.gpu_device).contiguous(), dst=0)
        return None

    def _save_memory(self) -> None:
        """`torch.save` of the two plain tensors, as the reference (hbird_eval.py:371-380).  Also under a row-sharded
        bank the files hold the WHOLE bank in the reference's row order (rank 0 writes them), so a bank saved by N
        ranks loads into one process, into the reference, or into any other number of ranks."""
        for path, name in ((self.f_mem_p, "feature"), (self.l_mem_p, "label")):
            if path is None:
                continue
            t = self.feature_memory if name == "feature" else self.label_memory
            if self.sharded:
                t = self._gather_to_rank0(t)
            if t is not None:
                torch.save(t, path)
                logger.info("Saved %s memory to: %s", name, path)
        if self.sharded and (self.f_mem_p is not None or self.l_mem_p is not None):
            torch.distributed.barrier()

    def _memory_files_exist(self) -> bool:
        return (self.f_mem_p is not None and self.l_mem_p is not None and os.path.isfile(self.f_mem_p)
                and os.path.isfile(self.l_mem_p))

    def _reusable_memory(self) -> bool:
        """May the file pair f_mem_p / l_mem_p stand in for a bank build?  (The reference always rebuilds and overwrites, hbird_eval.py:175;
        re-use is this engine's addition.)  The files must exist AND fit this run -- feature width, class count, equal row counts, and with
        memory_size set no more rows than it allows -- otherwise the bank is rebuilt with a warning.  Under torch.distributed RANK 0 decides
        and broadcasts: ranks that looked for themselves could disagree (a file system that is not shared, a half-written file) and would
        then meet in different collectives -- load_memory's and _create_memory's -- and hang."""
        ok = 0
        if self.rank == 0:
            ok = 1 if self._memory_files_exist() else 0
            if ok:
                try:
                    fm = torch.load(self.f_mem_p, mmap=True); lm = torch.load(self.l_mem_p, mmap=True)
                    why = None
                    if fm.dim() != 2 or lm.dim() != 2 or fm.shape[0] != lm.shape[0]:
                        why = f"shapes {tuple(fm.shape)} / {tuple(lm.shape)} are not [M, D] / [M, C]"
                    elif fm.shape[1] != self.feature_extractor.d_model:
                        why = f"feature width {fm.shape[1]} != the extractor's {self.feature_extractor.d_model}"
                    elif lm.shape[1] != self.num_classes:
                        why = f"{lm.shape[1]} label columns != num_classes {self.num_classes}"
                    elif self.memory_size is not None and fm.shape[0] > self.memory_size:
                        why = f"{fm.shape[0]} rows > memory_size {self.memory_size}"
                    if why is not None:
                        logger.warning("saved bank %s / %s does not fit this run (%s): rebuilding", self.f_mem_p, self.l_mem_p, why)
                        ok = 0
                except Exception as e:          # noqa: BLE001 -- an unreadable file is a reason to rebuild, not to fail
                    logger.warning("saved bank %s / %s cannot be read (%r): rebuilding", self.f_mem_p, self.l_mem_p, e)
                    ok = 0
        if self.world > 1:
            t = torch.tensor([ok], device=self.gpu_device)
            torch.distributed.broadcast(t, 0)
            ok = int(t.item())
            if ok and not self._memory_files_exist():
                raise RuntimeError(f"rank 0 decided to re-use the saved bank {self.f_mem_p} / {self.l_mem_p}, which rank {self.rank} cannot see: "
                                   "the bank files must be on a file system all ranks share")
        return bool(ok)

    def load_memory(self) -> bool:
        """Load a bank saved by `_save_memory` (or by the reference) and rebuild the index from it; under a
        row-sharded bank every rank takes its contiguous row range of the one file."""
        if self._memory_files_exist():
            fm = torch.load(self.f_mem_p, mmap=True)
            lm = torch.load(self.l_mem_p, mmap=True)
            lo, hi = hdist.shard_range(fm.shape[0], self.rank, self.world) if self.sharded else (0, fm.shape[0])
            with torch.cuda.device(self.gpu_device):
                self.index.reset()
                P = self._infer_label_denominator(lm) if self.compress_labels else 0
                if P != self.index.label_denominator:
                    self.index.set_label_denominator(P)
                self.index.reserve(max(1, hi - lo))
                self.index.add(fm[lo:hi].to(self.gpu_device), normalize=False)
                self.index.add_labels(lm[lo:hi].to(self.gpu_device))
                self._finalize_shards()
            logger.info("Loaded memory from disk.")
            return True
        for p in (self.f_mem_p, self.l_mem_p):
            if p is not None and os.path.isfile(p + ".rank0"):
                logger.warning("%s.rankN: per-rank bank files of an earlier version are no longer read -- a sharded bank is "
                               "saved as ONE reference-format file pair now (INTEGRATION.md); rebuilding", p)
        logger.warning("Memory files not found or paths not provided; skipping load.")
        return False

    # ------------------------------------------------------------------------------------------------
    # query path (reference evaluate 184-265, _find_nearest_key_to_query 611-637, _cross_attention 575-609)
    # ------------------------------------------------------------------------------------------------
    def find_neighbours(self, q_flat: torch.Tensor, k: Optional[int] = None):
        """q_flat [nq, D] on the GPU -> (idx int64 [nq,k] global ids, dist [nq,k]); un-normalised queries (625)."""
        k = self.n_neighbours if k is None else k
        self.index.use_current_stream()
        if not self.sharded:
            return self.index.search(q_flat, k, id_base=self.id_base)
        return hdist.sharded_search(self.index.search_scores, merge_topk, q_flat, k, self.id_base, self.metric,
                                    finish=self.index.distances_from_scores, merge_packed=merge_topk_packed)

    def _label_hat(self, feats: torch.Tensor, want_details: bool):
        """feats [B,N,D] -> label_hat [B,N,C] (+ neighbours when details are requested)."""
        B, N, D = feats.shape
        q = feats.reshape(B * N, D).contiguous()
        k = self.n_neighbours
        self.index.use_current_stream()
        if not self.sharded and not want_details:
            lh = self.index.search_aggregate(q, k, beta=0.02, id_base=0)
            return lh.view(B, N, -1), None, None
        idx, dist = self.find_neighbours(q, k)
        lh = self.index.aggregate(q, idx, dist, beta=0.02, id_base=self.id_base)
        return lh.view(B, N, -1), idx, dist

    def _replicated_label_rows(self, ids: torch.Tensor) -> torch.Tensor:
        """label_memory.index_select(0, ids) (hbird_eval.py:633) on the table replicated over the ranks; ids outside it give zeros."""
        tab = self._label_table[0]
        if getattr(self, "_label_table_P", 0):      # uint16 counts -> the fp32 values (a float32 division: K2's own quotient)
            ok = (ids >= 0) & (ids < tab.shape[0])
            # (tensor divisor: a Python-scalar divisor becomes a multiplication by the reciprocal on the GPU -- another rounding)
            out = tab[ids.clamp(0, max(0, tab.shape[0] - 1))].to(torch.float32) / torch.tensor(float(self._label_table_P), device=tab.device)
            out[~ok] = 0.0
            return out
        return ops.gather_rows(tab, ids)

    def _gather_details(self, idx: torch.Tensor, B: int, N: int):
        k = idx.shape[1]
        flat = idx.reshape(-1)
        if self.sharded:
            # rows live on their owners: every rank reconstructs its own rows, the sum fills the rest
            own = (flat >= self.id_base) & (flat < self.id_base + self.index.ntotal)
            kf = self.index.reconstruct(torch.where(own, flat, torch.full_like(flat, -1)), id_base=self.id_base)
            torch.distributed.all_reduce(kf)
            kl = self._replicated_label_rows(flat)
        else:
            kf = self.index.reconstruct(flat)
            kl = self.index.gather_labels(flat)
        return kf.view(B, N, k, -1), kl.view(B, N, k, -1)

    def evaluate(self, val_loader, eval_spatial_resolution: int, return_knn_details: bool = False,
                 ignore_index: int = 255, window: Optional[Tuple[int, int]] = None):
        """Reference evaluate (hbird_eval.py:205-263).  `window=(win, stride)` (not in the reference): the loader
        yields frames larger than the extractor's input; every win x win window goes through the hot path and the
        windows' upsampled soft predictions are summed on the device before the argmax (hbird_mi/tiling.py)."""
        if window is not None and return_knn_details:
            raise ValueError("return_knn_details is not available with sliding windows")
        metric = PredsmIoU(self.num_classes, self.num_classes, ignore_index=ignore_index, device=self.gpu_device,
                           store_reordered_preds=False)
        self.feature_extractor = self.feature_extractor.to(self.device)
        S = eval_spatial_resolution
        knns, knns_labels, knns_ca_labels = [], [], []
        logger.info("Starting evaluation loop...")
        with torch.no_grad(), torch.cuda.device(self.gpu_device):
            if self.sharded:
                self._evaluate_sharded(val_loader, S, metric, return_knn_details, knns, knns_labels, knns_ca_labels,
                                       window)
            else:
                # replica mode (idx_shard=False) under torch.distributed: validation batches are data-parallel, and a rank loads only its own
                batches = (enumerate(val_loader) if self.world == 1
                           else hdist.rank_batches(val_loader, lambda i: i % self.world == self.rank))
                # the NEXT batch is fetched from the loader and copied to the GPU (side stream) while this one is searched; with
                # `self.profile = True` every batch's stages are timed by events (stage_times() afterwards; bench.py --e2e)
                prof = self._stage_profile = [] if getattr(self, "profile", False) else None
                for bi, (x, y) in tqdm(self._prefetched(batches, window is None), desc="Evaluation loop"):
                    _, _, h, w = x.shape
                    ev = self._mark(prof)                                           # batch on the GPU
                    y = (y.to(self.gpu_device) * 255).long()                       # 219 (255 is NOT remapped here)
                    if window is not None:
                        metric.update(y, self._windowed_cluster_map(x, S, window))
                        continue
                    feats = self._tokens(x)                                       # 217 (stays on the GPU)
                    self._mark(prof, ev)                                            # ViT forward
                    label_hat, idx, _ = self._label_hat(feats, return_knn_details)  # 224-227
                    self._mark(prof, ev)                                            # K4 + K5
                    if return_knn_details:
                        kf, kl = self._gather_details(idx, feats.shape[0], feats.shape[1])
                        knns.append(kf.cpu()); knns_labels.append(kl.cpu()); knns_ca_labels.append(label_hat.cpu())
                    metric.update_from_label_hat(y, label_hat, S)                   # 235-243 + 252: upsample + argmax + confusion counts, one kernel
                    self._mark(prof, ev)                                            # K6 + K7
        jac, tp, fp, fn, _, _ = metric.compute(is_global_zero=True, sync_distributed=self.world > 1,
                                               return_reordered=False)            # 253
        if return_knn_details:
            details = {"knns": torch.cat(knns), "knns_labels": torch.cat(knns_labels),
                       "knns_ca_labels": torch.cat(knns_ca_labels)}
            logger.info("Evaluation complete (with KNN details).")
            return jac, details
        logger.info("Evaluation complete.")
        return jac

    # -- batch prefetch + stage timing of the evaluation loop ---------------------------------------------------------------------
    def _prefetched(self, batches, to_gpu: bool):
        """(bi, (x, y)) of `batches`, one batch ahead: while the caller works on batch i, batch i + 1 is pulled from the loader (its
        workers decode in the background anyway) and -- `to_gpu` -- copied to the GPU on a side stream from pinned memory, so that neither
        the loader nor PCIe sits between two searches.  The host time spent waiting for the loader is kept in `loader_wait_s`."""
        import time as _time
        self.loader_wait_s = 0.0
        self.h2d_ms = []
        copy = torch.cuda.Stream(self.gpu_device) if to_gpu else None
        main = torch.cuda.current_stream(self.gpu_device)

        def stage(item):
            bi, (x, y) = item
            if copy is None or not isinstance(x, torch.Tensor) or x.is_cuda:
                return bi, x, y, None, None
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(copy):
                e0.record(copy)
                xd = (x if x.is_pinned() else x.pin_memory()).to(self.gpu_device, non_blocking=True)
                yd = (y if y.is_pinned() else y.pin_memory()).to(self.gpu_device, non_blocking=True) if isinstance(y, torch.Tensor) and not y.is_cuda else y
                e1.record(copy)
            return bi, xd, yd, e0, e1

        it = iter(batches)

        def pull():
            t0 = _time.perf_counter()
            try:
                item = next(it)
            except StopIteration:
                item = None
            self.loader_wait_s += _time.perf_counter() - t0
            return None if item is None else stage(item)

        nxt = pull()
        while nxt is not None:
            bi, x, y, e0, e1 = nxt
            if e1 is not None:
                main.wait_event(e1)
                x.record_stream(main)
                if isinstance(y, torch.Tensor) and y.is_cuda:
                    y.record_stream(main)
                self.h2d_ms.append((e0, e1))
            nxt = pull()                    # enqueue the next batch's copy before this one's work is enqueued
            yield bi, (x, y)

    def _mark(self, prof, ev=None):
        """Stage boundary of the profiled evaluation loop: one event on the current stream."""
        if prof is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.gpu_device))
        if ev is None:
            ev = [e]
            prof.append(ev)
        else:
            ev.append(e)
        return ev

    def stage_times(self) -> Optional[Dict[str, float]]:
        """After evaluate() with `self.profile = True`: mean ms per batch of the stages (GPU time between events on the evaluation stream),
        of the prefetched H2D copies (side stream), and the host seconds spent waiting for the loader."""
        prof = getattr(self, "_stage_profile", None)
        if not prof:
            return None
        torch.cuda.synchronize(self.gpu_device)
        rows = [[ev[i].elapsed_time(ev[i + 1]) for i in range(3)] for ev in prof if len(ev) == 4]
        if not rows:
            return None
        n = len(rows)
        out = {"batches": n, "vit_forward_ms": sum(r[0] for r in rows) / n, "knn_k5_ms": sum(r[1] for r in rows) / n,
               "k6_k7_ms": sum(r[2] for r in rows) / n, "loader_wait_s_total": self.loader_wait_s}
        if self.h2d_ms:
            out["h2d_ms"] = sum(a.elapsed_time(b) for a, b in self.h2d_ms) / len(self.h2d_ms)
        return out

    def _windowed_cluster_map(self, x: torch.Tensor, S: int, window: Tuple[int, int]) -> torch.Tensor:
        """Frames x [B,3,H,W] -> class map [B,1,H,W]: hot path per window, stitched on the device."""
        win, stride = window
        B, _, h, w = x.shape
        acc = torch.zeros((B, h, w, self.num_classes), dtype=torch.float32, device=self.gpu_device)
        for y0, x0 in tiling.window_origins(h, w, win, stride):
            feats = self._tokens(x[..., y0:y0 + win, x0:x0 + win])
            label_hat, _, _ = self._label_hat(feats, False)
            ops.upsample_accumulate(label_hat, S, acc, y0, x0, win, win)
        return ops.argmax_channels(acc)

    def _evaluate_sharded(self, val_loader, S, metric, want_details, knns, knns_labels, knns_ca_labels, window=None):
        """Validation batches are dealt round-robin; all ranks step together so that every search sees all
        shards.  A rank that has run out of batches contributes zero queries.  With sliding windows the ranks also
        step window by window (all frames of a step must have the same size)."""
        n_batches = len(val_loader)
        D = self.index.d
        steps = (n_batches + self.world - 1) // self.world
        k = self.n_neighbours
        # every rank LOADS only the batches it is dealt (hdist.rank_batches: a DataLoader fetches just those index batches), and the
        # ragged query all-gather has a fixed shape when the loader's batch size is known: one collective per step, no host round
        # trip before it
        own = iter(hdist.rank_batches(val_loader, lambda i: i % self.world == self.rank))
        bsz = getattr(val_loader, "batch_size", None)
        max_rows = int(bsz) * S * S if isinstance(bsz, int) and bsz > 0 else None
        for step in range(steps):
            mine = next(own)[1] if step * self.world + self.rank < n_batches else None
            fh = fw = 0
            if window is not None:      # all ranks step through the same windows: agree on the step's frame size
                hw = torch.tensor([mine[0].shape[-2], mine[0].shape[-1]] if mine is not None else [0, 0], dtype=torch.int64, device=self.gpu_device)
                torch.distributed.all_reduce(hw, op=torch.distributed.ReduceOp.MAX)
                fh, fw = (int(v) for v in hw.cpu().tolist())
            origins = [(0, 0)] if window is None else tiling.window_origins(fh, fw, window[0], window[1])
            acc, cluster_map = None, None
            for y0, x0 in origins:
                if mine is not None:
                    x, y = mine
                    if window is not None:
                        x = x[..., y0:y0 + window[0], x0:x0 + window[0]]
                    h, w = x.shape[-2:]
                    feats = self._tokens(x)
                    B, N, _ = feats.shape
                    q = feats.reshape(B * N, D)
                else:
                    q = torch.zeros((0, D), dtype=torch.float32, device=self.gpu_device)
                qall, nq = hdist.allgather_rows(q, max_rows)      # ragged query batches, zero-padded (to the loader's batch size x tokens)
                mx = qall.shape[1]
                if mx == 0:
                    continue
                q_flat_all = qall.reshape(self.world * mx, D)
                idx, dist = self.find_neighbours(q_flat_all, k)   # collective inside
                lh_all = None
                if self.label_shard:
                    # every rank: the weights of the full lists, the label sum over the neighbours it owns; the all-reduce completes it
                    self.index.use_current_stream()
                    lh_all = self.index.aggregate_partial(q_flat_all, idx, dist, self._label_table[1], beta=0.02, id_base=self.id_base)
                    torch.distributed.all_reduce(lh_all)
                kf_all, kl_all = None, None
                if want_details:
                    # neighbour features live on their owning shard: every rank fills in its own rows, the
                    # all-reduce (sum with zeros) completes them -- a collective, so all ranks take part
                    flat = idx.reshape(-1)
                    own = (flat >= self.id_base) & (flat < self.id_base + self.index.ntotal)
                    kf_all = self.index.reconstruct(torch.where(own, flat, torch.full_like(flat, -1)),
                                                    id_base=self.id_base)
                    torch.distributed.all_reduce(kf_all)
                    kf_all = kf_all.view(self.world * mx, k, D)
                    if self.label_shard:       # neighbour label rows the same way (ids outside this shard come back as zero rows)
                        local = torch.where(own, flat - self.id_base, torch.full_like(flat, -1))
                        kl_all = (self.index.gather_labels(local) if self.index.ntotal
                                  else torch.zeros((flat.numel(), self.num_classes), device=self.gpu_device))
                        torch.distributed.all_reduce(kl_all)
                        kl_all = kl_all.view(self.world * mx, k, -1)
                if mine is None:
                    continue
                lo = self.rank * mx
                my_idx, my_dist = idx[lo:lo + q.shape[0]].contiguous(), dist[lo:lo + q.shape[0]].contiguous()
                self.index.use_current_stream()
                if self.label_shard:
                    label_hat = lh_all[lo:lo + q.shape[0]].contiguous().view(B, N, -1)
                else:
                    label_hat = self.index.aggregate(q.contiguous(), my_idx, my_dist, beta=0.02).view(B, N, -1)
                if want_details:
                    kl = (kl_all[lo:lo + q.shape[0]].reshape(B, N, k, -1) if self.label_shard
                          else self._replicated_label_rows(my_idx.reshape(-1)).view(B, N, k, -1))
                    knns.append(kf_all[lo:lo + q.shape[0]].reshape(B, N, k, D).cpu())
                    knns_labels.append(kl.cpu())
                    knns_ca_labels.append(label_hat.cpu())
                if window is None:
                    metric.update_from_label_hat((mine[1].to(self.gpu_device) * 255).long(), label_hat, S)   # K6 + K7 fused
                else:
                    if acc is None:
                        acc = torch.zeros((B, fh, fw, self.num_classes), dtype=torch.float32, device=self.gpu_device)
                    ops.upsample_accumulate(label_hat, S, acc, y0, x0, h, w)
            if mine is None:
                continue
            if window is not None:
                metric.update((mine[1].to(self.gpu_device) * 255).long(), ops.argmax_channels(acc))
        if want_details and not knns:
            z = torch.zeros(0)
            knns.append(z); knns_labels.append(z); knns_ca_labels.append(z)


def hbird_evaluation(model, d_model: int, patch_size: int, dataset_name: str, data_dir: str, batch_size: int = 64,
                     input_size: int = 224, augmentation_epoch: int = 1, device: str | torch.device = "cpu",
                     return_knn_details: bool = False, n_neighbours: int = 30, nn_method: str = "scann",
                     nn_params: Optional[Dict[str, Any]] = None, ftr_extr_fn=None,
                     memory_size: Optional[int] = None, num_workers: int = 8, ignore_index: int = 255,
                     train_fs_path: Optional[str] = None, val_fs_path: Optional[str] = None,
                     frame_size: Optional[Tuple[int, int]] = None, window_stride: Optional[int] = None,
                     f_mem_p: Optional[str] = None, l_mem_p: Optional[str] = None):
    """High-level entry point with the reference's signature (hbird_eval.py:640-660).

    Four trailing keywords are not in the reference: `frame_size=(H, W)` makes the datasets deliver H x W frames that
    are processed through `input_size` windows with stride `window_stride` (default: input_size, i.e. no overlap) --
    bank build from the window crops, evaluation stitched over the windows (BASELINE cfg-5, hbird_mi/tiling.py).
    `f_mem_p` / `l_mem_p`: the bank's file pair (the constructor arguments the reference never passes, hbird_eval.py:701-712):
    a first run builds the bank and saves it there, a later run with both files present loads it and skips the build --
    one bank reused across runs, for any number of ranks."""
    if nn_params is None:
        nn_params = {}
    eval_spatial_resolution = input_size // patch_size                                   # 671
    if ftr_extr_fn is None:
        feature_extractor = FeatureExtractor(model, eval_spatial_resolution=eval_spatial_resolution, d_model=d_model)
    else:
        feature_extractor = FeatureExtractorSimple(model, ftr_extr_fn=ftr_extr_fn,
                                                   eval_spatial_resolution=eval_spatial_resolution, d_model=d_model)
    from hbird_mi.data import get_dataset
    window = None
    if frame_size is not None:
        fh, fw = int(frame_size[0]), int(frame_size[1])
        window = (input_size, int(window_stride) if window_stride else input_size)
        if fh < input_size or fw < input_size:
            raise ValueError(f"frame_size {frame_size} smaller than the {input_size} x {input_size} window")
    elif window_stride is not None:
        raise ValueError("window_stride needs frame_size")
    dataset, ignore_index_local = get_dataset(dataset_name, data_dir, batch_size, num_workers,
                                              input_size if window is None else (fh, fw), train_fs_path, val_fs_path)
    dataset_size = dataset.get_train_dataset_size()
    num_classes = dataset.get_num_classes()
    train_loader = dataset.train_dataloader()
    val_loader = dataset.val_dataloader()
    if window is not None:
        train_loader = tiling.WindowedLoader(train_loader, window[0], window[1], frame_hw=(fh, fw))
        dataset_size *= train_loader.windows_per_frame()       # every window is a bank image
    evaluator = HbirdEvaluation(feature_extractor, train_loader, num_classes=num_classes, n_neighbours=n_neighbours,
                                augmentation_epoch=augmentation_epoch, device=device, nn_method=nn_method,
                                nn_params=nn_params, memory_size=memory_size, dataset_size=dataset_size,
                                f_mem_p=f_mem_p, l_mem_p=l_mem_p, reuse_memory=True)
    effective_ignore = ignore_index if ignore_index != 255 else ignore_index_local        # 715
    last_run_info.clear()
    last_run_info.update(bank_loaded=bool(evaluator.bank_loaded), bank_build_s=float(evaluator.bank_build_s),
                         bank_rows=int(evaluator.total_rows), train_batches_loaded=int(evaluator.batches_loaded))
    return evaluator.evaluate(val_loader, eval_spatial_resolution=eval_spatial_resolution,
                              return_knn_details=return_knn_details, ignore_index=effective_ignore, window=window)
