"""Shared test helpers (no reference imports)."""
import numpy as np
import torch


class ReplayExtractor(torch.nn.Module):
    """Fake extractor returning pre-computed tokens in call order (same as tests/golden/gen_golden.py)."""

    def __init__(self, tokens, eval_spatial_resolution, d_model):
        super().__init__()
        self.tokens = [torch.from_numpy(np.ascontiguousarray(t)) for t in tokens]
        self.eval_spatial_resolution = eval_spatial_resolution
        self.d_model = d_model
        self.i = 0

    def forward_features(self, x):
        t = self.tokens[self.i]
        self.i += 1
        return t.clone().to(x.device), None


def golden_case(g, name):
    C, D, H, ps, nb, B, k, mem, aug, ign = g[f"cfg_{name}"].tolist()
    train = [(torch.zeros((B, 3, H, H)), torch.from_numpy(g[f"train_y_{name}_{i}"])) for i in range(nb)]
    val = [(torch.zeros((B, 3, H, H)), torch.from_numpy(g[f"val_y_{name}_{i}"])) for i in range(2)]
    tr_tok = [g[f"train_tok_{name}_{i}"] for i in range(nb)] * aug
    va_tok = [g[f"val_tok_{name}_{i}"] for i in range(2)]
    return dict(C=C, D=D, H=H, ps=ps, nb=nb, B=B, k=k, mem=None if mem < 0 else mem, aug=aug, ign=ign,
                train=train, val=val, tr_tok=tr_tok, va_tok=va_tok, S=H // ps)


class IndexedReplayExtractor(torch.nn.Module):
    """Like ReplayExtractor, but the batch identifies itself through x[0,0,0,0] (so that ranks which skip
    batches in a sharded build still get the right tokens)."""

    def __init__(self, tokens_by_key, eval_spatial_resolution, d_model):
        super().__init__()
        self.tokens = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in tokens_by_key.items()}
        self.eval_spatial_resolution = eval_spatial_resolution
        self.d_model = d_model

    def forward_features(self, x):
        return self.tokens[int(x[0, 0, 0, 0].item())].clone().to(x.device), None


def golden_case_indexed(g, name):
    c = golden_case(g, name)
    tok = {}
    for i, (x, _) in enumerate(c["train"]):
        x[0, 0, 0, 0] = float(i); tok[i] = c["tr_tok"][i]
    for i, (x, _) in enumerate(c["val"]):
        x[0, 0, 0, 0] = float(1000 + i); tok[1000 + i] = c["va_tok"][i]
    c["tokens_by_key"] = tok
    return c


def chain_oracle_topk_chunked(ix, q_sel, M, k, metric="dot_product", chunk=1_000_000):
    """The fp32 chain oracle (oracle.knn_chain_f32: the bit-exact target of the kernels) over a device-resident bank of ANY size:
    the bank is reconstructed chunk by chunk (no 30-80 GB host copy), each chunk is searched by the oracle with its id base, and the
    per-chunk lists are merged on the host by the kernels' ordering key (score descending / distance ascending, then id ascending).
    A row's chain score does not depend on which chunk it sits in, so the merged list is the oracle's answer for the whole bank.
    -> (idx int64 [n, k], dist float32 [n, k]) as numpy."""
    import oracle
    qn = q_sel.detach().cpu().numpy().astype(np.float32)
    n = qn.shape[0]
    best_i = np.full((n, 0), -1, dtype=np.int64)
    best_d = np.zeros((n, 0), dtype=np.float32)
    sign = 1.0 if metric.lower() in ("l2", "euclidean") else -1.0          # sort key: ascending
    for r in range(0, M, chunk):
        ids = torch.arange(r, min(M, r + chunk), device=q_sel.device)
        rows = ix.reconstruct(ids)
        rows = rows.cpu().numpy() if isinstance(rows, torch.Tensor) else np.asarray(rows)
        ci, cd = oracle.knn_chain_f32(qn, rows, min(k, rows.shape[0]), metric, id_base=r)
        del rows
        ai = np.concatenate([best_i, ci], axis=1); ad = np.concatenate([best_d, cd], axis=1)
        order = np.lexsort((ai, sign * ad.astype(np.float64)), axis=1)[:, :k]      # primary: the distance, ties: the lower id
        best_i = np.take_along_axis(ai, order, axis=1); best_d = np.take_along_axis(ad, order, axis=1)
    return best_i, best_d
