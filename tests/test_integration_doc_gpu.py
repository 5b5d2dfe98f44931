"""The ctypes stub that INTEGRATION.md tells a reference maintainer to add is executed as written (only the library
path and the base-class import are pointed at this repository) and must return the oracle's neighbours."""
import os
import re

import numpy as np
import pytest
import torch

import golden_inputs as gi
import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches_the_oracle(cuda_device):
    from hbird_mi import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes.*?)```", md, re.S).group(1)
    assert "class NearestNeighborSearchHIP" in code
    code = code.replace("from hbird.nn.search_base import NearestNeighborSearchBase",
                        "from hbird_mi.nn.search_base import NearestNeighborSearchBase")
    code = code.replace("/path/to/libhbird_hip.so", _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    M, D, nq, k = 7000, 96, 300, 30
    bank = gi.unit_bank(M, D, seed=1)
    q = gi.vit_like_queries(nq, D, seed=2)
    for metric in ("dot_product", "l2"):
        nn = ns["NearestNeighborSearchHIP"](torch.from_numpy(bank), n_neighbors=k, distance_measure=metric, gpu_ids=[0])
        idx, dist = nn.find_nearest_neighbors(torch.from_numpy(q))
        ridx, rdist = oracle.knn_chain_f32(q, bank, k, metric)
        assert np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
        idx5, _ = nn.find_nearest_neighbors(torch.from_numpy(q), k=5)          # k override (search_faiss.py:84-85)
        assert np.array_equal(idx5, ridx[:, :5])
    with pytest.raises(ValueError):
        ns["NearestNeighborSearchHIP"](torch.from_numpy(bank), distance_measure="cosine")
