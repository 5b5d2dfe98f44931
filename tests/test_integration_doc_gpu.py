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


def test_integration_md_c_snippet_compiles_and_matches_the_oracle(cuda_device, tmp_path):
    """The C snippet of INTEGRATION.md section 4 (a host that is not Python: hb_multi_* over several GPUs) is compiled as
    written with gcc against include/hbird_hip.h + libhbird_hip.so, run on a bank read from a file, and must return the oracle's
    neighbours (cuda:0 listed four times: four row shards on the one GPU of this box)."""
    import subprocess
    from hbird_mi import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippet = re.search(r"```c\n(.*?)```", md, re.S).group(1)
    assert "hb_multi_create" in snippet and "hb_multi_search" in snippet
    M, D, nq, k = 30_011, 768, 200, 30
    bank = gi.unit_bank(M, D, seed=3)
    q = gi.vit_like_queries(nq, D, seed=4)
    bank.tofile(tmp_path / "bank.f32"); q.tofile(tmp_path / "q.f32")
    body = snippet.replace("{0, 1, 2, 3}", "{0, 0, 0, 0}")             # this box has one GPU
    src = f"""
#include <stdio.h>
#include <stdlib.h>
#include "hbird_hip.h"
static void* slurp(const char* p, size_t bytes) {{ void* b = malloc(bytes); FILE* f = fopen(p, "rb"); if (!f || fread(b, 1, bytes, f) != bytes) exit(3); fclose(f); return b; }}
int main(void) {{
    const int64_t n_rows = {M}, nq = {nq};
    float* bank = slurp("{tmp_path}/bank.f32", (size_t)n_rows * 768 * 4);
    float* q = slurp("{tmp_path}/q.f32", (size_t)nq * 768 * 4);
    int64_t* idx = malloc((size_t)nq * 30 * 8); float* dist = malloc((size_t)nq * 30 * 4);
    {body}
    FILE* f = fopen("{tmp_path}/idx.i64", "wb"); fwrite(idx, 8, (size_t)nq * 30, f); fclose(f);
    f = fopen("{tmp_path}/dist.f32", "wb"); fwrite(dist, 4, (size_t)nq * 30, f); fclose(f);
    printf("%s\\n", hb_last_error());
    return 0;
}}
"""
    (tmp_path / "host.c").write_text(src)
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), str(tmp_path / "host.c"), "-L", libdir,
                    "-lhbird_hip", f"-Wl,-rpath,{libdir}", "-o", str(tmp_path / "host")], check=True)
    r = subprocess.run([str(tmp_path / "host")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    idx = np.fromfile(tmp_path / "idx.i64", dtype=np.int64).reshape(nq, k)
    dist = np.fromfile(tmp_path / "dist.f32", dtype=np.float32).reshape(nq, k)
    ridx, rdist = oracle.knn_chain_f32(q, bank, k, "dot_product")
    assert np.array_equal(idx, ridx) and np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
