#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/f16v2
HBIRD_KNN_VARIANT=2 timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or random_shapes or width_1024 or k90_width or cfg1 or candidate_pool or two_level" > gpurun_out/f16v2/pytest.log 2>&1; tail -5 gpurun_out/f16v2/pytest.log
python - <<'PY'
import os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for (M, D, nq) in ((10_000_000, 768, 21904), (2_074_072, 384, 12544)):
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ref = ix.search(q, 30)
    ix.set_fp16(True)
    for r in range(3):
        for v in (0, 2):
            ix.set_variant(v); ix.set_timing(True)
            i, d = ix.search(q, 30); ms = ix.last_knn_ms(); ix.set_timing(False)
            print(M, D, "variant", v, "ms", round(ms, 2), "same", bool(torch.equal(i, ref[0]) and torch.equal(d, ref[1])), "fallbacks", ix.last_fp16_fallbacks(), flush=True)
    del ix
PY
