#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2r
python -m pytest tests -m gpu -x -q > gpurun_out/r2r/pytest.log 2>&1; tail -4 gpurun_out/r2r/pytest.log
HBIRD_KNN_VARIANT=2 python -m pytest tests/test_knn_gpu.py -m gpu -x -q -k "fp16 or random_shapes or candidate_pool or clustered" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2r/bench.json 2> gpurun_out/r2r/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r2r/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"]["frac"], r["roofline"]["traffic"], r["config"]["schedule"], r.get("use_fp16_mode"))
PY
