#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2n
python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r2n/pytest.log 2>&1; tail -4 gpurun_out/r2n/pytest.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2n/bench.json 2> gpurun_out/r2n/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r2n/bench.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["roofline"], r["config"]["schedule"], r.get("use_fp16_mode"))
PY
