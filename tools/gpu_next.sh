#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2final4; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py --rows 5000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_shard_5000000.json 2>/dev/null
python - <<'PY'
import json
for n in ("default", "shard_5000000"):
    r = json.load(open(f"gpurun_out/r2final4/bench_{n}.json")); u = r.get("use_fp16_mode") or {}
    print(n, round(r["value"]), round(r["ms_per_step"], 1), round(r["roofline"]["frac"], 4), r["roofline"]["kernel"], r["roofline"]["traffic"], r["config"]["schedule"].get("cluster"), r.get("without_clusters"), "| fp16", round(u.get("value", 0)), round(u.get("ms_per_step", 0), 1))
PY
