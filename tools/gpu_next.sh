#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2final3; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py --rows 20345364 --dim 1024 --classes 15 --nq 21904 --k 30 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_cfg4.json 2>/dev/null
python bench.py --rows 5000000 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_shard_5000000.json 2>/dev/null
python - <<'PY'
import json
for n in ("default", "cfg4", "shard_5000000"):
    r = json.load(open(f"gpurun_out/r2final3/bench_{n}.json")); u = r.get("use_fp16_mode") or {}
    print(n, round(r["value"]), round(r["ms_per_step"], 1), round(r["roofline"]["frac"], 4), r["roofline"]["kernel"], r["roofline"]["traffic"], r["config"]["schedule"].get("cluster"), "| fp16", round(u.get("value", 0)), round(u.get("ms_per_step", 0), 1))
PY
bash tools/gpu_profile.sh r2c > $OUT/profile.log 2>&1; tail -30 $OUT/profile.log | grep -E "bd_kernel|csv$"
