#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2final2; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --rows 10000000 --dim 768 --classes 19 --nq 21904 --k 90 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_cfg5.json 2>/dev/null
python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
for n in ("cfg5", "default"):
    r = json.load(open(f"gpurun_out/r2final2/bench_{n}.json")); u = r.get("use_fp16_mode") or {}
    print(n, round(r["value"]), round(r["ms_per_step"], 1), round(r["roofline"]["frac"], 4), r["roofline"]["kernel"], r["roofline"]["traffic"], "| fp16", round(u.get("value", 0)), round(u.get("ms_per_step", 0), 1))
PY
