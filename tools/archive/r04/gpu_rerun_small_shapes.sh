#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r3final_b; mkdir -p $OUT
for cfg in "cfg1 50176 384 21 12544 30" "cfg2 2074072 384 21 12544 30"; do set -- $cfg
  python bench.py --rows $2 --dim $3 --classes $4 --nq $5 --k $6 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/bench_$1.json 2>/dev/null
done
for rows in 2500000 1250000; do
  python bench.py --rows $rows --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_shard_$rows.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3final_b/bench_*.json")):
    r = json.load(open(f)); u = r.get("use_fp16_mode") or {}
    print(f.split("/")[-1], "q/s", round(r["value"]), "ms", round(r["ms_per_step"], 3), "frac", round(r["roofline"]["frac"], 4), "kernel_ms", round(r["roofline"]["avg_kernel_ms"], 3))
PY
