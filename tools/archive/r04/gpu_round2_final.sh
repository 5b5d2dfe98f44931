#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2final; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py --steps 10 --warmup 3 --fp16 --no-cpu-baseline --no-traffic > $OUT/bench_fp16.json 2>/dev/null
HBIRD_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_2ranks_one_gpu_gloo.json 2>/dev/null
HBIRD_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2>/dev/null
# per-rank share of the headline bank (what one rank of 2 / 4 / 8 searches)
for rows in 5000000 2500000 1250000; do
  python bench.py --rows $rows --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_shard_$rows.json 2>/dev/null
done
# BASELINE shapes on one GPU
for cfg in "cfg1 50176 384 21 12544 30" "cfg2 2074072 384 21 12544 30" "cfg4 20345364 1024 15 21904 30" "cfg5 10000000 768 19 21904 90"; do set -- $cfg
  python bench.py --rows $2 --dim $3 --classes $4 --nq $5 --k $6 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_$1.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2final/bench_*.json")):
    try: r = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    u = r.get("use_fp16_mode") or {}
    print(f.split("/")[-1], "n_gpus", r["n_gpus"], "q/s", round(r["value"]), "ms", round(r["ms_per_step"], 2), "frac", round(r["roofline"]["frac"], 4), "kernel_ms", round(r["roofline"]["avg_kernel_ms"], 2),
          "traffic", r["roofline"].get("traffic"), "| fp16 mode:", round(u.get("value", 0)), round(u.get("ms_per_step", 0), 1), round(u.get("candidate_kernel_frac_of_fp16_mfma_peak", 0), 3), "|", r.get("multi_gpu", {}).get("knn_ms_per_rank"), r.get("multi_gpu", {}).get("exchange_ms_per_rank"))
PY
bash tools/gpu_profile.sh r2 > $OUT/profile.log 2>&1; tail -40 $OUT/profile.log
