#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_s5; mkdir -p $OUT
timeout 3000 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
timeout 900 python tools/bench_ops.py $OUT/bench_ops.json > $OUT/bench_ops.txt 2>&1; cat $OUT/bench_ops.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -3 $OUT/bench_default.err
HBIRD_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 8 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum > $OUT/bench_8ranks_one_gpu_gloo.json 2> $OUT/bench_8ranks.err; echo "8 ranks rc=$?"; tail -3 $OUT/bench_8ranks.err
timeout 600 python bench.py --gpus 1 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum --no-e2e > $OUT/bench_1rank_same_bank.json 2>/dev/null
HBIRD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err; echo "rccl world1 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_s5/bench_*.json")):
    if f.endswith("bench_ops.json"): continue
    try: r = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    u = r.get("use_fp16_mode") or {}
    print(f.split("/")[-1], "n_gpus", r["n_gpus"], "q/s", round(r["value"]), "ms", round(r["ms_per_step"], 2), "frac", round(r["roofline"]["frac"], 4),
          "clock", r["roofline"].get("clock_ghz"), "busy", r["roofline"].get("mfma_busy"), "| fp16:", round(u.get("value", 0)), u.get("clock_ghz"), u.get("mfma_busy"),
          "|", (r.get("multi_gpu") or {}).get("selftest"), r.get("label_hat_checksum"))
    if "e2e" in r: print("   e2e:", json.dumps(r["e2e"]))
PY
