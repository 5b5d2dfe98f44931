#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_xw2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -s -k "xcd" > $OUT/pytest_xcd.log 2>&1; grep -v amdgpu $OUT/pytest_xcd.log | tail -5
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-e2e > $OUT/bench_default.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic --rows 20345364 --dim 1024 --classes 15 > $OUT/bench_cfg4.json 2>/dev/null
timeout 900 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic --k 90 --classes 19 > $OUT/bench_cfg5.json 2>/dev/null
timeout 900 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic --rows 1250000 > $OUT/bench_shard_1250000.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default", "bench_cfg4", "bench_cfg5", "bench_shard_1250000"):
    try: r = json.load(open(f"gpurun_out/r5_xw2/{f}.json"))
    except Exception as e: print(f, "failed", e); continue
    ro = r["roofline"]; u = r.get("use_fp16_mode") or {}
    print(f, round(r["value"]), "q-p/s", round(r["ms_per_step"], 1), "ms, kernel", round(ro["avg_kernel_ms"], 1), "frac", round(ro["frac"], 4), "clock", ro.get("clock_ghz"), "busy", ro.get("mfma_busy"),
          "| without_clusters", (r.get("without_clusters") or {}).get("frac"), "| fp16", round(u.get("value", 0)), u.get("candidate_kernel_frac_of_fp16_mfma_peak"), "| shares", r["config"].get("xcd_shares"))
PY
