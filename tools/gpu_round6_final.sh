#!/bin/bash
# The measurement session of round 6 (one gpurun call): full -m gpu suite, the driver-shaped bench line (with its A/B legs), the N-rank dry run
# on one GPU, the shard sizes behind the pre-registered scaling model, every BASELINE shape, the use_fp16 cliff and residency tools, the
# secondary kernels, then the rocprofv3 evidence (tools/gpu_profile.sh).  Output: gpurun_out/r6final/.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/${R6OUT:-r6final}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"     # exactly what the driver runs
python bench.py --steps 10 --warmup 3 --fp16 --no-cpu-baseline --no-traffic --no-e2e > $OUT/bench_fp16.json 2>/dev/null
HBIRD_BENCH_ONE_GPU=1 python bench.py --gpus 8 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum > $OUT/bench_8ranks_one_gpu_gloo.json 2> $OUT/bench_8ranks.err
python bench.py --gpus 1 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum --no-e2e > $OUT/bench_1rank_same_bank.json 2>/dev/null
HBIRD_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
for rows in 5000000 2500000 1250000; do
  python bench.py --rows $rows --steps 5 --warmup 3 --no-cpu-baseline --no-traffic --no-e2e > $OUT/bench_shard_$rows.json 2>/dev/null
done
for cfg in "cfg1 50176 384 21 12544 30" "cfg2 2074072 384 21 12544 30" "cfg4 20345364 1024 15 21904 30" "cfg5 10000000 768 19 21904 90"; do set -- $cfg
  python bench.py --rows $2 --dim $3 --classes $4 --nq $5 --k $6 --steps 5 --warmup 3 --no-cpu-baseline --no-traffic --e2e-batches 2 > $OUT/bench_$1.json 2>/dev/null
done
python tools/bench_ops.py $OUT/bench_ops.json > $OUT/bench_ops.txt 2>&1
R6OUT=${R6OUT:-r6final} python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/" + os.environ.get("R6OUT", "r6final") + "/bench_*.json")):
    if f.endswith("bench_ops.json"): continue
    try: r = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    rf = r["roofline"]
    print(f.split("/")[-1], "n_gpus", r["n_gpus"], "q/s", round(r["value"]), "ms", round(r["ms_per_step"], 2), "frac", round(rf["frac"], 4), "kernel_ms", round(rf["avg_kernel_ms"], 2),
          "[", rf.get("kernel_ms_min"), rf.get("kernel_ms_max"), "] clock", rf.get("clock_ghz_unprofiled"), "frac@clock", rf.get("frac_of_peak_at_measured_clock"), "replans", rf.get("work_list_replans_in_timed_steps"),
          "| equal/cal ms", rf.get("equal_shares_kernel_ms"), rf.get("calibrated_shares_kernel_ms"), "| no clusters", rf.get("without_clusters_kernel_ms"),
          "| fp16", rf.get("fp16_value"), rf.get("fp16_ms_per_step"), rf.get("fp16_frac_of_fp16_peak"), "clock", rf.get("fp16_clock_ghz_unprofiled"),
          "| e2e", rf.get("e2e_fp32_images_per_s"), rf.get("e2e_fp16_images_per_s"), "| traffic", rf.get("traffic"),
          "|", r.get("multi_gpu", {}).get("rows_per_rank"), r.get("label_hat_checksum"))
    cb = r.get("cpu_baseline") or {}
    if cb: print("     cpu:", cb.get("value"), "extrapolated", cb.get("extrapolated"), "|", (cb.get("sample") or "")[:160])
PY
grep -i "nccl\|rccl" $OUT/bench_rccl_world1.err | head -3
python tools/exp_fp16_cliff.py 2000000 384 12544 30 21 $OUT/fp16_cliff_2Mx384.json 0.5 0.4 0.3 0.2 0.1 0.05 > $OUT/fp16_cliff_2Mx384.txt 2>&1
python tools/exp_fp16_cliff.py 10000000 768 21904 30 151 $OUT/fp16_cliff_10Mx768.json 0.5 0.47 0.45 0.4 0.3 0.2 0.1 > $OUT/fp16_cliff_10Mx768.txt 2>&1; cut -c1-400 $OUT/fp16_cliff_10Mx768.txt
python tools/exp_fp16_residency.py $OUT/fp16_residency.json 2074072 384 12544 30 10000000 768 21904 30 20345364 1024 21904 30 27700000 768 21904 30 > $OUT/fp16_residency.txt 2>&1; cut -c1-600 $OUT/fp16_residency.txt
python tools/exp_clock_guard.py 10000000 768 21904 30 8 > $OUT/clock_guard_headline.txt 2>&1
# pool searches: unphased / shipped, and small fp32 searches on lists (variant 6) / shipped
S="50176 384 12544 30 f16 300000 768 12544 30 f16 2074072 384 12544 30 f16 50176 384 12544 90 f32 2074072 384 12544 90 f32"
{ EXP_PHASES=0 python tools/exp_phases.py $S 2>&1 | grep phases | sed 's/^/unphased /'; python tools/exp_phases.py $S 2>&1 | grep phases | sed 's/^/shipped  /'; } > $OUT/pool_searches_ab.txt
cat $OUT/pool_searches_ab.txt
python tools/exp_k_sweep.py 2074072 384 12544 "30,90,256,257,512,1024,2048" f32 > $OUT/k_sweep.txt 2>&1; tail -8 $OUT/k_sweep.txt
bash tools/gpu_profile.sh r6 > $OUT/profile.log 2>&1; tail -30 $OUT/profile.log
