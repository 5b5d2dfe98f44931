#!/bin/bash
# The measurement session of round 5 (one gpurun call): full -m gpu suite, the driver-shaped bench line, the N-rank dry runs on one GPU,
# every BASELINE shape, the secondary kernels, then the rocprofv3 evidence (tools/gpu_profile.sh).  Output: gpurun_out/r5final/.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/${R5OUT:-r5final}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"     # exactly what the driver runs
python bench.py --steps 10 --warmup 3 --fp16 --no-cpu-baseline --no-traffic --no-e2e > $OUT/bench_fp16.json 2>/dev/null
HBIRD_BENCH_ONE_GPU=1 python bench.py --gpus 8 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum > $OUT/bench_8ranks_one_gpu_gloo.json 2> $OUT/bench_8ranks.err
python bench.py --gpus 1 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum --no-e2e > $OUT/bench_1rank_same_bank.json 2>/dev/null
HBIRD_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
for rows in 5000000 2500000 1250000; do
  python bench.py --rows $rows --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-e2e > $OUT/bench_shard_$rows.json 2>/dev/null
done
for cfg in "cfg1 50176 384 21 12544 30" "cfg2 2074072 384 21 12544 30" "cfg4 20345364 1024 15 21904 30" "cfg5 10000000 768 19 21904 90"; do set -- $cfg
  python bench.py --rows $2 --dim $3 --classes $4 --nq $5 --k $6 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --e2e-batches 2 > $OUT/bench_$1.json 2>/dev/null
done
python tools/bench_ops.py $OUT/bench_ops.json > $OUT/bench_ops.txt 2>&1
R5OUT=${R5OUT:-r5final} python - <<'PY'
import json, glob
import os
for f in sorted(glob.glob("gpurun_out/" + os.environ.get("R5OUT", "r5final") + "/bench_*.json")):
    if f.endswith("bench_ops.json"): continue
    try: r = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    u = r.get("use_fp16_mode") or {}
    print(f.split("/")[-1], "n_gpus", r["n_gpus"], "q/s", round(r["value"]), "ms", round(r["ms_per_step"], 2), "frac", round(r["roofline"]["frac"], 4), "kernel_ms", round(r["roofline"]["avg_kernel_ms"], 2),
          "traffic", r["roofline"].get("traffic"), "| fp16 mode:", round(u.get("value", 0)), round(u.get("ms_per_step", 0), 1), round(u.get("candidate_kernel_frac_of_fp16_mfma_peak", 0), 3), "|",
          r.get("multi_gpu", {}).get("rows_per_rank"), r.get("label_hat_checksum"), "| selftest", (r.get("multi_gpu") or {}).get("selftest", {}) if isinstance((r.get("multi_gpu") or {}).get("selftest"), str) else ((r.get("multi_gpu") or {}).get("selftest") or {}).get("ids_and_score_bits_equal_the_chain_oracle"))
    e = r.get("e2e") or {}
    for mode in ("fp32", "use_fp16"):
        if mode in e: print("     e2e", mode, round(e[mode]["images_per_s"], 2), "img/s", e[mode]["per_batch_ms"], "bound by", e[mode]["bound_by"])
    if r["roofline"].get("clock_ghz"): print("     counters: fp32 clock", round(r["roofline"]["clock_ghz"], 3), "busy", round(r["roofline"]["mfma_busy"], 3), "| fp16 clock", u.get("clock_ghz"), "busy", u.get("mfma_busy"))
PY
grep -i "nccl\|rccl" $OUT/bench_rccl_world1.err | head -3
# pool searches: unphased / shipped, and small fp32 searches on lists (variant 6) / shipped
S="50176 384 12544 30 f16 300000 768 12544 30 f16 2074072 384 12544 30 f16 50176 384 12544 90 f32 2074072 384 12544 90 f32"
L="50176 384 12544 30 f32 50176 384 21904 30 f32 200000 384 12544 30 f32 2074072 384 12544 30 f32 20000 384 784 30 f32"
{ EXP_PHASES=0 python tools/exp_phases.py $S 2>&1 | grep phases | sed 's/^/unphased /'; python tools/exp_phases.py $S 2>&1 | grep phases | sed 's/^/shipped  /';
  EXP_VARIANT=6 python tools/exp_phases.py $L 2>&1 | grep phases | sed 's/^/lists    /'; python tools/exp_phases.py $L 2>&1 | grep phases | sed 's/^/shipped  /'; } > $OUT/pool_searches_ab.txt
cat $OUT/pool_searches_ab.txt
python tools/exp_fp16_crossover.py 384 12544 30 16384 50176 2074072 > $OUT/fp16_whole_search.txt 2>&1; python tools/exp_fp16_crossover.py 768 21904 30 50176 1250000 >> $OUT/fp16_whole_search.txt 2>&1; cat $OUT/fp16_whole_search.txt
# a launch per phase (default) against ONE launch with grid barriers (opt-in), with block 0's barrier statistics
python tools/exp_one_launch.py 50176 384 12544 30 f32 50176 384 12544 30 f16 300000 768 12544 30 f16 2074072 384 12544 30 f16 2074072 384 12544 30 f32 > $OUT/one_launch_ab.txt 2>&1; grep -v amdgpu $OUT/one_launch_ab.txt
bash tools/gpu_profile.sh r5 > $OUT/profile.log 2>&1; tail -30 $OUT/profile.log
