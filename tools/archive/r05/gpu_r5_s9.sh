#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_s9; mkdir -p $OUT
HBIRD_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 8 --rows 600001 --dim 64 --classes 21 --nq 3001 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --checksum > $OUT/bench_8ranks_one_gpu_gloo.json 2> $OUT/bench_8ranks.err; echo "8 ranks rc=$?"
HBIRD_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --rows 300001 --dim 384 --classes 151 --nq 2738 --k 90 --fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --checksum > $OUT/bench_2ranks_fp16_k90.json 2> $OUT/bench_2ranks.err; echo "2 ranks fp16 k90 rc=$?"
python - <<'PY'
import json
for f in ("bench_8ranks_one_gpu_gloo.json", "bench_2ranks_fp16_k90.json"):
    r = json.load(open("gpurun_out/r5_s9/" + f)); m = r["multi_gpu"]
    print(f, r["n_gpus"], round(r["value"]), m["selftest"], m["replicated_label_table"], r.get("label_hat_checksum"))
PY
