#!/bin/bash
# the driver's 8-rank command at the headline size, rehearsed with eight ranks on ONE GPU over gloo (RCCL refuses two ranks on a device):
# same code path as `torchrun --nproc-per-node 8 bench.py --gpus 8` except the backend and the devices
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_rehearsal; mkdir -p $OUT
s=$(date +%s)
HBIRD_BENCH_ONE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --steps 3 --warmup 1 > $OUT/bench_8ranks_headline_one_gpu.json 2> $OUT/bench_8ranks_headline.err; echo "rc=$? seconds=$(( $(date +%s) - s ))"
tail -3 $OUT/bench_8ranks_headline.err | cut -c1-300
timeout 900 python bench.py --rows 2000000 --dim 1536 --nq 10952 --classes 151 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_vitg_2Mx1536.json 2>/dev/null; echo "vit-g rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5_rehearsal/bench_8ranks_headline_one_gpu.json")); m = r["multi_gpu"]
print("8 ranks, headline bank:", round(r["value"]), "q-p/s", round(r["ms_per_step"], 1), "ms/step; knn ms per rank", m["knn_ms_per_rank"], "exchange", m["exchange_ms_per_rank"], "split", m["exchange_split_ms_per_rank"],
      "efficiency", round(m["efficiency"], 3), "selftest", m["selftest"].get("ids_and_score_bits_equal_the_chain_oracle") if isinstance(m["selftest"], dict) else m["selftest"], m["replicated_label_table"])
print("   use_fp16_mode", r.get("use_fp16_mode", {}).get("ms_per_step"), r.get("use_fp16_mode", {}).get("knn_ms_per_rank"))
g = json.load(open("gpurun_out/r5_rehearsal/bench_vitg_2Mx1536.json")); u = g.get("use_fp16_mode", {})
print("ViT-g width 2 M x 1536, 10,952 queries:", round(g["value"]), "q-p/s", round(g["ms_per_step"], 1), "ms, frac", round(g["roofline"]["frac"], 4), "| use_fp16", round(u.get("value", 0)), round(u.get("ms_per_step", 0), 1), round(u.get("candidate_kernel_frac_of_fp16_mfma_peak", 0), 3))
e = g.get("e2e", {})
for mode in ("fp32", "use_fp16"):
    if mode in e: print("   e2e", mode, round(e[mode]["images_per_s"], 2), e[mode]["per_batch_ms"], e[mode]["bound_by"])
PY
