#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_f16cal2; mkdir -p $OUT
timeout 900 python tools/exp_xcd_auto.py 10000000 768 21904 30 f16 2074072 384 12544 30 f16 > $OUT/xcd_auto_f16.txt 2>&1; grep -v amdgpu $OUT/xcd_auto_f16.txt | cut -c1-230
for i in 1 2; do python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic > $OUT/bench_short_$i.json 2>/dev/null; python -c "
import json; r=json.load(open('$OUT/bench_short_$i.json')); u=r['use_fp16_mode']; print('fp32', round(r['value']), round(r['roofline']['frac'],4), '| fp16', round(u['value']), round(u['ms_per_step'],1), round(u['candidate_kernel_frac_of_fp16_mfma_peak'],4), u.get('xcd_shares'), u.get('clock_ghz'), u.get('mfma_busy'))"; done
python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic --rows 2074072 --dim 384 --classes 21 --nq 12544 > $OUT/bench_cfg2.json 2>/dev/null; python -c "
import json; r=json.load(open('$OUT/bench_cfg2.json')); u=r['use_fp16_mode']; print('cfg2 fp32', round(r['value']), round(r['roofline']['frac'],4), '| fp16', round(u['value']), round(u['ms_per_step'],2), round(u['candidate_kernel_frac_of_fp16_mfma_peak'],4), u.get('xcd_shares'))"
