#!/bin/bash
# the fp16 candidate kernel with calibrated XCD shares of its own: search after search beside equal shares; tests; fuzz with random shares in use_fp16 mode too
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_f16cal; mkdir -p $OUT
timeout 900 python tools/exp_xcd_auto.py 10000000 768 21904 30 f16 2074072 384 12544 30 f16 10000000 768 21904 90 f16 > $OUT/xcd_auto_f16.txt 2>&1; grep -v amdgpu $OUT/xcd_auto_f16.txt | cut -c1-260
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
FUZZ_XCD=1 timeout 1500 python tests/fuzz_small.py 500 71 > $OUT/fuzz_xcd_500.txt 2>&1; tail -1 $OUT/fuzz_xcd_500.txt; grep -c "fp16 True" $OUT/fuzz_xcd_500.txt
python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic > $OUT/bench_short.json 2>/dev/null; python -c "
import json; r=json.load(open('$OUT/bench_short.json')); u=r['use_fp16_mode']; print('fp32', round(r['value']), round(r['roofline']['frac'],4), '| fp16', round(u['value']), round(u['ms_per_step'],1), round(u['candidate_kernel_frac_of_fp16_mfma_peak'],4), u.get('xcd_shares'))"
