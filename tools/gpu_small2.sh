#!/bin/bash
export TMPDIR=/tmp
for wg in 0 -8 -16; do
  HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_ablation.so python bench.py --rows 50176 --dim 384 --nq 12544 --classes 21 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --workgroups $wg 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('wg $wg', round(r['ms_per_step'],3), round(r['roofline']['avg_kernel_ms'],3))"
done
