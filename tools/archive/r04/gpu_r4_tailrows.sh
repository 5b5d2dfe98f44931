#!/bin/bash
# Round 4: leftover query tiles in rows of their own shape (no idle cluster members): parity tests, then kernel ms against the build
# before (lib/abl/libhbird_hip_nocarry.so = the previous commit) at the headline shape and at 49 query tiles, fp16 and fp32.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_tailrows; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not headline and not full_size" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for shape in "10000000 768 21904 30" "5000000 768 12544 30" "10000000 768 21904 90"; do
  AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_nocarry.so $L/libhbird_hip.so 2>&1 | tail -2 | sed "s/^/fp16 $shape: /" | tee -a $OUT/t.txt
done
timeout 900 python tools/ab_lib.py 5000000 768 12544 30 $L/abl/libhbird_hip_nocarry.so $L/libhbird_hip.so 2>&1 | tail -2 | sed "s/^/fp32 5000000 768 12544 30: /" | tee -a $OUT/t.txt
EXP_ROWS=5000000 EXP_NQ=12544 EXP_MODES=f16 EXP_ROUNDS=3 EXP_CFGS="8,1,-1,2;4,2,-1,2;0,0,-1,0" EXP_OUT=r4_tailrows/f16_49.json timeout 900 python tools/exp_cluster.py 2>&1 | grep same_bits | tee -a $OUT/t.txt
