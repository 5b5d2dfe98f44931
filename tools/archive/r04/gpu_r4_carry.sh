#!/bin/bash
# Round 4: A/B of the fp16 candidate kernel against an earlier build (lib/abl/libhbird_hip_<tag>.so), same box, interleaved rounds:
# kernel ms of use_fp16 searches at the headline shape, k = 90, cfg-2, cfg-4-ish, cfg-1, and the headline shape under L2.
# Used for: one deferred candidate per lane ("nocarry" = without; no gain, dropped) and the C = 0 tile start ("nocarry" = before it).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/${1:-r4_carry_ab}; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or f16 or cluster or random or pool or phase or nan or overflow or config or cfg" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for shape in "10000000 768 21904 30" "10000000 768 21904 90" "2074072 384 12544 30" "5000000 1024 21904 30" "50176 384 12544 30"; do
  AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_${2:-nocarry}.so $L/libhbird_hip.so 2>&1 | tail -2 | sed "s/^/$shape: /" | tee -a $OUT/t.txt
done
AB_METRIC=1 AB_FP16=1 timeout 900 python tools/ab_lib.py 10000000 768 21904 30 $L/abl/libhbird_hip_${2:-nocarry}.so $L/libhbird_hip.so 2>&1 | tail -2 | sed "s/^/L2 10000000 768 21904 30: /" | tee -a $OUT/t.txt
