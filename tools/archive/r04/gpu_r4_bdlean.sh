#!/bin/bash
# Round 4, experiment 4: the fp32 kernel (hbird_knn_bd.hip) with running fetch pointers + saddr LDS-DMA against the previous commit's library,
# same box, interleaved (tools/ab_lib.py), after the kNN parity tests.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_bdlean; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py -m gpu -x -q -k "not headline" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
L=$ROOT/open-hummingbird-eval_amd/lib
for shape in "10000000 768 21904 30" "5000000 768 21904 90" "2074072 384 12544 30" "50176 384 12544 30" "1250000 768 21904 30"; do
  timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_prelean.so $L/libhbird_hip.so 2>&1 | tail -4 | sed "s/^/$shape: /" >> $OUT/ab.txt
done
cat $OUT/ab.txt
