#!/bin/bash
# A/B (same box, interleaved): phased searches under XCD shares with common cuts (prev) / cuts that follow the shares (this tree); then the fuzz on weighted lists
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_cuts; mkdir -p $OUT
P=open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so; N=open-hummingbird-eval_amd/lib/libhbird_hip.so
for shape in "2074072 384 12544 30" "1000000 384 12544 30" "2074072 384 12544 90" "10000000 768 21904 90" "600000 768 12544 30"; do
  echo "== $shape"; AB_ROUNDS=7 timeout 900 python tools/ab_lib.py $shape $P $N 2>&1 | grep -v amdgpu | tail -2
done > $OUT/cuts_ab.txt 2>&1
cat $OUT/cuts_ab.txt
FUZZ_XCD=1 timeout 1200 python tests/fuzz_small.py 400 61 > $OUT/fuzz_xcd_400.txt 2>&1; tail -1 $OUT/fuzz_xcd_400.txt
FUZZ_XCD=1 FUZZ_ONE_LAUNCH=1 timeout 1200 python tests/fuzz_small.py 120 62 > $OUT/fuzz_xcd_one_launch_120.txt 2>&1; tail -1 $OUT/fuzz_xcd_one_launch_120.txt; grep -c "one_launch 1" $OUT/fuzz_xcd_one_launch_120.txt
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
