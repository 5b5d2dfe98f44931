#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_s8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -s -k "one_launch or phased" > $OUT/pytest_ol.log 2>&1; tail -6 $OUT/pytest_ol.log | grep -v amdgpu
FUZZ_ONE_LAUNCH=1 timeout 1500 python tests/fuzz_small.py 60 11 > $OUT/fuzz_one_launch.txt 2>&1; tail -3 $OUT/fuzz_one_launch.txt; grep -c "one_launch 1" $OUT/fuzz_one_launch.txt
timeout 900 python tests/fuzz_small.py 60 12 > $OUT/fuzz_default.txt 2>&1; tail -2 $OUT/fuzz_default.txt
