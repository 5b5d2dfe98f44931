#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-.}
cd $ROOT
OUT=$ROOT/gpurun_out/r5_s6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "one_launch or phased" > $OUT/pytest_ol.log 2>&1; tail -4 $OUT/pytest_ol.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ops -- python3 $ROOT/tools/bench_ops.py > $OUT/bench_ops_under_rocprof.txt 2>&1
find /tmp/p_ops -name "*kernel_stats.csv" -exec cp {} $OUT/bench_ops_kernel_stats.csv \;
head -40 $OUT/bench_ops_kernel_stats.csv | cut -c1-200
