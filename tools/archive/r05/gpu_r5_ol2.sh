#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_ol2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "one_launch or phased" > $OUT/pytest_ol.log 2>&1; tail -5 $OUT/pytest_ol.log
timeout 900 python tools/exp_ol_trace.py 50176 384 12544 30 f32 50176 384 12544 30 f16 300000 768 12544 30 f16 2074072 384 12544 30 f16 > $OUT/trace.txt 2>&1; cat $OUT/trace.txt
timeout 900 python tools/exp_one_launch.py 50176 384 12544 30 f32 50176 384 12544 30 f16 300000 768 12544 30 f16 > $OUT/exp_one_launch.txt 2>&1; cat $OUT/exp_one_launch.txt
