#!/bin/bash
# round 5, first GPU session: the one-launch search -- its tests, the whole GPU suite, then timing against a launch per phase
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_ol1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "one_launch or phased" > $OUT/pytest_ol.log 2>&1; tail -5 $OUT/pytest_ol.log
timeout 1500 python tools/exp_one_launch.py 50176 384 12544 30 f32 50176 384 12544 30 f16 300000 768 12544 30 f16 300000 768 12544 30 f32 2074072 384 12544 30 f16 50176 384 12544 90 f32 300000 768 21904 90 f16 > $OUT/exp_one_launch.txt 2>&1; cat $OUT/exp_one_launch.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
