#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_s7; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_eval_gpu.py tests/test_dist_gpu.py tests/test_boundary_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python tools/bench_ops.py $OUT/bench_ops.json > $OUT/bench_ops.txt 2>&1; grep -v amdgpu $OUT/bench_ops.txt
