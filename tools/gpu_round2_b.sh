#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
python -m pytest tests/test_knn_gpu.py -m gpu -x -q -k "clustered or partition or random_shapes" > gpurun_out/r2b/pytest.log 2>&1; tail -5 gpurun_out/r2b/pytest.log
EXP_OUT=r2b/exp_cluster.json timeout 900 python tools/exp_cluster.py 2>&1 | tail -30
