#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_knn_gpu.py -m gpu -x -q -k "cluster" 2>&1 | tail -1
EXP_MODES=f32 EXP_ROUNDS=3 EXP_CFGS="1,1,0;0,0,-1" python tools/exp_cluster.py | sed "s/^/final /"
export EXP_CL=8,1,16
python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1
for f in f16p64 f16p128; do HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_$f.so python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1; done
python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1
