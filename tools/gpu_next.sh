#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_knn_gpu.py -m gpu -x -q -k "register_resident or wide or candidate_pool or k90 or random" 2>&1 | tail -2
for r in 1 2; do
HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so python tools/exp_variant.py 5000000 768 21904 90 0 | sed "s/^/prev k90 /"
python tools/exp_variant.py 5000000 768 21904 90 0 | sed "s/^/new  k90 /"
done
HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/prev k30 /"
python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/new  k30 /"
