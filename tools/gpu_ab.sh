#!/bin/bash
R=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib
python tools/ab_lib.py 5000000 768 21904 90 $R/abl/libhbird_hip_r01.so $R/libhbird_hip.so
python tools/ab_lib.py 5000000 768 21904 30 $R/abl/libhbird_hip_r01.so $R/libhbird_hip.so
python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not headline and not full_size" 2>&1 | tail -2
python tools/exp_f16_abl.py 10000000 768 21904 0,2 | tail -1
