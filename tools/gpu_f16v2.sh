#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/f16v2
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or random_shapes or width_1024 or k90_width or cfg1 or candidate_pool or two_level or clustered" > gpurun_out/f16v2/pytest.log 2>&1; tail -3 gpurun_out/f16v2/pytest.log
python tools/exp_f16_abl.py 10000000 768 21904 0,2
EXP_CL=8,1,16 python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1
python tools/exp_f16_abl.py 2074072 384 12544 0,2 | tail -1
