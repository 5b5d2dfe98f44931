#!/bin/bash
HBIRD_KNN_VARIANT=3 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not headline and not full_size" 2>&1 | tail -2
python tools/exp_variant.py 5000000 768 21904 90 4,0
python tools/exp_variant.py 5000000 768 21904 30 4,0
