#!/bin/bash
HBIRD_KNN_VARIANT=2 timeout 600 python -m pytest tests/test_knn_gpu.py -m gpu -x -q -k "fp16 or random_shapes or candidate_pool" 2>&1 | tail -4
