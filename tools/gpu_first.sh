#!/bin/bash
# first contact with the GPU: environment facts + the kNN parity tests
mkdir -p gpurun_out
{
  rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8
  lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"
  free -g | head -2
} > gpurun_out/env.txt 2>&1
python -m pytest tests/test_knn_gpu.py -x -q -m gpu 2>&1 | tail -30
