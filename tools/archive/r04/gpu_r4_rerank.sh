#!/bin/bash
# Round 4: the exact re-rank kernel with block prefetch + scalar query loads + 64-bit ranking keys: fp16 parity tests, then the traces of
# small searches again (tools/gpu_r4_trace_small.sh).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/${1:-r4_rerank}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or f16 or nan or overflow or config or cfg or zero or denormal or outlier" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
bash tools/gpu_r4_trace_small.sh > $OUT/trace.txt 2>&1; cat $OUT/trace.txt | grep -v "^ .*rocclr\|query_aux\|tiles_to"
