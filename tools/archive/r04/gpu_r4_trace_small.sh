#!/bin/bash
# Round 4: kernel traces of small searches (cfg-1 shape, fp16 and fp32 mode): kernels vs gaps per search.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r4_trace_small; mkdir -p $OUT
for mode in f16 f32; do
  for shape in "50176 384 12544 30" "16384 384 12544 30"; do
    tag=${mode}_$(echo $shape | tr ' ' '_')
    python3 tools/trace_small.py run $shape $mode 2>&1 | grep "per search" | tee -a $OUT/summary.txt
    (cd /tmp && rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python3 $ROOT/tools/trace_small.py run $shape $mode > $OUT/run_$tag.log 2>&1)
    f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
    python3 tools/trace_small.py parse $f 20 2>&1 | tee -a $OUT/summary.txt
  done
done
