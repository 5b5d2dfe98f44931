#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r4_trace_k90; mkdir -p $OUT
for shape in "300000 768 21904 90" "50176 384 12544 90" "300000 768 21904 30" "50176 384 12544 30"; do
  tag=f16_$(echo $shape | tr ' ' '_')
  echo "== $shape" | tee -a $OUT/summary.txt
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python3 $ROOT/tools/trace_small.py run $shape f16 > $OUT/run_$tag.log 2>&1)
  f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_small.py parse $f 20 2>&1 | head -9 | tee -a $OUT/summary.txt
done
