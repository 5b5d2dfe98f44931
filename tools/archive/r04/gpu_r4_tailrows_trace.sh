#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r4_tailrows_trace; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
for lib in $L/abl/libhbird_hip_nocarry.so $L/libhbird_hip.so; do
for shape in "10000000 768 21904 90" "5000000 768 12544 30"; do
  tag=$(basename $lib .so)_$(echo $shape | tr ' ' '_')
  echo "== $tag" | tee -a $OUT/summary.txt
  (cd /tmp && HBIRD_HIP_LIB=$lib rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python3 $ROOT/tools/trace_small.py run $shape f16 > $OUT/run_$tag.log 2>&1)
  f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_small.py parse $f 10 2>&1 | head -8 | tee -a $OUT/summary.txt
done; done
