#!/bin/bash
# Round 4: after the cheap compaction: slack of the bisection and pool capacity (experiment builds in lib/abl/), use_fp16 kernel ms.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_pooltune; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
LIBS="$L/libhbird_hip.so"; for n in $LIBNAMES; do LIBS="$LIBS $L/abl/libhbird_hip_$n.so"; done
for shape in "300000 768 21904 30" "300000 768 21904 90" "1250000 768 21904 90" "2074072 384 12544 30" "10000000 768 21904 30" "10000000 768 21904 90"; do
  AB_MS2=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $LIBS 2>&1 | grep same | sed "s/^/fp16 $shape: /" | tee -a $OUT/t.txt
done
