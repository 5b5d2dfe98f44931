#!/bin/bash
# Round 4: variants of the exact re-rank kernel (experiment builds in lib/abl/), whole-search ms of small use_fp16 searches.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_rr_ab; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib/abl
LIBS=""; for n in $@; do LIBS="$LIBS $L/libhbird_hip_$n.so"; done
for shape in "50176 384 12544 30" "16384 384 12544 30" "300000 768 12544 30"; do
  AB_WALL=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $LIBS 2>&1 | grep same | sed "s/^/$shape: /" | tee -a $OUT/t.txt
done
