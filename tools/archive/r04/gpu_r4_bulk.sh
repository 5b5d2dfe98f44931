#!/bin/bash
# Round 4: the survivors-per-tile level at which the pool epilogue switches to the quarter loop (HB_BULK_QUADS; experiment builds of one
# unit in lib/abl/): kernel ms of use_fp16 (or fp32 with AB_FP16 unset) searches at small and mid-size shapes.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_bulk; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
LIBS="$L/libhbird_hip.so"; for n in $LIBNAMES; do LIBS="$LIBS $L/abl/libhbird_hip_$n.so"; done
for shape in "50176 384 12544 30" "50176 384 21904 90" "300000 384 21904 30" "300000 768 21904 90" "1250000 768 21904 30" "1250000 768 21904 90" "10000000 768 21904 30"; do
  timeout 900 python tools/ab_lib.py $shape $LIBS 2>&1 | grep same | sed "s/^/$shape: /" | tee -a $OUT/t_${TAG:-x}.txt
done
