#!/bin/bash
# Round 4: the phase schedule of pool searches (first cut, growth; experiment builds lib/abl/libhbird_hip_ph_<first>_<growth>.so) after the
# cheaper compaction / floor kernels: whole-search ms (AB_WALL), use_fp16 and fp32 pools, small to large.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_phases; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
LIBS="$L/libhbird_hip.so"; for n in ph_1_2 ph_1_3 ph_2_2 ph_2_3 ph_4_3; do LIBS="$LIBS $L/abl/libhbird_hip_$n.so"; done
for shape in "50176 384 12544 30" "50176 384 12544 90" "300000 768 21904 30" "300000 768 21904 90" "2074072 384 12544 30" "10000000 768 21904 30"; do
  AB_WALL=1 AB_MS2=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $LIBS 2>&1 | grep same | sed "s/^/fp16 $shape: /" | tee -a $OUT/t.txt
done
for shape in "50176 384 12544 30" "50176 384 12544 90" "300000 768 21904 90"; do
  AB_WALL=1 AB_MS2=1 timeout 900 python tools/ab_lib.py $shape $LIBS 2>&1 | grep same | sed "s/^/fp32 $shape: /" | tee -a $OUT/t.txt
done
