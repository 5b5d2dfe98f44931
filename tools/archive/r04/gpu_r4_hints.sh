#!/bin/bash
# Round 4: cache-policy bits on the fp16 candidate kernel's bank copies (global_load_lds_dwordx4 nt / sc1 / sc0) and query-fragment loads
# (experiment builds lib/abl/libhbird_hip_x_{dma,bl}_*.so): kernel ms, clustered headline and unclustered mid sizes.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_hints; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
LIBS="$L/libhbird_hip.so"; for n in x_dma_nt x_dma_sc1 x_dma_sc1nt x_dma_sc0 x_bl_sc0 x_bl_nt; do LIBS="$LIBS $L/abl/libhbird_hip_$n.so"; done
for shape in "10000000 768 21904 30" "2074072 384 12544 30" "1250000 768 12544 30" "10000000 768 196 30"; do
  AB_MS2=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $LIBS 2>&1 | grep same | sed "s/^/fp16 $shape: /" | tee -a $OUT/t.txt
done
