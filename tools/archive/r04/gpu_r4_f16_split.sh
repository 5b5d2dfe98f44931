#!/bin/bash
# Round 4: what the fp16 candidate kernel's tile epilogue costs at 10 M x 768 -- timing-only experiment builds (lib/abl/libhbird_hip_x_*.so:
# no epilogue / no epilogue and no accumulator initialisation / a plain 128-value max instead of the threshold scan) against the product.
# Their results are wrong by construction (every query falls back to the fp32 search, outside the timed kernel).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_f16_split; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
AB_FP16=1 timeout 1500 python tools/ab_lib.py 10000000 768 21904 30 $L/libhbird_hip.so $L/abl/libhbird_hip_x_noepi.so $L/abl/libhbird_hip_x_noepi_noinit.so $L/abl/libhbird_hip_x_scanonly.so 2>&1 | grep same | tee -a $OUT/t.txt
