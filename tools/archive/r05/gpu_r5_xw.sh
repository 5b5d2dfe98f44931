#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_xw; mkdir -p $OUT
timeout 1500 python tools/exp_xcd_weights.py 10000000 768 21904 30 f32 10000000 768 21904 30 f16 2500000 768 21904 30 f32 > $OUT/xcd_weights.txt 2>&1; grep -v amdgpu $OUT/xcd_weights.txt
