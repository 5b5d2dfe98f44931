#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_stamps; mkdir -p $OUT
timeout 900 python tools/exp_wg_stamps.py 10000000 768 21904 30 f32 > $OUT/stamps_headline.txt 2>&1; grep -v amdgpu $OUT/stamps_headline.txt
EXP_NO_CLUSTERS=1 timeout 900 python tools/exp_wg_stamps.py 5000000 768 21904 30 f32 > $OUT/stamps_5M_noclusters.txt 2>&1; grep -v amdgpu $OUT/stamps_5M_noclusters.txt | tail -10
