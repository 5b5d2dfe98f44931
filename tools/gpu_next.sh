#!/bin/bash
export TMPDIR=/tmp
EXP_CL=8,1,16 python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1
for f in 256 128 16 32 64 1; do EXP_CL=8,1,16 HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_f16abl$f.so python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1; done
EXP_CL=8,1,16 python tools/exp_f16_abl.py 10000000 768 21904 0 | tail -1
