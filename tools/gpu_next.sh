#!/bin/bash
export TMPDIR=/tmp
export EXP_CL=8,1,16
HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_stamps.so python tools/exp_f16_abl.py 10000000 768 21904 0 2>&1 | grep -E "STAMPS|cluster" | head -40
