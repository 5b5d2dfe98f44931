#!/bin/bash
export TMPDIR=/tmp
for c in "8,1,16" "8,1,4" "8,1,8" "8,1,12" "8,1,24" "8,1,40" "1,1,0"; do EXP_CL=$c python tools/exp_f16_abl.py 10000000 768 21904 0 2>&1 | tail -1; done
