#!/bin/bash
export TMPDIR=/tmp
for e in 128 64 192; do HBIRD_POOL_EXTRA=$e EXP_CL=0,0,-1 python tools/exp_f16_abl.py 10000000 768 21904 0 2>&1 | tail -1; done
