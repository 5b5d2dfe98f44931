#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2q
EXP_ROUNDS=2 EXP_MODES=f32 EXP_ROWS=2000000 EXP_CFGS="1,1,0;2,2,0;2,2,16;2,2,8;2,2,4" EXP_OUT=r2q/exp_cluster.json timeout 1200 python tools/exp_cluster.py 2>&1 | tail -22
