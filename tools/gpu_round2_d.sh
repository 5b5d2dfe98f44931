#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
EXP_ROUNDS=2 EXP_MODES=f32 EXP_CFGS="1,1,0;8,1,16;8,1,8;4,2,16;2,2,16" EXP_OUT=r2s/exp_cluster.json timeout 1200 python tools/exp_cluster.py 2>&1 | tail -22
