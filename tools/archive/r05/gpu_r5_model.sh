#!/bin/bash
# verdict r4 item 4b as a measurement: the fp16 stage-loop model (every operand random) with 32x32x16 and with 16x16x32 MFMAs, this round's box
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_model; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/f16_stage_model.hip -o /tmp/f16_stage_model 2> $OUT/build.err || cat $OUT/build.err
timeout 600 /tmp/f16_stage_model > $OUT/f16_stage_model.txt 2>&1; cat $OUT/f16_stage_model.txt
timeout 600 /tmp/f16_stage_model streams > $OUT/f16_stage_model_streams.txt 2>&1; cat $OUT/f16_stage_model_streams.txt
