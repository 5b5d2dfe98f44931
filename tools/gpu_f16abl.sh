#!/bin/bash
export TMPDIR=/tmp
python tools/exp_f16_abl.py 10000000 768 21904 0,2
for f in open-hummingbird-eval_amd/lib/abl/*.so; do
  HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/$f python tools/exp_f16_abl.py 10000000 768 21904 0 2>&1 | tail -1
done
