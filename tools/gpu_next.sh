#!/bin/bash
export TMPDIR=/tmp
python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/place0 /"
for v in 1 2; do HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_place$v.so python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/place$v /"; done
python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/place0 /"
