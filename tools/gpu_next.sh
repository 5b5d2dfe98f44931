#!/bin/bash
export TMPDIR=/tmp
python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/default /"
for b in 1 2 4 8 15; do HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_bdabl$b.so python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/abl $b /"; done
python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/default /"
