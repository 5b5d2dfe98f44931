#!/bin/bash
python tools/exp_variant.py 5000000 768 21904 30 0
for f in prio1 prio2; do HBIRD_HIP_LIB=$GRAFT_REPO_ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_$f.so python tools/exp_variant.py 5000000 768 21904 30 0 | sed "s/^/$f /"; done
python tools/exp_variant.py 5000000 768 21904 30 0
