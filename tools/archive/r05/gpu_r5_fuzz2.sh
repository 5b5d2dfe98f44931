#!/bin/bash
# differential fuzz against the chain oracle: weighted work lists (random per-XCD shares), the default paths with new seeds, one-launch mode
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_fuzz2; mkdir -p $OUT
FUZZ_XCD=1 timeout 1200 python tests/fuzz_small.py 80 21 > $OUT/fuzz_xcd_80_cases.txt 2>&1; echo "xcd rc=$?"; tail -1 $OUT/fuzz_xcd_80_cases.txt
timeout 1200 python tests/fuzz_small.py 80 22 > $OUT/fuzz_default_80_cases.txt 2>&1; echo "default rc=$?"; tail -1 $OUT/fuzz_default_80_cases.txt
FUZZ_ONE_LAUNCH=1 timeout 1500 python tests/fuzz_small.py 40 23 > $OUT/fuzz_one_launch_40_cases.txt 2>&1; echo "one-launch rc=$?"; tail -1 $OUT/fuzz_one_launch_40_cases.txt
grep -c "MISMATCH" $OUT/*.txt
