#!/bin/bash
# the very last fuzz of round 5 on the final tree (new seeds): weighted lists in both kernel families, default paths, the list kernel, one-launch mode
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_fuzz5; mkdir -p $OUT
FUZZ_XCD=1 timeout 2400 python tests/fuzz_small.py 1200 91 > $OUT/fuzz_xcd_1200.txt 2>&1; echo "xcd rc=$?"; tail -1 $OUT/fuzz_xcd_1200.txt
timeout 2400 python tests/fuzz_small.py 800 92 > $OUT/fuzz_default_800.txt 2>&1; echo "default rc=$?"; tail -1 $OUT/fuzz_default_800.txt
FUZZ_XCD=1 FUZZ_ONE_LAUNCH=1 timeout 2400 python tests/fuzz_small.py 200 93 > $OUT/fuzz_xcd_one_launch_200.txt 2>&1; echo "xcd one-launch rc=$?"; tail -1 $OUT/fuzz_xcd_one_launch_200.txt
FUZZ_MID=1 FUZZ_XCD=1 timeout 2400 python tests/fuzz_small.py 100 94 > $OUT/fuzz_mid_xcd_100.txt 2>&1; echo "mid xcd rc=$?"; tail -1 $OUT/fuzz_mid_xcd_100.txt
grep -c "MISMATCH" $OUT/*.txt; true
