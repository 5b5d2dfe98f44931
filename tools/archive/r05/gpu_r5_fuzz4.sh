#!/bin/bash
# the last fuzz of round 5 (new seeds): default paths, weighted work lists, one-launch mode, the small-search list kernel
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_fuzz4; mkdir -p $OUT
timeout 2400 python tests/fuzz_small.py 1500 51 > $OUT/fuzz_default_1500.txt 2>&1; echo "default rc=$?"; tail -1 $OUT/fuzz_default_1500.txt
FUZZ_XCD=1 timeout 2400 python tests/fuzz_small.py 1000 52 > $OUT/fuzz_xcd_1000.txt 2>&1; echo "xcd rc=$?"; tail -1 $OUT/fuzz_xcd_1000.txt
FUZZ_ONE_LAUNCH=1 timeout 2400 python tests/fuzz_small.py 300 53 > $OUT/fuzz_one_launch_300.txt 2>&1; echo "one-launch rc=$?"; tail -1 $OUT/fuzz_one_launch_300.txt
FUZZ_MID=1 timeout 2400 python tests/fuzz_small.py 150 54 > $OUT/fuzz_mid_150.txt 2>&1; echo "mid rc=$?"; tail -1 $OUT/fuzz_mid_150.txt
FUZZ_MID=1 FUZZ_XCD=1 timeout 2400 python tests/fuzz_small.py 60 55 > $OUT/fuzz_mid_xcd_60.txt 2>&1; echo "mid xcd rc=$?"; tail -1 $OUT/fuzz_mid_xcd_60.txt
grep -c "MISMATCH" $OUT/*.txt; true
