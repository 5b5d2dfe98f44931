#!/bin/bash
# a longer differential fuzz against the chain oracle (tests/fuzz_small.py): weighted work lists, default paths, one-launch mode
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_fuzz3; mkdir -p $OUT
FUZZ_XCD=1 timeout 1500 python tests/fuzz_small.py 500 31 > $OUT/fuzz_xcd_500_cases.txt 2>&1; echo "xcd rc=$?"; tail -1 $OUT/fuzz_xcd_500_cases.txt
timeout 1500 python tests/fuzz_small.py 500 32 > $OUT/fuzz_default_500_cases.txt 2>&1; echo "default rc=$?"; tail -1 $OUT/fuzz_default_500_cases.txt
FUZZ_ONE_LAUNCH=1 timeout 1500 python tests/fuzz_small.py 200 33 > $OUT/fuzz_one_launch_200_cases.txt 2>&1; echo "one-launch rc=$?"; tail -1 $OUT/fuzz_one_launch_200_cases.txt
FUZZ_MAX_ROWS=600000 timeout 1500 python tests/fuzz_small.py 60 34 > $OUT/fuzz_bigger_banks_60_cases.txt 2>&1; echo "bigger rc=$?"; tail -1 $OUT/fuzz_bigger_banks_60_cases.txt
FUZZ_XCD=1 FUZZ_MAX_ROWS=600000 timeout 1500 python tests/fuzz_small.py 60 35 > $OUT/fuzz_xcd_bigger_banks_60_cases.txt 2>&1; echo "xcd bigger rc=$?"; tail -1 $OUT/fuzz_xcd_bigger_banks_60_cases.txt
grep -c "MISMATCH" $OUT/*.txt; true
