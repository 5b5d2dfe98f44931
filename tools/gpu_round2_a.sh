#!/bin/bash
# round-2 first GPU session: full GPU test-suite, default bench (live traffic), counter list
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -15 gpurun_out/r2a/pytest.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
cat gpurun_out/r2a/bench.json | head -c 3000
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r2a/counters.txt 2>&1)
grep -i -E "EA0?_RDREQ|MALL|DRAM|HBM|EA_RD|WRREQ" gpurun_out/r2a/counters.txt | head -60
