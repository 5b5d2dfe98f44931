#!/bin/bash
# Round 4: kernel traces of mid-size use_fp16 searches (what the re-rank / merge / floors cost beside the candidate kernel), and the
# re-rank kernel's FETCH_SIZE (counter pass of its own) at 300,000 x 768.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r4_trace_mid; mkdir -p $OUT
for shape in "300000 768 12544 30" "1250000 768 21904 30" "2074072 384 12544 30"; do
  tag=f16_$(echo $shape | tr ' ' '_')
  python3 tools/trace_small.py run $shape f16 2>&1 | grep "per search" | tee -a $OUT/summary.txt
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/tr_$tag -o t --output-format csv -- python3 $ROOT/tools/trace_small.py run $shape f16 > $OUT/run_$tag.log 2>&1)
  f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_small.py parse $f 20 2>&1 | tee -a $OUT/summary.txt
done
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_rr -o t -- python3 $ROOT/tools/trace_small.py run 300000 768 12544 30 f16 > $OUT/run_pmc.log 2>&1)
f=$(find /tmp/pmc_rr -name "*counter_collection.csv" | head -1)
python3 - $f <<'PY' | tee -a $OUT/summary.txt
import csv, sys
tot = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0][:40]
    c = tot.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
for n, (c, v) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:6]:
    print(f"FETCH_SIZE raw per launch {v / c:14.1f} (x {c})  {n}")
PY
