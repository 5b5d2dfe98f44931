#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/small
python bench.py --rows 50176 --dim 384 --nq 12544 --classes 21 --steps 200 --warmup 20 --no-cpu-baseline --no-traffic > gpurun_out/small/bench.json 2>/dev/null
python - <<'PY'
import json; r = json.load(open("gpurun_out/small/bench.json")); print(r["ms_per_step"], r["roofline"]["avg_kernel_ms"], r["roofline"]["frac"], r["config"]["schedule"])
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/small_trace -- python3 $GRAFT_REPO_ROOT/bench.py --rows 50176 --dim 384 --nq 12544 --classes 21 --steps 50 --warmup 5 --no-cpu-baseline --no-traffic > /dev/null 2>&1)
find /tmp/small_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/small/kernel_stats.csv \;
find /tmp/small_trace -name "*kernel_trace.csv" -exec cp {} gpurun_out/small/kernel_trace.csv \;
head -12 gpurun_out/small/kernel_stats.csv | cut -c1-200
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/small/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 dispatches: name, duration, gap from previous end
last = rows[-40:]
prev = None
for r in last[-16:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Kernel_Name"][:50].ljust(50), "dur_us", round((e - s) / 1e3, 1), "gap_us", None if prev is None else round((s - prev) / 1e3, 1))
    prev = e
PY
