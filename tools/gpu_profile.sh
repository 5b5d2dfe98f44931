#!/bin/bash
# rocprofv3 evidence for the headline bench: kernel-trace stats + HBM traffic counters (separate passes)
set -x
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-r1}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | xargs ls -la
# keep only the small summaries (kernel stats + the knn kernel's counter rows)
for d in trace pmc_fetch pmc_write; do
  find $OUT/$d -name "*kernel_stats.csv" -exec cp {} $OUT/${d}_kernel_stats.csv \;
  find $OUT/$d -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_|rows_to_tiles|aggregate|query_aux" "$1" >> "$2"' _ {} $OUT/${d}_counters.csv \;
  find $OUT/$d -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_|aggregate_kernel|query_aux|rows_to_tiles_kernel<false, false>" "$1" | tail -40 >> "$2"' _ {} $OUT/${d}_kernel_trace_hot.csv \;
  rm -rf $OUT/$d
done
python3 - <<PY
import csv, json
def val(f, ctr):
    rows = [r for r in csv.DictReader(open(f)) if "knn_fused" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    return sum(float(r["Counter_Value"]) for r in rows) / max(1, len(rows))
fetch = val("$OUT/pmc_fetch_counters.csv", "FETCH_SIZE"); write = val("$OUT/pmc_write_counters.csv", "WRITE_SIZE")
b = json.load(open("$OUT/bench_trace.json"))
out = {"kernel": "knn_fused_kernel", "workload": {k: b["config"][k] for k in ("bank_rows", "dim", "k", "queries_per_step")},
       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide streaming reads); counts L2 misses incl. Infinity-Cache hits",
       "rocprof_avg_kernel_ms": None}
for r in csv.DictReader(open("$OUT/trace_kernel_stats.csv")):
    if "knn_fused" in r["Name"]: out["rocprof_avg_kernel_ms"] = float(r["AverageNs"]) / 1e6
json.dump(out, open("$OUT/knn_traffic.json", "w"), indent=1)
print(out)
PY
ls -la $OUT; tail -3 $OUT/*.err
