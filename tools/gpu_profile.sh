#!/bin/bash
# rocprofv3 evidence for the headline bench (round 2): kernel-trace stats + counter passes of the SHIPPED kernels, 10 M x 768.
# usage: tools/gpu_profile.sh <tag>     -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/<round>/)
set -x
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-r3}
mkdir -p $OUT
cd /tmp
B="python3 $ROOT/bench.py --timed-only"
# the driver's warm-up (5: the share calibration and the cluster decision settle in it) and 5 timed steps, nothing else: the statistics of the
# kernel the timed steps ran are those of the timed steps (+ the last warm-up searches in the same form)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- $B --steps 5 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
# the counter passes run ONE search of a fresh index: in the form the timed steps ran (clusters kept or dropped by measurement)
CS=$(python3 -c "import json; d = json.load(open('$OUT/bench_under_rocprof.json')); print(*d['config']['schedule']['cluster'])")
B="$B --cluster-shape $CS"
find /tmp/p_trace -name "*kernel_stats.csv" -exec cp {} $OUT/knn_10Mx768_kernel_stats.csv \;
find /tmp/p_trace -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_|aggregate_kernel|query_aux|rerank|rows_to_tiles_kernel<false, false>|tiles_to_f16" "$1" | tail -60 >> "$2"' _ {} $OUT/knn_10Mx768_kernel_trace_hot.csv \;
pass() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/p_$name -- $B --steps 1 --warmup 0 > /dev/null 2> $OUT/$name.err
  find /tmp/p_$name -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_fused|knn_f16" "$1" >> "$2"' _ {} $OUT/knn_10Mx768_pmc_$name.csv \;
  rm -rf /tmp/p_$name
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass EA_LATENCY TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum
pass SQ_GRBM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pass LDS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, json, os
out = {}
for f in sorted(glob.glob("$OUT/knn_10Mx768_pmc_*.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0][:40], int(r["Dispatch_Id"]))
        d = agg.setdefault(key, {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out[os.path.basename(f)] = {f"{k[0]}#{k[1]}": v for k, v in agg.items()}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for f, d in out.items():
    print(f)
    for k, v in d.items(): print("   ", k, {a: (round(b, 2) if a == "ms" else f"{b:.5g}") for a, b in v.items()})
PY
ls -la $OUT; tail -2 $OUT/*.err
