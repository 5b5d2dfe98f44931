#!/bin/bash
# Round 4, experiment 1: XCD-level query-tile sharing of the clustered work lists (hb_index_set_cluster_sharing) -- kernel ms of the fp16
# candidate kernel and the fp32 kernel at 10 M x 768 per (cluster shape, lag, sharing mode, panel, variant), then FETCH_SIZE of a subset.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_xs; mkdir -p $OUT
F16="8,1,-1;8,1,-1,2;4,1,-1,2;4,1,-1,2,256;4,2,-1,2;4,2,-1,2,256;2,1,-1,2;4,1,0,2;4,1,-1,1;8,1,-1,0,256;8,1,-1,0,0,5;4,1,-1,2,0,5;4,1,-1,2,256,5"
EXP_MODES=f16 EXP_ROUNDS=2 EXP_CFGS="$F16" EXP_OUT=r4_xs/f16.json timeout 900 python tools/exp_cluster.py > $OUT/f16.txt 2>&1; cat $OUT/f16.txt | tail -20
F32="2,4,-1;2,4,-1,2;2,2,-1,2;2,1,-1,2;4,1,-1,2;1,1,0"
EXP_MODES=f32 EXP_ROUNDS=1 EXP_CFGS="$F32" EXP_OUT=r4_xs/f32.json timeout 900 python tools/exp_cluster.py > $OUT/f32.txt 2>&1; cat $OUT/f32.txt | tail -10
P16="8,1,-1;4,1,-1,2;4,1,-1,2,256;4,2,-1,2,256;8,1,-1,2"
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$tag -- python3 $ROOT/tools/pmc_cluster.py 10000000 1 "$P16" > /dev/null 2>&1)
  find /tmp/pmc_$tag -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_fused|knn_f16" "$1" >> "$2"' _ {} $OUT/pmc16_$tag.csv \;
  rm -rf /tmp/pmc_$tag
done
P32="2,4,-1;2,4,-1,2;2,1,-1,2"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f32 -- python3 $ROOT/tools/pmc_cluster.py 10000000 0 "$P32" > /dev/null 2>&1)
find /tmp/pmc_f32 -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_fused|knn_f16" "$1" >> "$2"' _ {} $OUT/pmc32_FETCH_SIZE.csv \;
rm -rf /tmp/pmc_f32
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/pmc*.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        d = agg.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f)
    for k in sorted(agg):
        if agg[k]["ms"] > 20: print("  ", k, {a: (round(b, 1) if a == "ms" else f"{b:.4g}") for a, b in agg[k].items()})
PY
