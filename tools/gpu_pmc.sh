#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>" [bench args...]
export TMPDIR=/tmp
TAG=$1; CTRS=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
find $OUT/raw -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_fused|knn_f16" "$1" >> "$2"' _ {} $OUT/knn_counters.csv \;
rm -rf $OUT/raw
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/knn_counters.csv")))
agg={}
for r in rows:
    agg.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
    agg[r["Dispatch_Id"]]["ns"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for d,v in agg.items(): print(d, v)
PY
tail -2 $OUT/err.txt
