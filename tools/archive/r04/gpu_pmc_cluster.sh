#!/bin/bash
# usage: gpu_pmc_cluster.sh <outdir> <rows> <fp16> "<cfgs>"
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$tag -- python3 $ROOT/tools/pmc_cluster.py $2 $3 "$4" > /dev/null 2>&1)
  find /tmp/pmc_$tag -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_fused|knn_f16" "$1" >> "$2"' _ {} $OUT/pmc_$tag.csv \;
  rm -rf /tmp/pmc_$tag
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/pmc_*.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        d = agg.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f)
    for k in sorted(agg): print("  ", k, {a: (round(b, 1) if a == "ms" else f"{b:.4g}") for a, b in agg[k].items()})
PY
