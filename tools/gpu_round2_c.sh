#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
EXP_ROUNDS=2 EXP_CFGS="1,1,0;2,2,0;2,2,4008;2,2,8008;2,2,8016;2,2,16024" EXP_OUT=r2c/exp_cluster.json timeout 900 python tools/exp_cluster.py 2>&1 | tail -14
# TCC counters: plain vs clustered (no sync / sync), fp32 (2 M rows to keep it short) and fp16 (10 M rows)
cat > /tmp/pmc_one.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
import torch, bench
from hbird_mi.nn.search_hip import HipFlatIndex
M = int(sys.argv[1]); fp16 = int(sys.argv[2]); cfgs = [tuple(int(x) for x in c.split(",")) for c in sys.argv[3].split(";")]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ix = HipFlatIndex(768, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
bench.build_bank(ix, 0, M, 768, 21, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
q = 3.0 * torch.randn((21904, 768), generator=g, device=dev)
ix.set_fp16(bool(fp16))
for c in cfgs:
    ix.set_cluster(*c)
    ix.search(q, 30); torch.cuda.synchronize()
PY
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_f32_$tag -- python3 /tmp/pmc_one.py 2000000 0 "1,1,0;2,2,0;2,2,8008" > /dev/null 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_f16_$tag -- python3 /tmp/pmc_one.py 10000000 1 "1,1,0;2,2,0;2,2,8008" > /dev/null 2>&1)
  for m in f32 f16; do
    find /tmp/pmc_${m}_$tag -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > "$2"; grep -E "knn_fused|knn_f16" "$1" >> "$2"' _ {} gpurun_out/r2c/pmc_${m}_$tag.csv \;
  done
done
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r2c/pmc_*.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        d = agg.setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f)
    for k in sorted(agg): print("  ", k, {a: (round(b, 1) if a == "ms" else f"{b:.4g}") for a, b in agg[k].items()})
PY
