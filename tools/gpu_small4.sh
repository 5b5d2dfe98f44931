#!/bin/bash
export TMPDIR=/tmp
for lim in 150000 100000000; do
for shape in "1250000 768 21904" "2500000 768 21904" "5000000 768 21904" "10000000 768 21904"; do set -- $shape
  HBIRD_COLD_LIMIT=$lim python bench.py --rows $1 --dim $2 --nq $3 --classes 21 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('limit $lim', '$shape', round(r['ms_per_step'],2), round(r['roofline']['avg_kernel_ms'],2), round(r['roofline']['frac'],4))"
done; done
