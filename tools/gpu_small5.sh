#!/bin/bash
export TMPDIR=/tmp
for fw in 0 1; do
for shape in "50176 384 12544" "200000 384 12544" "2074072 384 12544"; do set -- $shape
  if [ $fw = 1 ]; then export HBIRD_FORCE_WIDE=1; else unset HBIRD_FORCE_WIDE; fi
  python bench.py --rows $1 --dim $2 --nq $3 --classes 21 --steps 30 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('wide $fw', '$shape', round(r['ms_per_step'],3), round(r['roofline']['avg_kernel_ms'],3), round(r['roofline']['frac'],3))"
done; done
