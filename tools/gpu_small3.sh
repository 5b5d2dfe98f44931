#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not full_size and not headline" 2>&1 | tail -2
for shape in "50176 384 12544" "200000 384 12544" "2074072 384 12544"; do set -- $shape
  python bench.py --rows $1 --dim $2 --nq $3 --classes 21 --steps 50 --warmup 10 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('$shape', round(r['ms_per_step'],3), round(r['roofline']['avg_kernel_ms'],3), round(r['roofline']['frac'],3))"
done
