#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python eval.py --dataset-name synthetic --data-dir "" --d-model 3 --patch-size 8 --input-size 64 --device cuda 2>&1 | tail -3
