#!/bin/bash
python tools/exp_variant.py 1250000 768 21904 30 4,3
python tools/exp_variant.py 600000 768 21904 30 4,3
python tools/exp_variant.py 300000 768 21904 30 4,3
python tools/exp_variant.py 200000 384 12544 30 4,3
python tools/exp_variant.py 50176 384 12544 30 4,3
