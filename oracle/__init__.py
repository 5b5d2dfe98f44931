"""CPU oracle for the Hummingbird hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (open-hummingbird-eval_amd/hbird_mi) never does.
"""
from .oracle import *  # noqa: F401,F403
