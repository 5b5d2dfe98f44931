#!/usr/bin/env python3
"""Start offsets and durations of the last N kernels of a rocprofv3 kernel trace (one search: its phases, floor kernels, merge): usage trace_phase_durations.py kernel_trace.csv N"""
import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
sel=rows[-int(sys.argv[2]):]
t0=sel[0][0]
for s,e,n in sel: print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  {n[:60]}")
