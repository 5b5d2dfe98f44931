#!/bin/bash
# verdict r4 item 4c as a measurement: how tightly the 32 clusters of the fp16 candidate kernel walk the bank TOGETHER is set by the panel
# (all clusters finish a panel of bank tiles before any starts the next): smaller panels = closer to chip-wide lockstep.  Kernel ms, clock
# and matrix-pipe busy share (counter pass) per panel size.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_panel; mkdir -p $OUT
for p in 16 32 64 0 256 512; do
  timeout 600 python bench.py --fp16 --panel $p --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $OUT/bench_fp16_panel_$p.json 2>/dev/null
done
python - <<'PY'
import json
for p in (16, 32, 64, 0, 256, 512):
    try: r = json.load(open(f"gpurun_out/r5_panel/bench_fp16_panel_{p}.json"))
    except Exception as e: print(p, "failed", e); continue
    ro = r["roofline"]
    print(f"panel {p if p else 'auto'}: {r['config']['schedule']['panel_tiles']} tiles, slots {r['config']['schedule']['slots']}, cluster {r['config']['schedule']['cluster']}: "
          f"{r['value']:.0f} q-p/s, kernel {ro['avg_kernel_ms']:.1f} ms = {ro['frac']:.3f}, clock {ro.get('clock_ghz')}, busy {ro.get('mfma_busy')}, traffic {ro.get('traffic')}")
PY
