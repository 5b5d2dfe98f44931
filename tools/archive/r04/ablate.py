import sys, os, json, subprocess
for bits in (0, 1, 256, 384, 0):
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workgroups", str(-bits)], capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"ablate={bits:2d}  kernel_ms={r['roofline']['avg_kernel_ms']:.2f}  TF={r['roofline']['achieved']:.1f}", flush=True)
    except Exception as e:
        print(bits, "failed", out.stderr[-500:])
