import os, time, torch, numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket|NUMA'")
os.system("cat /proc/loadavg; free -g | head -2")
try:
    import scann; print("scann", scann.__version__)
except Exception as e: print("scann import:", repr(e))
D=768; ms=400_000
b=torch.randn(ms,D); q=torch.randn(4096,D)
for nt in (8,16,32,64,128):
    torch.set_num_threads(nt)
    (q[:256]@b.T)
    t=time.time(); 
    for i in range(0,4096,1024): (q[i:i+1024]@b.T)
    dt=time.time()-t
    print("threads", nt, "mm TFLOP/s", round(2*4096*ms*D/dt/1e12,3), flush=True)
