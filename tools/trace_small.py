#!/usr/bin/env python3
"""Where a SMALL search's time goes: run under `rocprofv3 --kernel-trace -d <dir> -o small -- python3 tools/trace_small.py run rows dim nq k f16|f32`,
then `python3 tools/trace_small.py parse <kernel_trace.csv> <searches>` prints, for the last <searches> searches of the trace: wall span per
search, the sum of kernel durations, the idle gaps between kernels, and the per-kernel totals (launch count, ms).  The run also prints the
wall time per search by HIP events, without the profiler's per-launch cost the trace's gaps include."""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TIMED = 20

def run(argv):
    sys.path[:0] = [os.path.join(ROOT, "open-hummingbird-eval_amd"), ROOT]
    import torch, bench
    from hbird_mi.nn.search_hip import HipFlatIndex
    M, D, nq, k = (int(x) for x in argv[:4]); mode = argv[4]
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    ix = HipFlatIndex(D, 0, 0); ix.set_num_classes(21); ix.use_current_stream()
    bench.build_bank(ix, 0, M, D, 21, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    q = 3.0 * torch.randn((nq, D), generator=g, device=dev)
    ix.set_fp16(mode == "f16")
    for _ in range(5):
        ix.search(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N_TIMED):
        ix.search(q, k)
    e1.record(); torch.cuda.synchronize()
    print(f"{M} x {D}, nq {nq}, k {k}, {mode}: {e0.elapsed_time(e1) / N_TIMED:.3f} ms per search (HIP events over {N_TIMED} searches)", flush=True)

def parse(path, n_searches):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    per = len(rows)
    # the last n searches: every search launches the same kernel sequence, so cut by count
    names = [r[2] for r in rows]
    # period = smallest p such that the tail repeats with period p
    tail = names[-4000:]
    p = next(p for p in range(1, len(tail) // 2) if all(tail[-1 - i] == tail[-1 - i - p] for i in range(min(3 * p, len(tail) - p))))
    sel = rows[-p * n_searches:]
    span = (sel[-1][1] - sel[0][0]) / 1e6 / n_searches
    busy = sum(e - s for s, e, _ in sel) / 1e6 / n_searches
    gaps = sum(max(0, sel[i + 1][0] - sel[i][1]) for i in range(len(sel) - 1)) / 1e6 / n_searches
    print(f"{p} launches per search; per search: span {span:.3f} ms, kernels {busy:.3f} ms, idle between kernels {gaps:.3f} ms ({100 * gaps / span:.1f} %)")
    tot = {}
    for s, e, n in sel:
        n = n.split("(")[0]
        c = tot.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += (e - s) / 1e6
    for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {c // n_searches:3d} x {ms / n_searches:8.4f} ms  {n[:110]}")

if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        parse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else N_TIMED)
