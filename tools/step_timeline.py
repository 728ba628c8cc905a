#!/usr/bin/env python3
"""Concurrency picture of the overlapped bench step from a rocprofv3 kernel trace (tools/kstats.sh <dir>):
tools/step_timeline.py <dir> -- over the steady-state half of the trace: time with 0 / 1 / 2 / 3+ kernels resident, per-kernel
union time, and for every kernel name the mean duration alone-in-time vs overlapped."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]) for r in rows)
# the timed region of tools/kstats.sh's run (bench.py --steps 10 --warmup 2: 2 + 3 + 16 + 3 untimed steps in front of it; a step
# ends with proj_resolve): N = argv[2] (8) steps from step index argv[3] (24)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
best = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ends = [e for s, e, n in ev if "proj_resolve" in n]
t0, t1 = ends[best], ends[best + N]
print(f"{N} steps in {(t1 - t0) / 1e6:.3f} ms = {(t1 - t0) / 1e6 / N:.4f} ms per step")
ev = [(max(s, t0), min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
pts = []
for s, e, n in ev:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
conc = collections.Counter(); cur = 0; last = t0
active = collections.Counter(); by_set = collections.Counter()
for t, d, n in pts:
    conc[min(cur, 3)] += t - last
    if cur:
        key = "+".join(sorted(k.split("_kernel")[0][:14] for k, v in active.items() if v > 0))
        by_set[key] += t - last
    last = t; cur += d; active[n] += d
tot = sum(conc.values())
print("window %.3f ms; kernels resident: " % (tot / 1e6) + ", ".join(f"{k}{'+' if k == 3 else ''}: {100 * v / tot:.1f} %" for k, v in sorted(conc.items())))
print("largest co-residency sets (share of the window):")
for k, v in by_set.most_common(14):
    print(f"  {100 * v / tot:5.1f} %  {k}")
dur = collections.defaultdict(list)
for s, e, n in ev:
    dur[n].append((e - s) / 1e3)
print("per kernel: launches, mean us, sum ms")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {n:42s} {len(v):5d} {sum(v) / len(v):9.1f} {sum(v) / 1e3:8.2f}")
