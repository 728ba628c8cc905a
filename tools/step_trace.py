#!/usr/bin/env python3
"""Per-launch times of the last bench step in a rocprofv3 kernel trace: tools/step_trace.py <dir with *_kernel_trace.csv>
Prints every kernel launch of the last step (name, grid, duration, gap to the previous launch)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
last = max(i for i, r in enumerate(rows) if "proj_resolve" in r["Kernel_Name"])
first = max(i for i, r in enumerate(rows[:last]) if "proj_resolve" in r["Kernel_Name"]) + 1 if any("proj_resolve" in r["Kernel_Name"] for r in rows[:last]) else 0
prev = None
tot = 0
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]:46s} grid {r['Grid_Size_X']:>8s} x {r['Grid_Size_Y']:>4s}  {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}")
    tot += (e - s) / 1e3
    prev = e
print(f"sum of kernels {tot / 1e3:.3f} ms; span {(int(rows[last]['End_Timestamp']) - int(rows[first]['Start_Timestamp'])) / 1e6:.3f} ms")
