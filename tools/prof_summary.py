#!/usr/bin/env python3
"""Condenses a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a small markdown table for profiles/."""
import csv
import sys


def main(path, out, title):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        f.write(f"# {title}\n\nSource: `rocprofv3 --kernel-trace --stats` (kernel_stats.csv), times in microseconds.\n\n")
        f.write("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            name = r["Name"].split("(")[0]
            if float(r["Percentage"]) < 0.05:
                continue
            f.write(f"| {name[:60]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.1f} | "
                    f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
