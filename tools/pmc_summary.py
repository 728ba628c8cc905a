#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv rows per kernel (later launches only) -> markdown / json."""
import csv, sys, collections, json
def main(paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {c: (sum(v[len(v)//3:]) / max(len(v[len(v)//3:]), 1)) for c, v in cs.items()}
    return out
if __name__ == "__main__":
    o = main(sys.argv[1:])
    for k, cs in o.items():
        if "at::native" in k or "rocclr" in k: continue
        print(k[:40], " ".join(f"{c}={v:.4g}" for c, v in sorted(cs.items())))
