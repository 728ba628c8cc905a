#!/bin/bash
# tools/graph_gap_probe.sh <outdir>: kernel timeline of the last one-image call per level count (4, 8, 12)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o gg -- python3 tools/graph_gap_probe.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
rows = sorted(csv.DictReader(open(glob.glob("$OUT/*gg_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "orient_describe" in r["Kernel_Name"]]
for call in (11, 23, 35):          # the last call of each extractor
    last = idx[call]; first = idx[call - 1] + 1
    prev = None
    print("---- call", call)
    for r in rows[first:last + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:36]:38s} {(e - s) / 1e3:7.1f} us  gap {((s - prev) / 1e3 if prev else 0):6.1f}")
        prev = e
PY
rm -f $OUT/*.csv
