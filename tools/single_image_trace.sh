#!/bin/bash
# Kernel trace of the one-image host call (the latency path): tools/single_image_trace.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
python3 tools/single_image_latency.py > $OUT/latency.txt 2>&1; cat $OUT/latency.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 tools/single_image_latency.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
rows = sorted(csv.DictReader(open(glob.glob("$OUT/*tr_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
# the last 1241x376 call: find the last orient_describe before the 640x480 part (grid differs); take the 14 launches before it
idx = [i for i, r in enumerate(rows) if "orient_describe" in r["Kernel_Name"]]
last = idx[len(idx) // 2 - 1]
first = last
while first > 0 and "orient_describe" not in rows[first - 1]["Kernel_Name"]:
    first -= 1
prev = None
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]:42s} wg {r['Workgroup_Size_X']:>4s} grid {r['Grid_Size_X']:>7s}  {(e - s) / 1e3:7.1f} us  gap {((s - prev) / 1e3 if prev else 0):6.1f}")
    prev = e
print("span %.1f us" % ((int(rows[last]["End_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e3))
PY
