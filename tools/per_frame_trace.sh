#!/bin/bash
# Kernel + memory-copy timeline of the per-frame path (examples/stereo_kitti.cc, one stereo pair at a time): tools/per_frame_trace.sh <outdir>
# Prints, for the last frames, every kernel / copy with its stream-relative start, so the two eyes' overlap and the host gaps show.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
make -s -C tests/cpp _build/stereo_kitti
rm -rf /tmp/pf_seq && python3 tools/make_kitti_layout.py /tmp/pf_seq 24 > /dev/null 2>&1
ARGS="/tmp/pf_seq --features 2000 --bf 386.1448 --fx 718.856 --fy 718.856 --cx 607.1928 --cy 185.2157 --th 7 --decode-threads 4 --prefetch 8"
tests/cpp/_build/stereo_kitti $ARGS > $OUT/plain.txt 2>&1; grep -E "median|two threads" $OUT/plain.txt
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o pf -- tests/cpp/_build/stereo_kitti $ARGS > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
k = list(csv.DictReader(open(glob.glob("$OUT/*pf_kernel_trace.csv")[0])))
m = list(csv.DictReader(open(glob.glob("$OUT/*pf_memory_copy_trace.csv")[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in k]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?")[:20], "c") for r in m]
ev.sort()
# the last frame: everything after the last-but-one orient_describe pair ... take the last 60 events
last = ev[-64:]
t0 = last[0][0]
for s, e, n, q in last:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  q{q:>4s}  {n}")
PY
rm -f $OUT/*.csv
