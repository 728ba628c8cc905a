#!/bin/bash
# Kernel + memory-copy trace of the C++ batched pipeline (examples/stereo_kitti.cc --batch 256 --preload 2): tools/pipeline_trace.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT/seq
python3 - <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from refactored_orb_slam2_amd import synth
seq = os.path.join("$OUT", "seq")
os.makedirs(os.path.join(seq, "image_0"), exist_ok=True); os.makedirs(os.path.join(seq, "image_1"), exist_ok=True)
pairs = synth.sequence(1241, 376, 64, seq=7, stereo=True)
with open(os.path.join(seq, "times.txt"), "w") as f:
    for i, (L, R) in enumerate(pairs):
        bench._write_png_gray(os.path.join(seq, "image_0", f"{i:06d}.png"), L, 1)
        bench._write_png_gray(os.path.join(seq, "image_1", f"{i:06d}.png"), R, 1)
        f.write(f"{i * 0.1:e}\n")
PY
make -s -C tests/cpp _build/stereo_kitti
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tr -- tests/cpp/_build/stereo_kitti $OUT/seq --batch 256 --preload 2 --repeat 48 --decode-threads 8 > $OUT/run.log 2>&1
tail -8 $OUT/run.log
python3 - <<PY
import csv, glob
cp = list(csv.DictReader(open(glob.glob("$OUT/*tr_memory_copy_trace.csv")[0])))
kt = list(csv.DictReader(open(glob.glob("$OUT/*tr_kernel_trace.csv")[0])))
big = [r for r in cp if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
t0 = min(int(r["Start_Timestamp"]) for r in big)
print("copies > 0.2 ms:", len(big))
for r in big[-24:]:
    print(r["Direction"], round((int(r["Start_Timestamp"]) - t0) / 1e6, 3), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
fast = [r for r in kt if "fast_cells" in r["Kernel_Name"]]
print("fast launches:", len(fast))
for r in fast[-8:]:
    print("fast", round((int(r["Start_Timestamp"]) - t0) / 1e6, 3), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
PY
