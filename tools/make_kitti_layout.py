#!/usr/bin/env python3
"""Writes a synthetic stereo sequence in the KITTI directory layout (image_0/, image_1/, times.txt; 8-bit grey PNGs) for the C++
sequence driver: python tools/make_kitti_layout.py <dir> <pairs>; then tests/cpp/_build/stereo_kitti <dir> [--gather root] ..."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from refactored_orb_slam2_amd import synth
seq = sys.argv[1]; n = int(sys.argv[2])
os.makedirs(os.path.join(seq, "image_0")); os.makedirs(os.path.join(seq, "image_1"))
pairs = synth.sequence(1241, 376, n, seq=3, stereo=True)
with open(os.path.join(seq, "times.txt"), "w") as f:
    for i, (L, R) in enumerate(pairs):
        bench._write_png_gray(os.path.join(seq, "image_0", f"{i:06d}.png"), L)
        bench._write_png_gray(os.path.join(seq, "image_1", f"{i:06d}.png"), R)
        f.write(f"{i * 0.1:e}\n")
