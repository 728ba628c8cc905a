#!/usr/bin/env python3
"""Per-frame latency of the C++ drop-in path (examples/stereo_kitti.cc) on a synthetic KITTI-size sequence: bench.py's
per_frame_ms with the per-phase means the driver prints.  usage: python tools/per_frame_latency.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
print(bench.per_frame_latency(bench.CONFIGS["kitti_stereo"], n))
