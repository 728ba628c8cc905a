#!/usr/bin/env python3
"""One-image calls (the hipGraph latency path) for extractors with 4, 8 and 12 pyramid levels: run under
rocprofv3 --kernel-trace and read where, inside a call's chain of launches, the device idles (tools/graph_gap_probe.sh)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from refactored_orb_slam2_amd import ORBextractor, synth
img = synth.sequence(1241, 376, 1, seq=3)[0]
for nl in (4, 8, 12):
    ex = ORBextractor(2000, 1.2, nl, 20, 7)
    for _ in range(12):
        k, d = ex(img)
    ex.close()
