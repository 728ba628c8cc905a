#!/usr/bin/env python3
"""Latency of the synchronous host API on one image (what the C++ drop-in's operator() pays per frame)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth

for (w, h, nf) in ((1241, 376, 2000), (640, 480, 1000)):
    img = synth.sequence(w, h, 1, seq=9)[0]
    ex = ORBextractor(nf, device=0)
    for _ in range(20):
        ex(img)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); k, d = ex(img); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{w}x{h} nf={nf}: median {ts[len(ts)//2]*1e3:.3f} ms, p10 {ts[len(ts)//10]*1e3:.3f} ms, p90 {ts[9*len(ts)//10]*1e3:.3f} ms, {len(k)} keypoints")
    ex.close()
