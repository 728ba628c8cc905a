#!/usr/bin/env python3
"""Latency of the synchronous per-frame matcher calls on one KITTI-sized frame pair (the device + transfer share of what the C++
drop-in's SearchByProjection(cur, last) and ComputeStereoMatches pay per frame; the host adapter's query building is extra)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth
from refactored_orb_slam2_amd.matcher import FrameView, ORBmatcher

w, h, nf = 1241, 376, 2000
a, b = synth.sequence(w, h, 2, seq=9)[:2]
ex = ORBextractor(nf, device=0)
k0, d0 = ex(a); k1, d1 = ex(b)
sf = ex.GetScaleFactors()
q = np.zeros(len(k0), _lib.QUERY_DTYPE)
q["u"], q["v"] = k0["x"], k0["y"]; q["u_r"] = k0["x"] - 10
q["radius"] = np.float32(7.0) * sf[k0["octave"]]
q["min_level"], q["max_level"] = k0["octave"] - 1, k0["octave"] + 1
q["valid"] = 1; q["blocks"] = 1; q["angle"] = k0["angle"]; q["desc"] = d0
fv = FrameView(k1, d1, 0, w, 0, h)
m = ORBmatcher(0.9, True)
L = _lib.lib()
blocked = np.zeros(fv.n, np.uint8); assigned = np.full(fv.n, -1, np.int32); nm = C.c_int(0)

def call():
    blocked[:] = 0
    L.orbfe_search_by_projection_frame(C.byref(fv.c), _lib.ptr(q), len(q), 1, _lib.ptr(blocked), _lib.ptr(assigned), C.byref(nm))

for _ in range(30):
    call()
ts = []
for _ in range(400):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"orbfe_search_by_projection_frame {len(q)} queries -> {fv.n} keypoints: median {ts[len(ts)//2]*1e3:.3f} ms, p10 {ts[len(ts)//10]*1e3:.3f}, "
      f"p90 {ts[9*len(ts)//10]*1e3:.3f}, {nm.value} matches")
ex.close()

if True:
    try:
        fn = L.orbfe_debug_rs_profile
    except AttributeError:
        fn = None
    if fn is not None:   # library built with -DFC_TIMING=1 (tools/ab_build.sh rs "-DFC_TIMING=1" match_kernels.hip)
        out = (C.c_ulonglong * 16)()
        fn(out, 1)
        for _ in range(100):
            call()
        fn(out, 0)
        names = ["set-up", "query records + scan", "staging", "rounds", "commit", "rotation check + write-back"]
        tot = sum(out[i] for i in range(6))
        print("proj_resolve phases, cycles per call (workgroup 0): " + ", ".join(f"{n} {out[i] / 100:.0f}" for i, n in enumerate(names)) +
              f"; total {tot / 100:.0f}; chunks {out[8] / 100:.1f}, rounds {out[9] / 100:.1f}")
