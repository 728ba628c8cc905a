#!/usr/bin/env python3
"""The C ABI's pipeline handle (orbfe_pipeline_*) at its kernels' rate: frames uploaded once, then orbfe_pipeline_submit_resident in a
ring of slots (results still copied to the pinned output blocks).  SLOTS (default 3), F (256), K chunks (60)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from refactored_orb_slam2_amd import synth
from refactored_orb_slam2_amd.pipeline import StereoPipeline
W, H, NF = 1241, 376, 2000
F, S, K = int(os.environ.get("F", "256")), int(os.environ.get("SLOTS", "3")), int(os.environ.get("K", "60"))
pairs = synth.sequence(W, H, 16, seq=0, stereo=True)
with StereoPipeline(W, H, F, 718.856, 718.856, 607.1928, 185.2157, 386.1448, 7.0, n_features=NF, slots=S) as p:
    for s in range(S):
        for j in range(F):
            p.left(s)[j, :, :W] = pairs[j % 16][0]; p.right(s)[j, :, :W] = pairs[j % 16][1]
        p.submit(s, F, has_predecessor=s > 0)
    for s in range(S):
        p.wait(s)
    res = {}
    for name, fn in (("upload", p.submit), ("resident", p.submit_resident)):
        t0 = time.perf_counter()
        for k in range(K):
            s = k % S
            p.wait(s)
            fn(s, F, has_predecessor=True)
        for s in range(S):
            p.wait(s)
        res[name + "_frames_per_s"] = round(F * K / (time.perf_counter() - t0), 1)
    res["keypoints_frame0"] = int(p.output(0)["n_left"][0]); res["tracked_frame1"] = int(p.output(0)["n_tracked"][1])
    res["slots"] = S; res["batch"] = F
print(json.dumps(res))
